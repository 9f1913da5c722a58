# Tile shape of the two stencil kernels (udc_mom_lds.hip: MOM_TX x MOM_TY, default 32 x 8 with three workgroups per CU): a variant
# library built with -DMOM_TX=64 -DMOM_TY=8 -DMOM_WAVES=1 (512 threads, 100 KB of LDS: one workgroup per CU) beside the default.
show='import json,sys; d=json.loads(sys.stdin.read()); k=d["kernels"]; print(sys.argv[1], round(d["ms_per_step"],3), {n:round(v["avg_ms_net"],3) for n,v in k.items() if n.startswith(("mom","closure"))})'
for lib in "" t64x8; do
  if [ -n "$lib" ]; then export UDC_LIBPATH=$GRAFT_REPO_ROOT/u-dales_amd/lib/libudcore_$lib.so; else unset UDC_LIBPATH; fi
  python bench.py --steps 60 --warmup 12 --no-dropin --no-cpu --no-pmc 2>/dev/null | tail -1 | python -c "$show" "256^3 lib=$lib"
  python bench.py --size 1024x512x512 --steps 4 --warmup 6 --no-dropin --no-cpu --no-pmc 2>/dev/null | tail -1 | python -c "$show" "1024x512x512 lib=$lib"
done
