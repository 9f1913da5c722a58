# Workgroup size of the x line-FFT kernels of the slab path (u-dales_amd/csrc/udc_fft.hip, xthreads): variant libraries built with
# -DUDC_XT_LM8=256 / 1024 beside the default, forced-slab bench lines at a grid with x lines of 256 complex, and the default at
# 1024 x 512 x 512 (lines of 512 complex -> 1024 threads).
show='import json,sys; d=json.loads(sys.stdin.read()); k=d["kernels"]; print(sys.argv[1], round(d["ms_per_step"],3), {n:round(v["avg_ms"],3) for n,v in k.items() if "fft" in n})'
for lib in "" xt8_256 xt8_1024; do
  if [ -n "$lib" ]; then export UDC_LIBPATH=$GRAFT_REPO_ROOT/u-dales_amd/lib/libudcore_$lib.so; else unset UDC_LIBPATH; fi
  UDC_FORCE_SLAB=1 python bench.py --size 512x512x256 --steps 6 --warmup 6 --no-dropin --no-cpu --no-pmc 2>/dev/null | tail -1 | python -c "$show" "512x512x256 lib=$lib"
done
unset UDC_LIBPATH
UDC_FORCE_SLAB=1 python bench.py --size 1024x512x512 --steps 4 --warmup 6 --no-dropin --no-cpu --no-pmc 2>/dev/null | tail -1 | tee gpurun_out/bench_1024_forced_slab_r03b.json | python -c "$show" "1024x512x512 forced slab"
python bench.py --size 1024x512x512 --steps 4 --warmup 6 --no-dropin --no-cpu --no-pmc 2>/dev/null | tail -1 | tee gpurun_out/bench_1024_r03b.json | python -c "$show" "1024x512x512 single"
UDC_FORCE_SLAB=1 python bench.py --steps 20 --warmup 10 --no-dropin --no-cpu --no-pmc 2>/dev/null | tail -1 | python -c "$show" "256^3 forced slab"
