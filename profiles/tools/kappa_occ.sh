# kappa kernel: workgroups per CU limited through unused dynamic LDS (40.6 KB static: 4 per CU; +14 KB: 3; +41 KB: 2)
for pad in 0 14000 41000; do
UDC_KAPPA_LDSPAD=$pad python bench.py --size 512x512x256 --sgs smag --nsv 1 --steps 30 --warmup 9 --no-dropin --no-cpu --no-pmc 2>/dev/null | tail -1 > gpurun_out/kappa_occ_$pad.json
done
