#!/bin/bash
# A/B of the one-GPU forward half in udc_fft.hip (UDC_OWN_FWD=1: divergence + x transform + y pass) against div_rhs + rocFFT's 2-D plan
cd /root/repo
UDC_OWN_FWD=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_long.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/own_fwd_parity.txt
for rep in 1 2; do
  python bench.py --steps 150 --warmup 12 --no-dropin --no-cpu --no-pmc 2>/dev/null | tail -1 > gpurun_out/own_fwd_256_off_r${rep}.json
  for C in 4 8 16; do
    UDC_OWN_FWD=1 UDC_NAT_C=$C python bench.py --steps 150 --warmup 12 --no-dropin --no-cpu --no-pmc 2>/dev/null | tail -1 > gpurun_out/own_fwd_256_C${C}_r${rep}.json
  done
done
for L in 4 16; do
  UDC_OWN_FWD=1 UDC_NAT_L=$L python bench.py --steps 150 --warmup 12 --no-dropin --no-cpu --no-pmc 2>/dev/null | tail -1 > gpurun_out/own_fwd_256_L${L}.json
done
python bench.py --size 1024x512x512 --steps 6 --warmup 6 --no-dropin --no-cpu --no-pmc 2>/dev/null | tail -1 > gpurun_out/own_fwd_1024_off.json
UDC_OWN_FWD=1 python bench.py --size 1024x512x512 --steps 6 --warmup 6 --no-dropin --no-cpu --no-pmc 2>/dev/null | tail -1 > gpurun_out/own_fwd_1024_on.json
