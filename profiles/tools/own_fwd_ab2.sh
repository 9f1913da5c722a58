#!/bin/bash
# second A/B of UDC_OWN_FWD: the 16 x 16 register y pass, x rows per workgroup 4 / 8
cd /root/repo
UDC_OWN_FWD=1 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/own_fwd2_parity.txt
for rep in 1 2; do
  python bench.py --steps 150 --warmup 12 --no-dropin --no-cpu --no-pmc 2>/dev/null | tail -1 > gpurun_out/own_fwd2_256_off_r${rep}.json
  for C in 4 8 16; do
    UDC_OWN_FWD=1 UDC_NAT_L=4 UDC_NAT_C=$C python bench.py --steps 150 --warmup 12 --no-dropin --no-cpu --no-pmc 2>/dev/null | tail -1 > gpurun_out/own_fwd2_256_L4_C${C}_r${rep}.json
  done
done
UDC_OWN_FWD=1 UDC_NAT_L=2 UDC_NAT_C=8 python bench.py --steps 150 --warmup 12 --no-dropin --no-cpu --no-pmc 2>/dev/null | tail -1 > gpurun_out/own_fwd2_256_L2_C8.json
UDC_OWN_FWD=1 UDC_NAT_L=8 UDC_NAT_C=8 python bench.py --steps 150 --warmup 12 --no-dropin --no-cpu --no-pmc 2>/dev/null | tail -1 > gpurun_out/own_fwd2_256_L8_C8.json
