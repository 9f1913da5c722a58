#!/bin/bash
# A/B of the register-carry closure sweep (UDC_CLOSURE_CARRY) : parity first, then alternating bench lines
set -x
export UDC_CLOSURE_CARRY=1
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/carry_parity.txt
unset UDC_CLOSURE_CARRY
for rep in 1 2 3; do
  for c in 0 1; do
    UDC_CLOSURE_CARRY=$c timeout 300 python bench.py --steps 150 --warmup 12 --no-dropin --no-cpu --no-pmc 2>/dev/null | tail -1 > gpurun_out/carry_256_c${c}_r${rep}.json
  done
done
for c in 0 1; do
  UDC_CLOSURE_CARRY=$c timeout 300 python bench.py --size 512x512x256 --sgs smag --nsv 1 --steps 20 --warmup 6 --no-dropin --no-cpu --no-pmc 2>/dev/null | tail -1 > gpurun_out/carry_c2_c${c}.json
  UDC_CLOSURE_CARRY=$c timeout 300 python bench.py --size 1024x512x512 --steps 6 --warmup 6 --no-dropin --no-cpu --no-pmc 2>/dev/null | tail -1 > gpurun_out/carry_1024_c${c}.json
done
