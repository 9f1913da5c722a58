#!/bin/bash
# third pass: the own forward half as the default (register y pass for ny = 128 / 256 / 512): its tests, the parity / long / large suites, bench lines
cd /root/repo
timeout 600 python -m pytest tests/test_gpu_own_forward.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/own_fwd3_tests.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_long.py tests/test_gpu_large.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/own_fwd3_suites.txt
python bench.py --steps 150 --warmup 12 --no-dropin --no-cpu --no-pmc 2>/dev/null | tail -1 > gpurun_out/own_fwd3_256_on.json
UDC_OWN_FWD=0 python bench.py --steps 150 --warmup 12 --no-dropin --no-cpu --no-pmc 2>/dev/null | tail -1 > gpurun_out/own_fwd3_256_off.json
for c in on off; do
  [ $c = off ] && export UDC_OWN_FWD=0
  python bench.py --size 512x512x256 --sgs smag --nsv 1 --steps 20 --warmup 6 --no-dropin --no-cpu --no-pmc 2>/dev/null | tail -1 > gpurun_out/own_fwd3_c2_$c.json
  python bench.py --size 1024x512x512 --steps 6 --warmup 6 --no-dropin --no-cpu --no-pmc 2>/dev/null | tail -1 > gpurun_out/own_fwd3_1024_$c.json
  python bench.py --size 128x128x128 --steps 150 --warmup 12 --no-dropin --no-cpu --no-pmc 2>/dev/null | tail -1 > gpurun_out/own_fwd3_128_$c.json
done
