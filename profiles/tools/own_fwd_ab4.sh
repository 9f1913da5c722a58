#!/bin/bash
cd /root/repo
timeout 600 python -m pytest tests/test_gpu_own_forward.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/own_fwd4_tests.txt
for rep in 1 2; do
python bench.py --steps 150 --warmup 12 --no-dropin --no-cpu --no-pmc 2>/dev/null | tail -1 > gpurun_out/own_fwd4_256_on_r$rep.json
done
python bench.py --size 512x512x256 --sgs smag --nsv 1 --steps 20 --warmup 6 --no-dropin --no-cpu --no-pmc 2>/dev/null | tail -1 > gpurun_out/own_fwd4_c2_on.json
python bench.py --size 1024x512x512 --steps 6 --warmup 6 --no-dropin --no-cpu --no-pmc 2>/dev/null | tail -1 > gpurun_out/own_fwd4_1024_on.json
python bench.py --size 128x128x128 --steps 150 --warmup 12 --no-dropin --no-cpu --no-pmc 2>/dev/null | tail -1 > gpurun_out/own_fwd4_128_on.json
