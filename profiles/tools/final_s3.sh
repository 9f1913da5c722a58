#!/bin/bash
# end of round 3, session 3: full GPU suite, default bench line (with the live counter leg, CPU and drop-in legs), rocprofv3 stats + counters at 256^3
cd /root/repo
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/full_gpu_suite_final_s3.txt
python bench.py > gpurun_out/bench_default_final_s3.json 2> gpurun_out/bench_default_final_s3.err
bash profiles/tools/collect.sh r03c_256
python bench.py --size 512x512x256 --sgs smag --nsv 1 --steps 20 --warmup 6 --no-dropin --no-cpu 2>/dev/null | tail -1 > gpurun_out/bench_c2_final_s3.json
python bench.py --size 1024x512x512 --steps 12 --warmup 6 --no-dropin --no-cpu 2>/dev/null | tail -1 > gpurun_out/bench_1024_final_s3.json
