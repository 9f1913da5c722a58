#!/bin/bash
cd /root/repo
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/full_gpu_suite_final_s3b.txt
python bench.py > gpurun_out/bench_default_final_s3b.json 2> gpurun_out/bench_default_final_s3b.err
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_s3b.txt 2>&1
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_s3b -o stats -- python /root/repo/bench.py --no-cpu --no-dropin --no-pmc --steps 50 --warmup 6 > /root/repo/gpurun_out/prof_s3b.log 2>&1
