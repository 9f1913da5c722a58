#!/bin/bash
# checks bench.py's live counter leg (roofline.traffic measured in the run) and what rocprofv3 leaves in a child's environment
cd /root/repo
( time python bench.py --no-cpu --no-dropin --steps 60 --warmup 12 ) > gpurun_out/bench_livepmc.json 2> gpurun_out/bench_livepmc.err
cd /tmp
TMPDIR=/tmp rocprofv3 --kernel-trace --output-format csv -d /tmp/x -o t -- python -c "import os; print([k for k in os.environ if k.upper().startswith('ROCP')])" > /root/repo/gpurun_out/env_under_rocprof.txt 2>&1
