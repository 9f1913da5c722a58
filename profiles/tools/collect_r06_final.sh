#!/bin/bash
# final collection of round 6's last session: configs[1] (rocprofv3 stats + counters, the full default bench line) and the open-x substep's kernel table
bash profiles/tools/collect.sh r06_c1 256x256x256/vreman/nsv0 16777216
python bench.py > gpurun_out/bench_256cube_default_r06.json 2>gpurun_out/bench_256cube_default_r06.err
UDC_FORCE_SLAB=1 python bench.py --no-cpu --no-dropin --no-pmc --size 1024x64x512 --steps 60 --warmup 12 > gpurun_out/bench_r06_1024x64x512.json 2>/dev/null
bash profiles/tools/open_x_stats.sh > gpurun_out/open_x_stats.log 2>&1
ls gpurun_out/prof_r06_c1/ | head
tail -c 600 gpurun_out/bench_256cube_default_r06.json
