#!/bin/bash
# round 6, final collection (run through gpurun from the repo root): the slab (multi-GPU) code path at one rank's size of configs[3] on
# 8 / 4 / 2 GPUs -- rocprofv3 --kernel-trace --stats + counter passes of 1024x64x512 and the bench lines of all three -- then configs[1]
# (stats + counters + the full default bench line) and the bench lines of configs[2], configs[3]'s grid on one GPU (single slab and
# forced slab) and configs[4]'s grid.
sed 's/r05/r06/g' profiles/tools/collect_r05_slab.sh > /tmp/collect_r06_slab.sh
bash /tmp/collect_r06_slab.sh r06
bash profiles/tools/collect.sh r06_c1 256x256x256/vreman/nsv0 16777216
python bench.py > gpurun_out/bench_256cube_default_r06.json 2>gpurun_out/bench_256cube_default_r06.err
python bench.py --no-cpu --no-dropin --size 512x512x256 --sgs smag --nsv 1 --steps 60 --warmup 12 > gpurun_out/bench_512x512x256_smag_nsv1_r06.json 2>/dev/null
python bench.py --no-cpu --no-dropin --size 1024x512x512 --steps 30 --warmup 9 > gpurun_out/bench_1024_r06.json 2>/dev/null
UDC_FORCE_SLAB=1 python bench.py --no-cpu --no-dropin --no-pmc --size 1024x512x512 --steps 30 --warmup 9 > gpurun_out/bench_1024_forced_slab_r06.json 2>/dev/null
python bench.py --no-cpu --no-dropin --no-pmc --ibm --size 512x512x512 --steps 30 --warmup 9 > gpurun_out/bench_512cube_ibm_r06.json 2>/dev/null
ls gpurun_out/prof_r06*/
