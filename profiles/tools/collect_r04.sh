#!/bin/bash
# round 4: rocprofv3 --kernel-trace --stats and the counter passes of the three single-GPU workloads (BASELINE configs[1], [2], and
# configs[3]'s grid on one GPU), then the kept bench lines of the same workloads
bash profiles/tools/collect.sh r04_c1 256x256x256/vreman/nsv0 16777216
bash profiles/tools/collect.sh r04_c2 512x512x256/smag/nsv1 67108864 --size 512x512x256 --sgs smag --nsv 1
bash profiles/tools/collect.sh r04_c3 1024x512x512/vreman/nsv0 268435456 --size 1024x512x512 --steps 4
python bench.py --no-cpu --no-dropin --size 512x512x256 --sgs smag --nsv 1 --steps 60 --warmup 12 > gpurun_out/bench_512x512x256_smag_nsv1_r04.json 2>/dev/null
python bench.py --no-cpu --no-dropin --size 1024x512x512 --steps 30 --warmup 9 > gpurun_out/bench_1024_r04.json 2>/dev/null
UDC_FORCE_SLAB=1 python bench.py --no-cpu --no-dropin --no-pmc --size 1024x512x512 --steps 30 --warmup 9 > gpurun_out/bench_1024_forced_slab_r04.json 2>/dev/null
UDC_FORCE_SLAB=1 python bench.py --no-cpu --no-dropin --no-pmc --steps 150 --warmup 15 > gpurun_out/bench_256cube_forced_slab_r04.json 2>/dev/null
python bench.py --no-cpu --no-dropin --no-pmc --ibm --size 512x512x512 --steps 30 --warmup 9 > gpurun_out/bench_512cube_ibm_r04.json 2>/dev/null
ls gpurun_out/prof_r04_c*/
