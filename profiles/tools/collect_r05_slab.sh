#!/bin/bash
# round 5: the slab (multi-GPU) code path at the per-rank sizes of configs[3] on 8 / 4 / 2 GPUs, one rank as a whole domain
# (UDC_FORCE_SLAB=1): rocprofv3 --kernel-trace --stats + FETCH / WRITE counter passes + counters, then the bench lines.
#   bash profiles/tools/collect_r05_slab.sh <tag>
TAG=${1:-r05_slab}
export UDC_FORCE_SLAB=1
bash profiles/tools/collect.sh ${TAG}_1024x64x512 1024x64x512/vreman/nsv0 33554432 --size 1024x64x512
for s in 1024x64x512 1024x128x512 1024x256x512; do
  python bench.py --no-cpu --no-dropin --no-pmc --size $s --steps 60 --warmup 12 > gpurun_out/bench_${TAG}_$s.json 2>gpurun_out/bench_${TAG}_$s.err
done
