#!/bin/bash
# Profile collection on the GPU box (run through gpurun from the repo root):
#   bash profiles/tools/collect.sh <tag> [workload key] [cells per GPU] [extra bench.py flags ...]
# e.g.  bash profiles/tools/collect.sh c3 512x512x256/smag/nsv1 67108864 --size 512x512x256 --sgs smag --nsv 1
# Writes raw rocprofv3 output under gpurun_out/prof_<tag>/ and per-kernel summaries
# gpurun_out/prof_<tag>/{kernel_stats.csv,pmc_summary.json}; copy what should be kept into profiles/rNN/.
# Counters go in their own passes with --kernel-trace only (no sys/hip/hsa tracing next to --pmc).
set -u
TAG=${1:-run}
export WORKLOAD_KEY=${2:-256x256x256/vreman/nsv0}
export CELLS=${3:-16777216}
export TRAFFIC_SOURCE="profiles/tools/collect.sh $TAG"
shift; shift; shift
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
CMD="python $ROOT/bench.py --no-cpu --no-dropin --no-pmc --steps 12 --warmup 3 $*"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o stats -- $CMD > "$OUT/stats.log" 2>&1
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --output-format csv --pmc $set -d "$OUT/pmc$i" -o pmc -- $CMD > "$OUT/pmc$i.log" 2>&1
done
cd "$ROOT"
python profiles/tools/summarise.py "$OUT"
