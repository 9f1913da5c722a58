#!/bin/bash
# rocprofv3 kernel trace of the thin slab (one rank of eight of configs[3] on the slab code path) and one substep of it as a timeline
#   bash profiles/tools/slab_timeline.sh <tag> [size]
TAG=${1:-tl}
SIZE=${2:-1024x64x512}
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/timeline_$TAG
mkdir -p $OUT
cd /tmp
UDC_FORCE_SLAB=1 rocprofv3 --kernel-trace --output-format csv -d $OUT -o tr -- python $ROOT/bench.py --no-cpu --no-dropin --no-pmc --size $SIZE --steps 30 --warmup 6 > $OUT/bench.json 2> $OUT/bench.err
cd $ROOT
python profiles/tools/timeline.py $OUT 25 > gpurun_out/timeline_$TAG.txt 2>&1
rm -rf $OUT/*/*_agent_info.csv
# keep the trace small: only the last 400 rows travel back
for f in $(find $OUT -name "*kernel_trace.csv"); do (head -1 $f; tail -400 $f) > $f.tail; rm $f; done
