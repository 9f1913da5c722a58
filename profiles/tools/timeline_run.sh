#!/bin/bash
# rocprofv3 kernel trace of a bench.py run and one substep of it as a timeline (profiles/tools/timeline.py)
#   [ENV=..] bash profiles/tools/timeline_run.sh <tag> <first-kernel substring> [bench flags ...]
TAG=$1; FIRST=$2; shift 2
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/timeline_$TAG
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -o tr -- python $ROOT/bench.py --no-cpu --no-dropin --no-pmc --steps 30 --warmup 6 "$@" > $OUT/bench.json 2> $OUT/bench.err
cd $ROOT
python profiles/tools/timeline.py $OUT 25 $FIRST > gpurun_out/timeline_$TAG.txt 2>&1
for f in $(find $OUT -name "*kernel_trace.csv"); do rm $f; done
rm -f $OUT/*agent_info.csv
