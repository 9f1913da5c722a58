#!/bin/bash
# rocprofv3 --kernel-trace --stats of configs[1] over a run long enough for the clocks to settle (collect.sh profiles 12 substeps):
#   bash profiles/tools/stats_long_r06.sh   -> gpurun_out/prof_r06_c1_long/kernel_stats.csv
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_r06_c1_long; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o stats -- python $ROOT/bench.py --no-cpu --no-dropin --no-pmc --steps 150 --warmup 30 > "$OUT/stats.log" 2>&1
cd $ROOT
python profiles/tools/summarise.py "$OUT" > /dev/null 2>&1
head -12 $OUT/kernel_stats.csv
tail -1 $OUT/stats.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench under rocprofv3: ms_per_step', d['ms_per_step'], 'mom avg_launch_ms', d['roofline']['avg_launch_ms'])"
