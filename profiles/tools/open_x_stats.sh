#!/bin/bash
# rocprofv3 --kernel-trace --stats of the substep with open x boundaries (profiles/tools/open_x_rate.py, 256^3) -> gpurun_out/prof_open_x
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/prof_open_x
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_open_x -o open_x -- python $R/profiles/tools/open_x_rate.py 256 > $R/gpurun_out/prof_open_x/run.log 2>&1
f=$(find $R/gpurun_out/prof_open_x -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" $R/gpurun_out/open_x_kernel_stats_256.csv && head -25 "$f"
