# rocprofv3 kernel-trace statistics of the default bench workload over a run long enough for steady clocks (the 15-substep passes
# of collect.sh catch the ramp: mom_lds_kernel 255 .. 356 us in one trace): 120 substeps after 30 of warm-up.
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_r03_256_long; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python $ROOT/bench.py --no-cpu --no-pmc --no-dropin --steps 120 --warmup 30 > $OUT/stats.log 2>&1
grep "^{\"metric\"" $OUT/stats.log | tail -1 > $OUT/bench_line.json
cd $ROOT; python - <<PY
import csv, json
rows = list(csv.DictReader(open("$OUT/stats/stats_kernel_stats.csv")))
keep = [r for r in rows]
with open("$OUT/kernel_stats.csv", "w") as f:
    f.write("kernel,calls,total_us,avg_us,min_us,max_us,percent\n")
    for r in keep:
        f.write('"%s",%s,%.1f,%.2f,%.2f,%.2f,%s\n' % (r["Name"][:90], r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
d = json.loads(open("$OUT/bench_line.json").read())
print("bench line under rocprofv3:", d["ms_per_step"], d["roofline"]["avg_launch_ms"])
PY
head -8 $OUT/kernel_stats.csv
