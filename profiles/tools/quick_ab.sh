#!/bin/bash
# quick A/B on the GPU box: kernel-trace stats + FETCH_SIZE / WRITE_SIZE passes of one short bench run, then the bench line.
#   [ENV=...] bash profiles/tools/quick_ab.sh <tag> [bench.py flags ...]
# -> gpurun_out/prof_<tag>/{kernel_stats.csv,pmc_summary.json}, gpurun_out/bench_<tag>.json
set -u
TAG=$1; shift
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
CMD="python $ROOT/bench.py --no-cpu --no-dropin --no-pmc --steps 12 --warmup 3 $*"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o stats -- $CMD > "$OUT/stats.log" 2>&1
i=0
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --output-format csv --pmc $set -d "$OUT/pmc$i" -o pmc -- $CMD > "$OUT/pmc$i.log" 2>&1
done
cd "$ROOT"
python profiles/tools/summarise.py "$OUT" > /dev/null
python bench.py --no-cpu --no-dropin --no-pmc --steps 60 --warmup 12 $* > gpurun_out/bench_$TAG.json 2>gpurun_out/bench_$TAG.err
python - "$TAG" <<'PY'
import csv, json, sys
tag = sys.argv[1]
d = json.load(open(f"gpurun_out/bench_{tag}.json"))
print(tag, "ms_per_step", round(d["ms_per_step"], 4))
pm = json.load(open(f"gpurun_out/prof_{tag}/pmc_summary.json"))
for r in csv.DictReader(open(f"gpurun_out/prof_{tag}/kernel_stats.csv")):
    k = r["kernel"]
    if float(r["total_us"]) < 200: continue
    p = pm.get(k, {})
    print(f"  {k[:58]:58s} n={r['calls']:>4s} avg {float(r['avg_us']):8.1f} us  rd {p.get('hbm_read_bytes', 0) / 1e6:8.1f} MB wr {p.get('hbm_write_bytes', 0) / 1e6:8.1f} MB")
PY
