# (depths 3, 4 and 6 were measured with a build that instantiated them; the shipped library keeps 1 and 2)
for d in 1 2; do
  UDC_THOMAS_DEPTH=$d python bench.py --no-cpu --no-pmc --steps 150 --warmup 20 2>/dev/null > gpurun_out/td_$d.json
  python - $d <<'PY'
import json,sys
d=json.loads(open(f"gpurun_out/td_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print("depth", sys.argv[1], round(d["ms_per_step"],4), d["kernels"]["thomas"]["avg_ms"])
PY
done
