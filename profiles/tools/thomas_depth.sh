for d in 1 2 3 4 6; do
  UDC_THOMAS_DEPTH=$d python bench.py --no-cpu --steps 150 --warmup 20 2>/dev/null > gpurun_out/td_$d.json
  python - $d <<'PY'
import json,sys
d=json.loads(open(f"gpurun_out/td_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print("depth", sys.argv[1], round(d["ms_per_step"],4), d["kernels"]["thomas"]["avg_ms"])
PY
done
