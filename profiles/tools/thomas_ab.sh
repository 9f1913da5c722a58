#!/bin/bash
# Thomas variants A/B: per-kernel survey of bench.py (HIP events around every launch) at 256^3 and 1024x512x512.
# usage (GPU box): bash profiles/tools/thomas_ab.sh > gpurun_out/thomas_ab.txt
run() {   # label, size, env...
  local label=$1 size=$2; shift 2
  local out
  out=$(env "$@" python bench.py --no-cpu --no-pmc --no-dropin --steps 30 --warmup 15 --size $size 2>/dev/null | tail -1)
  python - "$label" "$size" "$out" <<'PY'
import json, sys
d = json.loads(sys.argv[3])
k = d["kernels"]["thomas"]
print(f"{sys.argv[1]:28s} {sys.argv[2]:14s} thomas {k['avg_ms_net']*1e3:8.1f} us  frac {k.get('frac')}  substep {d['ms_per_step']:.4f} ms  divmax {d['divmax_after_run']:.2e}")
PY
}
for size in 256x256x256 512x512x256 1024x512x512; do
  run "lds (r03 default)" $size UDC_THOMAS=3
  run "stream" $size UDC_THOMAS=0
  for w in 2 3 4; do run "reg sl8 w$w" $size UDC_THOMAS=4 UDC_THOMAS_W=$w; done
  run "reg sl16" $size UDC_THOMAS=4 UDC_THOMAS_SL=16
done
