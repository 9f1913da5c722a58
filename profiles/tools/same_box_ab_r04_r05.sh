#!/bin/bash
# Same-box A/B of the round-4 tree (git worktree of f854f70 under _r04tree/, built there) against the current tree: boxes differ by
# +-2 %, more than some of the changes.  Alternating runs, ms per substep.
run() {   # tree, label, flags...
  local tree=$1 label=$2; shift; shift
  (cd $tree && python bench.py --no-cpu --no-dropin --no-pmc "$@" 2>/dev/null) | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['kernels']
print('$label', round(d['ms_per_step'],4), {n:round(v['avg_ms'],4) for n,v in k.items() if n in ('unpack_fftx_bwd','thomas','bottom','mom_truetruetruetrue','fft_bwd')})
"
}
for rep in 1 2 3; do
  run _r04tree "256^3        r04" --steps 300 --warmup 30
  run .        "256^3        r05" --steps 300 --warmup 30
done
export UDC_FORCE_SLAB=1
for rep in 1 2 3; do
  run _r04tree "1024x64x512  r04" --size 1024x64x512 --steps 90 --warmup 12
  run .        "1024x64x512  r05" --size 1024x64x512 --steps 90 --warmup 12
done
for rep in 1 2; do
  run _r04tree "128x512x512  r04" --size 128x512x512 --steps 90 --warmup 12
  run .        "128x512x512  r05" --size 128x512x512 --steps 90 --warmup 12
done
