#!/bin/bash
# the pressure-total form of the fused substep against the reference's form (UDC_PTOTAL=1 / 0), alternating, one box
for wl in "--steps 150 --warmup 12" "--size 512x512x256 --sgs smag --nsv 1 --steps 60 --warmup 9" "--size 1024x512x512 --steps 30 --warmup 6"; do
  echo "== $wl"
  for v in 0 1 0 1; do UDC_PTOTAL=$v python bench.py --no-cpu --no-dropin --no-pmc $wl 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; print('ptotal=$v', round(d['ms_per_step'],4), {n[:8]:round(x.get('avg_ms_net',x['avg_ms']),4) for n,x in k.items() if n[:3] in ('mom','pro')}, 'divmax', d['divmax_after_run'])"; done
done
for wl in "--size 1024x64x512 --steps 60 --warmup 9" "--size 1024x128x512 --steps 40 --warmup 9"; do
  echo "== UDC_FORCE_SLAB=1 $wl"
  for v in 0 1 0 1; do UDC_FORCE_SLAB=1 UDC_PTOTAL=$v python bench.py --no-cpu --no-dropin --no-pmc $wl 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; print('ptotal=$v', round(d['ms_per_step'],4), {n[:8]:round(x.get('avg_ms_net',x['avg_ms']),4) for n,x in k.items() if n[:3] in ('mom','pro')}, 'divmax', d['divmax_after_run'])"; done
done
