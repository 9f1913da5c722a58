#!/bin/bash
# A/B of library variants (profiles/tools/mkvariant.sh) inside one gpurun call: every variant on every workload, twice, alternating.
#   VARIANTS="base new w4" KEYS="tho mom" bash profiles/tools/variants_ab.sh "<bench flags of workload 1>" "<... workload 2>" ...
#   a workload string may start with ENV=VALUE words (e.g. "UDC_FORCE_SLAB=1 --size 1024x64x512")
KEYS=${KEYS:-tho}
for wl in "$@"; do
  envs=""; flags=""
  for w in $wl; do case $w in [A-Z]*=*) envs="$envs $w";; *) flags="$flags $w";; esac; done
  echo "== $wl"
  for rep in 1 2; do for v in $VARIANTS; do
    if [ $v = default ]; then LP=""; else LP="UDC_LIBPATH=$GRAFT_REPO_ROOT/u-dales_amd/lib/libudcore_$v.so"; fi
    env $envs $LP python bench.py $flags --no-dropin --no-cpu --no-pmc 2>/dev/null | tail -1 | KEYS="$KEYS" python -c "
import json,sys,os; d=json.loads(sys.stdin.read()); k=d['kernels']; ks=os.environ['KEYS'].split()
print('$v', round(d['ms_per_step'],4), {n[:12]:round(v.get('avg_ms_net',v['avg_ms']),4) for n,v in k.items() if any(n.startswith(q) for q in ks)})"
  done; done
done
