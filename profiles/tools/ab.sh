for rep in 1 2; do for v in base new; do
  if [ $v = base ]; then export UDC_LIBPATH=$GRAFT_REPO_ROOT/u-dales_amd/lib/libudcore_base.so; else unset UDC_LIBPATH; fi
  python bench.py $BARGS --no-dropin --no-cpu --no-pmc 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; print('$v', round(d['ms_per_step'],4), {n[:7]:round(v['avg_ms'],4) for n,v in k.items() if n[:3] in ('mom','clo','pro','tho','sca')})"
done; done
