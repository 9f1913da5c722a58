# usage: VAR=NAME VALUES="a b c" BARGS="..." bash profiles/tools/envscan.sh   -- one bench line per value of an env switch
for v in $VALUES; do
  env $VAR=$v python bench.py $BARGS --no-dropin --no-cpu --no-pmc 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; print('$VAR=$v', round(d['ms_per_step'],4), {n[:7]:round(v['avg_ms'],4) for n,v in k.items() if n[:3] in ('mom','clo','pro','tho','sca','div','fft')})"
done
