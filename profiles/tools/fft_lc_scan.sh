for L in 2 4 8; do for C in 4 8 16; do
  UDC_FORCE_SLAB=1 UDC_FFT_L=$L UDC_FFT_C=$C python bench.py --size 1024x512x512 --steps 3 --warmup 6 --no-dropin --no-cpu --no-pmc 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; print('L=$L C=$C', round(d['ms_per_step'],3), {n:round(v['avg_ms'],3) for n,v in k.items() if 'fft' in n})"
done; done
