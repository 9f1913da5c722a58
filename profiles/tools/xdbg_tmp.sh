timeout 900 python -m pytest tests/test_gpu_slabs.py -x -q 2>&1 | tail -3
export UDC_FORCE_SLAB=1
for sz in 1024x64x512 256x256x256; do
for W in 100000 1024 768 512; do
for L in 4 8; do
UDC_R8_L=$L UDC_R8_WG=$W python bench.py --no-cpu --no-dropin --no-pmc --size $sz --steps 30 --warmup 6 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$sz L=$L WG=$W', round(d['ms_per_step'],4), 'xbwd', round(d['kernels']['unpack_fftx_bwd']['avg_ms'],4), 'div', d['divmax_after_run'])
"
done; done; done
