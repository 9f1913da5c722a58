# kappa scalar kernel A/B (UDC_KAPPA_W5: a fifth wave for the tile's extra faces): parity tests that run it, then the configs[2]
# bench line (512x512x256, Smagorinsky, one kappa scalar)
for w in 1 0; do
UDC_KAPPA_W5=$w python -m pytest tests/test_gpu_parity.py tests/test_gpu_large.py -m gpu -x -q -k "scalar or thlk or kappa or 512" 2>&1 | tail -2 > gpurun_out/kappa_tests_$w.log
for i in 1 2; do
UDC_KAPPA_W5=$w python bench.py --size 512x512x256 --sgs smag --nsv 1 --steps 30 --warmup 9 --no-dropin --no-cpu --no-pmc 2>/dev/null | tail -1 > gpurun_out/kappa_bench_${w}_$i.json
done
done
