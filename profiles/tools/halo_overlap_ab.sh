python -m pytest tests/test_gpu_slabs.py tests/test_mpi_route.py tests/test_gpu_large.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/ov_tests.log
for ov in 1 0; do
  UDC_HALO_OVERLAP=$ov UDC_FORCE_SLAB=1 python bench.py --steps 30 --warmup 9 --no-cpu --no-pmc --no-dropin > gpurun_out/ov_256_$ov.json 2> gpurun_out/ov_256_$ov.err
  UDC_HALO_OVERLAP=$ov UDC_FORCE_SLAB=1 python bench.py --steps 12 --warmup 6 --no-cpu --no-pmc --no-dropin --size 1024x512x512 > gpurun_out/ov_1024_$ov.json 2> gpurun_out/ov_1024_$ov.err
done
