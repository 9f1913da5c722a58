#!/bin/bash
# slab path: register y transforms (UDC_SLAB_YREG, default on for ny = 128 / 256 / 512) against the Stockham kernels
cd /root/repo
timeout 1500 python -m pytest tests/test_gpu_slabs.py tests/test_gpu_large.py tests/test_gpu_long.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/slab_yreg_tests.txt
for y in 1 0; do
  UDC_SLAB_YREG=$y UDC_FORCE_SLAB=1 python bench.py --steps 60 --warmup 12 --no-dropin --no-cpu --no-pmc 2>/dev/null | tail -1 > gpurun_out/slab_yreg_256_$y.json
  UDC_SLAB_YREG=$y UDC_FORCE_SLAB=1 python bench.py --size 1024x512x512 --steps 6 --warmup 6 --no-dropin --no-cpu --no-pmc 2>/dev/null | tail -1 > gpurun_out/slab_yreg_1024_$y.json
  UDC_SLAB_YREG=$y UDC_FORCE_SLAB=1 python bench.py --size 128x128x128 --steps 60 --warmup 12 --no-dropin --no-cpu --no-pmc 2>/dev/null | tail -1 > gpurun_out/slab_yreg_128_$y.json
done
