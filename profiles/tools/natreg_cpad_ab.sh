#!/bin/bash
# LDS column pitch of ffty_natreg_kernel: +2 complex (default build) against +0 (libudcore_cpad0.so built with -DNATREG_CPAD=0)
cd /root/repo
timeout 300 python -m pytest tests/test_gpu_own_forward.py -x -q -m gpu 2>&1 | tail -3 > gpurun_out/cpad_tests.txt
for rep in 1 2 3; do
  for lib in "" cpad0; do
    if [ -n "$lib" ]; then export UDC_LIBPATH=$GRAFT_REPO_ROOT/u-dales_amd/lib/libudcore_$lib.so; else unset UDC_LIBPATH; fi
    python bench.py --steps 150 --warmup 12 --no-dropin --no-cpu --no-pmc 2>/dev/null | tail -1 > gpurun_out/cpad_256_lib${lib}_r$rep.json
  done
done
for lib in "" cpad0; do
  if [ -n "$lib" ]; then export UDC_LIBPATH=$GRAFT_REPO_ROOT/u-dales_amd/lib/libudcore_$lib.so; else unset UDC_LIBPATH; fi
  python bench.py --size 1024x512x512 --steps 6 --warmup 6 --no-dropin --no-cpu --no-pmc 2>/dev/null | tail -1 > gpurun_out/cpad_1024_lib${lib}.json
done
