#!/bin/bash
# rows per workgroup of fftx_fwd_nat_kernel at the longer lines
cd /root/repo
for L in 2 4 8; do
  UDC_NAT_L=$L python bench.py --size 1024x512x512 --steps 6 --warmup 6 --no-dropin --no-cpu --no-pmc 2>/dev/null | tail -1 > gpurun_out/natl_1024_L$L.json
  UDC_NAT_L=$L python bench.py --size 512x512x256 --sgs smag --nsv 1 --steps 20 --warmup 6 --no-dropin --no-cpu --no-pmc 2>/dev/null | tail -1 > gpurun_out/natl_c2_L$L.json
done
