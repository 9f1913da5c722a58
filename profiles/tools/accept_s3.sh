#!/bin/bash
# acceptance run with the final build of session 3: 100 steps at 256^3 against the reference's own Fortran (neutral + all physics); 512^3 cube-array bench line
cd /root/repo
python bench.py --size 512x512x512 --ibm --steps 12 --warmup 6 --no-cpu --no-dropin 2>/dev/null | tail -1 > gpurun_out/bench_512cube_ibm_s3.json
UDC_LONG_SIZE=256 timeout 1500 python -m pytest tests/test_gpu_long.py -x -q -m gpu -s 2>&1 | tail -6 > gpurun_out/long_parity_256_session3.txt
