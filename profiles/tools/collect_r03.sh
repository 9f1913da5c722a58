# Round-3 kept profiles (run through gpurun from the repo root): rocprofv3 kernel stats + PMC passes for the three single-GPU
# workloads, the forced-slab bench lines, and the default bench line with its drop-in and CPU legs.
bash profiles/tools/collect.sh r03_256
bash profiles/tools/collect.sh r03_c2 512x512x256/smag/nsv1 67108864 --size 512x512x256 --sgs smag --nsv 1
bash profiles/tools/collect.sh r03_c3 1024x512x512/vreman/nsv0 268435456 --size 1024x512x512 --steps 4
python bench.py --size 512x512x256 --sgs smag --nsv 1 --steps 20 --warmup 6 --no-dropin --no-cpu --no-pmc 2>/dev/null | tail -1 > gpurun_out/bench_c2_r03.json
UDC_FORCE_SLAB=1 python bench.py --steps 60 --warmup 10 --no-dropin --no-cpu --no-pmc 2>/dev/null | tail -1 > gpurun_out/bench_256_forced_slab_r03.json
python bench.py 2>gpurun_out/bench_default_r03.err | tail -1 > gpurun_out/bench_default_r03.json
