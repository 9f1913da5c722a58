# Round 3, second session: kept profiles of the final build (TileGrid row subsets in every kernel's arguments, the kappa kernel's
# single extra-face pass): rocprofv3 kernel stats + PMC passes for the 256^3 default workload and configs[2], their bench lines,
# and the default bench line with its drop-in and CPU legs.
bash profiles/tools/collect.sh r03b_256
bash profiles/tools/collect.sh r03b_c2 512x512x256/smag/nsv1 67108864 --size 512x512x256 --sgs smag --nsv 1
python bench.py --size 512x512x256 --sgs smag --nsv 1 --steps 20 --warmup 6 --no-dropin --no-cpu --no-pmc 2>/dev/null | tail -1 > gpurun_out/bench_c2_r03b.json
python bench.py --size 1024x512x512 --steps 12 --warmup 6 --no-dropin --no-cpu --no-pmc 2>/dev/null | tail -1 > gpurun_out/bench_1024_r03b.json
python bench.py 2>gpurun_out/bench_default_r03b.err | tail -1 > gpurun_out/bench_default_r03b.json
