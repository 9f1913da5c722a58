# Round 3, second session: eight virtual ranks at 1024x512x512 with the ghost-row exchanges beside the sweeps (default) and in line
# (UDC_HALO_OVERLAP=0); the configs[4]-style cube-array bench line (512^3, immersed boundary + wall functions).
python profiles/tools/virtual_ranks.py 8 > gpurun_out/virtual_ranks_1024_overlap.json 2> gpurun_out/vr1.err
UDC_HALO_OVERLAP=0 python profiles/tools/virtual_ranks.py 8 > gpurun_out/virtual_ranks_1024_inline.json 2> gpurun_out/vr0.err
python bench.py --size 512x512x512 --ibm --steps 12 --warmup 6 --no-cpu --no-pmc --no-dropin 2> gpurun_out/bench_ibm512.err | tail -1 > gpurun_out/bench_512cube_ibm_session2.json
