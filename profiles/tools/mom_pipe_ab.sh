# the momentum sweep pipelined with the slab solve's k-chunks (UDC_MOM_PIPE, default on) against the plain order, forced slab on one
# GPU with four k-chunks (UDC_A2A_CHUNKS=4; the exchange is the identity there, so only the price of the extra launches shows)
for mp in 1 0; do
  UDC_MOM_PIPE=$mp UDC_A2A_CHUNKS=4 UDC_FORCE_SLAB=1 python bench.py --steps 30 --warmup 9 --no-cpu --no-pmc --no-dropin > gpurun_out/mp_256_$mp.json 2> gpurun_out/mp_256_$mp.err
  UDC_MOM_PIPE=$mp UDC_A2A_CHUNKS=4 UDC_FORCE_SLAB=1 python bench.py --steps 12 --warmup 6 --no-cpu --no-pmc --no-dropin --size 1024x512x512 > gpurun_out/mp_1024_$mp.json 2> gpurun_out/mp_1024_$mp.err
done
