# the 1024 x 512 x 512 transposes through a real one-rank RCCL communicator in 1, 2, 4 k-chunks (blocks of 2.1, 1.1, 0.5 GB per operation
# before comm_alltoall cut them into pieces of 512 MiB)
for n in 1 2 4; do
  echo "1024 512 512 chunks $n: $(UDC_FORCE_SLAB=1 UDC_FORCE_COMM=1 UDC_A2A_CHUNKS=$n PIPE_TRACE_SUBSTEPS=3 python profiles/tools/pipe_trace_run.py 1024 512 512 2>&1 | grep -a "ms_per_substep\|rror" | tail -2)" >> gpurun_out/dbg_chunks.txt
done
