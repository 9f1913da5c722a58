# the same one-rank real-RCCL setting: number of k-chunks of the transposes (and of the pipelined sweep)
for n in 1 2 4 8 16; do
  echo "chunks $n: $(UDC_FORCE_SLAB=1 UDC_FORCE_COMM=1 UDC_A2A_CHUNKS=$n PIPE_TRACE_SUBSTEPS=12 python profiles/tools/pipe_trace_run.py 1024 512 512 2>/dev/null | grep ms_per_substep)" >> gpurun_out/rccl_self_chunks.txt
done
