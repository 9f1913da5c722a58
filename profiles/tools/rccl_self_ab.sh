# One rank, slab path, REAL RCCL communicator (every exchange an ncclSend / ncclRecv to the rank itself: a device copy made by RCCL's
# kernel on the communication stream), four k-chunks: wall time per substep with the exchanges beside compute + the pipelined sweep
# against everything in line.  The only setting on one GPU in which an exchange costs time that overlap can win back.
for g in "512 256 256" "1024 512 512"; do
for mode in overlap inline; do
  if [ $mode = inline ]; then export UDC_HALO_OVERLAP=0 UDC_MOM_PIPE=0; else unset UDC_HALO_OVERLAP UDC_MOM_PIPE; fi
  for i in 1 2; do
  echo "$g $mode: $(UDC_FORCE_SLAB=1 UDC_FORCE_COMM=1 UDC_A2A_CHUNKS=4 PIPE_TRACE_SUBSTEPS=12 python profiles/tools/pipe_trace_run.py $g 2>/dev/null | grep ms_per_substep)" >> gpurun_out/rccl_self_ab.txt
  done
done
done
