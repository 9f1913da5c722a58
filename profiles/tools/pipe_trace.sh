# RCCL's own kernels against the compute stream's, by rocprofv3 time stamps (see pipe_trace_run.py)
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/pipe_trace
mkdir -p $OUT
cd /tmp
UDC_FORCE_SLAB=1 UDC_FORCE_COMM=1 UDC_A2A_CHUNKS=4 rocprofv3 --kernel-trace --output-format csv -d $OUT -o tr -- python $ROOT/profiles/tools/pipe_trace_run.py > $OUT/run.log 2>&1
cd $ROOT
python profiles/tools/overlap_trace.py $OUT nccl > gpurun_out/pipe_trace_summary.txt 2>&1
