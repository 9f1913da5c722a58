# Evidence that the exchanges run BESIDE compute: rocprofv3 kernel trace of the forced-slab substep (one GPU, exchange = device copy
# onto itself) at 1024x512x512 with four k-chunks; profiles/tools/overlap_trace.py then reports, for every pack / unpack kernel and
# every exchange copy, which compute kernel's interval it falls into.
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/overlap_trace
mkdir -p $OUT
cd /tmp
UDC_FORCE_SLAB=1 UDC_A2A_CHUNKS=4 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT -o tr -- python $ROOT/bench.py --size 1024x512x512 --steps 3 --warmup 0 --no-cpu --no-pmc --no-dropin --no-single > $OUT/run.log 2>&1
cd $ROOT
python profiles/tools/overlap_trace.py $OUT > gpurun_out/overlap_trace_summary.txt 2>&1
