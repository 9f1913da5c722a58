#!/bin/bash
# slab-path A/B on one GPU: per-kernel times of the bench line under a few switch settings
export UDC_FORCE_SLAB=1
run() { tag=$1; shift; env "$@" python bench.py --no-cpu --no-dropin --no-pmc --size 1024x64x512 --steps 60 --warmup 12 > gpurun_out/ab_$tag.json 2>gpurun_out/ab_$tag.err; python - $tag <<'PY'
import json,sys
d=json.load(open(f"gpurun_out/ab_{sys.argv[1]}.json"))
print(sys.argv[1], "ms", round(d["ms_per_step"],4), {k:round(v["avg_ms_net"],4) for k,v in d["kernels"].items() if v["avg_ms_net"]>0.003})
PY
}
run default A=1
run p_own_exchange UDC_P_TRANSPOSE=0
run nooverlap UDC_HALO_OVERLAP=0
run nopipe UDC_MOM_PIPE=0
run pipe_row0first UDC_MOM_PIPE=1
run chunks2 UDC_A2A_CHUNKS=2
run chunks1 UDC_A2A_CHUNKS=1
run default2 A=1
