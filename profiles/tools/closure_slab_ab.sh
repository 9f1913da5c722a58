run() { tag=$1; shift; env "$@" python bench.py --no-cpu --no-dropin --no-pmc --size 1024x64x512 --steps 60 --warmup 12 > gpurun_out/ab_$tag.json 2>gpurun_out/ab_$tag.err; python - $tag <<'PY'
import json,sys
d=json.load(open(f"gpurun_out/ab_{sys.argv[1]}.json"))
print(sys.argv[1], "ms", round(d["ms_per_step"],4), {k[:12]:round(v["avg_ms_net"],4) for k,v in d["kernels"].items() if v["avg_ms_net"]>0.003})
PY
}
run single A=1
run single_nofold UDC_NO_FOLD=1
run slab UDC_FORCE_SLAB=1
run slab_ekalways UDC_FORCE_SLAB=1 UDC_EK_ALWAYS=1
