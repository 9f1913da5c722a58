# spectral row pitch: nkx = nx/2 + 1 padded to a multiple of 8 (default) against unpadded (UDC_SPEC_PAD=0) and padded by 1
for pad in default 0 1; do
  if [ $pad = default ]; then unset UDC_SPEC_PAD; else export UDC_SPEC_PAD=$pad; fi
  python bench.py --steps 40 --warmup 9 --no-cpu --no-pmc --no-dropin 2>/dev/null | tail -1 > gpurun_out/specpad_$pad.json
done
