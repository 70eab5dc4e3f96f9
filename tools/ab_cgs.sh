#!/bin/bash
# A/B of the CGS dots kernel's column tile inside the bench step (DFL_CGS_TILE: 8 = round 2, 64 = w read once per row block)
OUT=${1:-gpurun_out/ab_cgs}; mkdir -p $OUT
for v in 64 8 16 64 8; do
  DFL_CGS_TILE=$v timeout -k 10 300 python bench.py --steps 10 --warmup 3 --cpu-M 0 --cpu-single-M 0 --solve-to-rtol 0 --coupled-M 0 --dem-particles 0 --placement default > $OUT/b.json 2> $OUT/b.err || exit 1
  python - $OUT/b.json $v <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); k = d["kernels"]
print("DFL_CGS_TILE=%s: %.2f ms/step, spmv %.4f ms, cgs avg %.4f ms (%.2f ms/step)" % (sys.argv[2], d["ms_per_step"], k["spmv"]["avg_ms"], k["cgs"]["avg_ms"], k["cgs"]["total_ms_per_step"]))
PY
done
