#!/bin/bash
# A/B of the CGS kernels inside the bench step: dots with all loads of a column tile in flight before the reductions
# (default) against one reduction per column (DFL_CGS_DOTS_TILE=0)
OUT=${1:-gpurun_out/ab_cgs}; mkdir -p $OUT
for v in 1 0 1 0; do
  DFL_CGS_DOTS_TILE=$v timeout -k 10 300 python bench.py --steps 10 --warmup 3 --cpu-M 0 --cpu-single-M 0 --solve-to-rtol 0 --coupled-M 0 --dem-particles 0 --placement default > $OUT/b.json 2> $OUT/b.err || exit 1
  python - $OUT/b.json $v <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); k = d["kernels"]
print("DFL_CGS_DOTS_TILE=%s: %.2f ms/step, spmv %.4f ms, cgs avg %.4f ms (%.2f ms/step)" % (sys.argv[2], d["ms_per_step"], k["spmv"]["avg_ms"], k["cgs"]["avg_ms"], k["cgs"]["total_ms_per_step"]))
PY
done
