#!/bin/bash
# The interleaved matvec under the library's DEFAULT (bounded) work-space pick, i.e. without the explicit calibration
OUT=${1:-gpurun_out/ab_x4_default}; mkdir -p $OUT
for v in 1 0 1 0 1 0; do
  DFL_WS_VERBOSE=1 DFL_SPMV_X4=$v timeout -k 10 300 python bench.py --steps 10 --warmup 3 --cpu-M 0 --cpu-single-M 0 --solve-to-rtol 0 --coupled-M 0 --dem-particles 0 --placement default > $OUT/b.json 2> $OUT/b.err || exit 1
  python - $OUT/b.json $v <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); k = d["kernels"]
print("DFL_SPMV_X4=%s default placement: %.2f ms/step, spmv %.4f ms | %s" % (sys.argv[2], d["ms_per_step"], k["spmv"]["avg_ms"], (d["spmv_placement_calibration"] or [""])[0][40:200]))
PY
done
