mkdir -p gpurun_out/r4u
timeout -k 10 400 python -m pytest tests/test_gpu_full_size.py -m gpu -x -q > gpurun_out/r4u/pytest.txt 2>&1; tail -1 gpurun_out/r4u/pytest.txt
for v in 1 0 1 0; do
  DFL_SPMV_X4=$v timeout -k 10 300 python bench.py --steps 10 --warmup 3 --cpu-M 0 --cpu-single-M 0 --coupled-M 0 --dem-particles 0 --solve-to-rtol 0 > gpurun_out/r4u/b.json 2> gpurun_out/r4u/b.err || exit 1
  python - gpurun_out/r4u/b.json $v <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); k = d["kernels"]
print("DFL_SPMV_X4=%s: %.2f ms/step  spmv %.4f  pc_apply %.4f  cgs %.4f  b2b %.4f" % (sys.argv[2], d["ms_per_step"], k["spmv"]["avg_ms"], k["pc_apply"]["avg_ms"], k["cgs"]["avg_ms"], d["spmv_back_to_back_ms"]))
print("   " + d["spmv_placement_calibration"][0][60:230])
print("   " + d["spmv_placement_calibration"][-1][:200])
PY
done
