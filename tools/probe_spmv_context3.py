"""Probe: cost of the SpMV inside an emulated Arnoldi step (grouped timing, no instrumentation inside a group):
   T[pc, spmv, dots, update] - T[pc, dots, update]  for a basis of k columns."""
import sys, os, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dedflow_amd import api
from dedflow_amd.meshgen import kuhn_cube, synthetic_fields
M = int(sys.argv[1]) if len(sys.argv) > 1 else 119
mesh = kuhn_cube(M, jitter=0.2)
wg, dwg = synthetic_fields(mesh)
P = api.Problem(mesh)
L = api.lib()
wg_d, dwg_d = api.DeviceArray.from_numpy(wg), api.DeviceArray.from_numpy(dwg)
P.assemble_system(wg_d, dwg_d, None, want_J=True)
N = P.N; n4 = 4 * N; KMAX = 41
Q = api.DeviceArray(n4 * KMAX)
tmp = api.DeviceArray(12 * N)
h = api.DeviceArray(256); nrm = api.DeviceArray(8)
L.dfl_cgs_work_size.restype = C.c_int64
work = api.DeviceArray(int(L.dfl_cgs_work_size(n4, KMAX)) + 16)
d33 = api.DeviceArray(9 * N); d1 = api.DeviceArray(N)
vp, i32, i64, f64 = C.c_void_p, C.c_int32, C.c_int64, C.c_double
L.dfl_pc_jacobi_setup.argtypes = [i32, vp, vp, vp, vp, vp, vp]
L.dfl_pc_jacobi_apply.argtypes = [i32, i32, vp, vp, vp, vp, vp]
L.dfl_cgs_dots.argtypes = [i32, i32, vp, i64, vp, vp, vp, vp]
L.dfl_cgs_update.argtypes = [i32, i32, vp, i64, vp, vp, vp, C.c_int, vp, vp]
L.dfl_bcsr_spmv.argtypes = [i32, vp, vp, vp, f64, vp, f64, vp, vp]
rp_h, ci_h = P.pattern()
rp = api.DeviceArray.from_numpy(rp_h.astype(np.int32)); ci = api.DeviceArray.from_numpy(ci_h.astype(np.int32))
val = L.MatrixFSBlockValues(P.J)
L.dfl_pc_jacobi_setup(N, rp.ptr, ci.ptr, val, d33.ptr, d1.ptr, None)
Q.upload(np.random.default_rng(0).normal(size=n4 * KMAX) * 1e-3)
col = lambda k: Q.ptr + 8 * n4 * k
def step(k, with_spmv, with_rest=True):
    if with_rest: L.dfl_pc_jacobi_apply(N, n4, d33.ptr, d1.ptr, col(k), tmp.ptr, None)
    if with_spmv: L.dfl_bcsr_spmv(N, rp.ptr, ci.ptr, val, 1.0, tmp.ptr, 0.0, col(k + 1), None)
    if with_rest:
        L.dfl_cgs_dots(n4, k + 1, Q.ptr, n4, col(k + 1), h.ptr, work.ptr, None)
        L.dfl_cgs_update(n4, k + 1, Q.ptr, n4, h.ptr, col(k + 1), nrm.ptr, 1, work.ptr, None)
def group(k, with_spmv, with_rest=True, reps=8):
    t = api.Timer(); res = []
    for r in range(4):
        step(k, with_spmv, with_rest)
        t.start()
        for _ in range(reps): step(k, with_spmv, with_rest)
        t.stop(); res.append(t.ms() / reps)
    return min(res)
print("SpMV alone, grouped: %.4f ms" % group(5, True, False))
for k in (0, 5, 20, 39):
    a, b = group(k, True), group(k, False)
    print("k=%2d: step with SpMV %.4f ms, without %.4f ms -> SpMV costs %.4f ms" % (k, a, b, a - b), flush=True)
