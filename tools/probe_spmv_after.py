"""Probe: event-timed single SpMV right after (a) another SpMV, (b) a 2.2 GB streaming read (CGS dots over the basis),
(c) the CGS update (streaming read + 55 MB write), (d) the Jacobi apply that writes its input vector."""
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
N = P.N; n4 = 4 * N; K = 40
Q = api.DeviceArray(n4 * (K + 1)); tmp = api.DeviceArray(12 * N)
h = api.DeviceArray(256); nrm = api.DeviceArray(8)
L.dfl_cgs_work_size.restype = C.c_int64
work = api.DeviceArray(int(L.dfl_cgs_work_size(n4, K + 1)) + 16)
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
Q.upload(np.random.default_rng(0).normal(size=n4 * (K + 1)) * 1e-3)
col = lambda k: Q.ptr + 8 * n4 * k
spmv = lambda: L.dfl_bcsr_spmv(N, rp.ptr, ci.ptr, val, 1.0, tmp.ptr, 0.0, col(K), None)
before = {
    "another SpMV": spmv,
    "CGS dots over 39 columns (2.2 GB read)": lambda: L.dfl_cgs_dots(n4, 39, Q.ptr, n4, col(K), h.ptr, work.ptr, None),
    "CGS update over 39 columns": lambda: L.dfl_cgs_update(n4, 39, Q.ptr, n4, h.ptr, col(K), nrm.ptr, 1, work.ptr, None),
    "Jacobi apply (writes the SpMV input)": lambda: L.dfl_pc_jacobi_apply(N, n4, d33.ptr, d1.ptr, col(3), tmp.ptr, None),
    "CGS update then Jacobi apply": lambda: (L.dfl_cgs_update(n4, 39, Q.ptr, n4, h.ptr, col(K), nrm.ptr, 1, work.ptr, None),
                                             L.dfl_pc_jacobi_apply(N, n4, d33.ptr, d1.ptr, col(3), tmp.ptr, None)),
}
t = api.Timer()
for name, fn in before.items():
    res = []
    for rep in range(8):
        fn()
        t.start(); spmv(); t.stop()
        res.append(t.ms())
    print("SpMV right after %-42s: %.4f ms (min %.4f)" % (name, float(np.median(res[1:])), min(res[1:])), flush=True)
