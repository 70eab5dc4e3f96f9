"""Probe (round 2): is the in-loop SpMV time a property of WHERE the Krylov work space lies?  K candidate work spaces
(basis Q of 42 columns + tmp), each a fresh hipMalloc, held simultaneously; emulated Arnoldi steps
[cgs_dots, cgs_update, pc_apply_scaled, spmv -> next column] with the SpMV event-timed, for every (Q_i, tmp_j) on the diagonal
and a few off-diagonal pairs."""
import sys, os, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dedflow_amd import api
from dedflow_amd.meshgen import kuhn_cube, synthetic_fields
M = int(sys.argv[1]) if len(sys.argv) > 1 else 119
K = int(sys.argv[2]) if len(sys.argv) > 2 else 6
mesh = kuhn_cube(M, jitter=0.2)
wg, dwg = synthetic_fields(mesh)
P = api.Problem(mesh, maxit=40, atol=0.0, rtol=0.0)
L = api.lib()
vp, i32, i64, f64 = C.c_void_p, C.c_int32, C.c_int64, C.c_double
wg_d, dwg_d = api.DeviceArray.from_numpy(wg), api.DeviceArray.from_numpy(dwg)
N = P.N; n4 = 4 * N
P.assemble_system(wg_d, dwg_d, None, want_J=True)
val = L.MatrixFSBlockValues(P.J)
L.dfl_cgs_work_size.restype = C.c_int64
L.dfl_pc_jacobi_setup.argtypes = [i32, vp, vp, vp, vp, vp, vp]
L.dfl_pc_jacobi_apply_scaled.argtypes = [i32, i32, vp, vp, vp, vp, vp, vp, vp]
L.dfl_cgs_dots.argtypes = [i32, i32, vp, i64, vp, vp, vp, vp]
L.dfl_cgs_update.argtypes = [i32, i32, vp, i64, vp, vp, vp, C.c_int, vp, vp]
L.dfl_bcsr_spmv.argtypes = [i32, vp, vp, vp, f64, vp, f64, vp, vp]
rp_h, ci_h = P.pattern()
rp = api.DeviceArray.from_numpy(rp_h.astype(np.int32)); ci = api.DeviceArray.from_numpy(ci_h.astype(np.int32))
h = api.DeviceArray(256); nrm = api.DeviceArray(8); nrm.upload(np.ones(8))
KC = 42
work = api.DeviceArray(int(L.dfl_cgs_work_size(n4, KC)) + 16)
d33 = api.DeviceArray(9 * N); d1 = api.DeviceArray(N)
L.dfl_pc_jacobi_setup(N, rp.ptr, ci.ptr, val, d33.ptr, d1.ptr, None)
init = np.random.default_rng(0).normal(size=n4 * KC) * 1e-3
Qs, tmps = [], []
for i in range(K):
    q = api.DeviceArray(n4 * KC); q.upload(init); Qs.append(q)
    tmps.append(api.DeviceArray(12 * N))
t = api.Timer()


def emulate(Qp, tmpp):
    col = lambda k: Qp + 8 * n4 * k
    res = []
    for i in range(12):
        k = 10 + i
        L.dfl_cgs_dots(n4, k, Qp, n4, col(k), h.ptr, work.ptr, None)
        L.dfl_cgs_update(n4, k, Qp, n4, h.ptr, col(k), nrm.ptr + 8, 1, work.ptr, None)
        L.dfl_pc_jacobi_apply_scaled(N, n4, d33.ptr, d1.ptr, col(k), nrm.ptr, col(k), tmpp, None)
        t.start()
        L.dfl_bcsr_spmv(N, rp.ptr, ci.ptr, val, 1.0, tmpp, 0.0, col(k + 1), None)
        t.stop(); res.append(t.ms())
    return float(np.median(res[2:]))


print("val @ %#x" % val)
for rep in range(2):
    for i in range(K):
        print("rep %d: Q%d @ %#x, tmp%d @ %#x : in-loop spmv %.4f ms" % (rep, i, Qs[i].ptr, i, tmps[i].ptr, emulate(Qs[i].ptr, tmps[i].ptr)), flush=True)
for i, j in ((0, 1), (1, 0), (2, 3), (3, 2)):
    print("Q%d with tmp%d: %.4f ms" % (i, j, emulate(Qs[i].ptr, tmps[j].ptr)), flush=True)
P.close()
