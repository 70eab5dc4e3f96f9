"""Probe (round 2): why does the block-CSR SpMV take 0.71 ms inside the Krylov loop and 0.57 ms back to back?

Labelled launch sequences at M (default 119), every SpMV bracketed by its own event pair when PROBE_EVENTS=1 (default);
with PROBE_EVENTS=0 the same sequences run bare so that `rocprofv3 --kernel-trace --pmc ...` can attribute counters to
dispatches by their order:
   seq A   12 x  spmv                                   (x constant, y the same vector)
   seq B   12 x [pc_apply -> spmv]                      (x = freshly written tmp)
   seq C   12 x [pc_apply_scaled (writes q AND tmp) -> spmv]
   seq D   12 x [cgs_dots(k=20) -> cgs_update(k=20) -> spmv]
   seq E   12 x [cgs_dots -> cgs_update -> pc_apply_scaled -> spmv into column k+1]   (one emulated Arnoldi step)
   seq F   as E with the spmv writing ONE fixed output vector instead of a fresh basis column
"""
import sys, os, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dedflow_amd import api
from dedflow_amd.meshgen import kuhn_cube, synthetic_fields

M = int(sys.argv[1]) if len(sys.argv) > 1 else 119
EV = os.environ.get("PROBE_EVENTS", "1") == "1"
REPS = int(os.environ.get("PROBE_REPS", "12"))
mesh = kuhn_cube(M, jitter=0.2)
wg, dwg = synthetic_fields(mesh)
P = api.Problem(mesh)
L = api.lib()
wg_d, dwg_d = api.DeviceArray.from_numpy(wg), api.DeviceArray.from_numpy(dwg)
P.assemble_system(wg_d, dwg_d, None, want_J=True)
N = P.N; n4 = 4 * N; K = 40
Q = api.DeviceArray(n4 * (K + 2)); tmp = api.DeviceArray(12 * N); yfix = api.DeviceArray(6 * N)
h = api.DeviceArray(256); nrm = api.DeviceArray(8)
L.dfl_cgs_work_size.restype = C.c_int64
work = api.DeviceArray(int(L.dfl_cgs_work_size(n4, K + 2)) + 16)
d33 = api.DeviceArray(9 * N); d1 = api.DeviceArray(N)
vp, i32, i64, f64 = C.c_void_p, C.c_int32, C.c_int64, C.c_double
L.dfl_pc_jacobi_setup.argtypes = [i32, vp, vp, vp, vp, vp, vp]
L.dfl_pc_jacobi_apply.argtypes = [i32, i32, vp, vp, vp, vp, vp]
L.dfl_pc_jacobi_apply_scaled.argtypes = [i32, i32, vp, vp, vp, vp, vp, vp, vp]
L.dfl_cgs_dots.argtypes = [i32, i32, vp, i64, vp, vp, vp, vp]
L.dfl_cgs_update.argtypes = [i32, i32, vp, i64, vp, vp, vp, C.c_int, vp, vp]
L.dfl_bcsr_spmv.argtypes = [i32, vp, vp, vp, f64, vp, f64, vp, vp]
rp_h, ci_h = P.pattern()
rp = api.DeviceArray.from_numpy(rp_h.astype(np.int32)); ci = api.DeviceArray.from_numpy(ci_h.astype(np.int32))
val = L.MatrixFSBlockValues(P.J)
L.dfl_pc_jacobi_setup(N, rp.ptr, ci.ptr, val, d33.ptr, d1.ptr, None)
Q.upload(np.random.default_rng(0).normal(size=n4 * (K + 2)) * 1e-3)
nrm.upload(np.ones(8))
col = lambda k: Q.ptr + 8 * n4 * k
t = api.Timer()


def timed(fn):
    if not EV:
        fn(); return 0.0
    t.start(); fn(); t.stop()
    return t.ms()


def spmv(x, y):
    return timed(lambda: L.dfl_bcsr_spmv(N, rp.ptr, ci.ptr, val, 1.0, x, 0.0, y, None))


def pc(k):
    L.dfl_pc_jacobi_apply(N, n4, d33.ptr, d1.ptr, col(k), tmp.ptr, None)


def pcs(k):
    L.dfl_pc_jacobi_apply_scaled(N, n4, d33.ptr, d1.ptr, col(k), nrm.ptr, col(k), tmp.ptr, None)


def cgs(k):
    L.dfl_cgs_dots(n4, k + 1, Q.ptr, n4, col(k + 1), h.ptr, work.ptr, None)
    L.dfl_cgs_update(n4, k + 1, Q.ptr, n4, h.ptr, col(k + 1), nrm.ptr + 8, 1, work.ptr, None)


def report(name, res):
    if EV:
        r = np.array(res[2:])
        print("%-78s median %.4f  min %.4f  max %.4f ms" % (name, np.median(r), r.min(), r.max()), flush=True)
    else:
        print("%-78s (bare, %d launches)" % (name, len(res)), flush=True)


kk = 20
report("A  spmv back to back", [spmv(tmp.ptr, yfix.ptr) for _ in range(REPS)])
res = []
for _ in range(REPS):
    pc(3); res.append(spmv(tmp.ptr, yfix.ptr))
report("B  [pc_apply -> spmv]", res)
res = []
for _ in range(REPS):
    pcs(3); res.append(spmv(tmp.ptr, yfix.ptr))
report("C  [pc_apply_scaled -> spmv]", res)
res = []
for _ in range(REPS):
    cgs(kk); res.append(spmv(tmp.ptr, yfix.ptr))
report("D  [cgs_dots, cgs_update (k=20) -> spmv]", res)
res = []
for i in range(REPS):
    k = 10 + i
    cgs(k - 1); pcs(k); res.append(spmv(tmp.ptr, col(k + 1)))
report("E  [cgs(k) -> pc_apply_scaled -> spmv into Q[:,k+1]], k = 10..", res)
res = []
for i in range(REPS):
    k = 10 + i
    cgs(k - 1); pcs(k); res.append(spmv(tmp.ptr, yfix.ptr))
report("F  as E, spmv output into one fixed vector", res)
if EV:
    # sustained: 400 back-to-back launches timed in groups of 50 (does the clock sag?)
    out = []
    for g in range(8):
        t.start()
        for _ in range(50):
            L.dfl_bcsr_spmv(N, rp.ptr, ci.ptr, val, 1.0, tmp.ptr, 0.0, yfix.ptr, None)
        t.stop(); out.append(round(t.ms() / 50, 4))
    print("sustained groups of 50 spmv:", out, flush=True)
    # full solve through the library for reference (in-library profiler)
P.close()
