"""Probe (round 2): SpMV variants INSIDE the real GMRES loop (in-library event profiler, 40 fixed iterations) and back to
back.  variants: 3 = no XCD map, 4 = one slab per XCD (round-1 default), 7 = slab + LDS-staged store,
8 / 9 / 11 = XCD chunks of 64 / 512 / 8 workgroups, 10 = chunks of 64 + LDS-staged store."""
import sys, os, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dedflow_amd import api
from dedflow_amd.meshgen import kuhn_cube, synthetic_fields
M = int(sys.argv[1]) if len(sys.argv) > 1 else 119
variants = [int(v) for v in sys.argv[2:]] or [4, 3, 7, 8, 9, 10, 11, 4]
# values >= 100: pc_apply mode (v - 100) with SpMV variant 4; 200 / 201: Krylov work space from the heap / from the device pool
mesh = kuhn_cube(M, jitter=0.2)
wg, dwg = synthetic_fields(mesh)
P = api.Problem(mesh, maxit=40, atol=0.0, rtol=0.0)
L = api.lib()
L.dfl_tune.argtypes = [C.c_int, C.c_int]
wg_d, dwg_d = api.DeviceArray.from_numpy(wg), api.DeviceArray.from_numpy(dwg)
N = P.N
F_d, x_d, y_d = api.DeviceArray(6 * N), api.DeviceArray(6 * N), api.DeviceArray(6 * N)
P.assemble_system(wg_d, dwg_d, F_d, want_J=False)
P.assemble_system(wg_d, dwg_d, None, want_J=True)
P.solve(x_d, F_d)
ref = None
t = api.Timer()
for v in variants:
    L.dfl_tune(0, 4 if v >= 100 else v)
    L.dfl_tune(1, v - 100 if 100 <= v < 200 else 0)
    if v >= 200:
        L.DflKrylovWorkspaceInPool(v - 200)
        x_d.zero(); P.solve(x_d, F_d)   # reallocates the work space
    res = []
    for rep in range(3):
        x_d.zero()
        L.DflProfileEnable(1)
        t.start()
        it, r0, hist, _ = P.solve(x_d, F_d)
        t.stop()
        solve_ms = t.ms()
        tot, mn = C.c_double(0), C.c_double(0)
        cnt = L.DflProfileCollect(0, C.byref(tot), C.byref(mn))
        if rep == 2 and os.environ.get("PROBE_SEQ") == "1":
            buf = (C.c_double * 64)()
            L.DflProfileDurations.argtypes = [C.c_int, C.POINTER(C.c_double), C.c_int]
            k = L.DflProfileDurations(0, buf, 64)
            print("   per-launch ms:", " ".join("%.3f" % buf[i] for i in range(k)))
        L.DflProfileEnable(0)
        res.append((tot.value / max(cnt, 1), mn.value, solve_ms))
    x_d.zero()
    t.start(); P.solve(x_d, F_d); t.stop(); bare = t.ms()          # no per-kernel events
    P.matvec(F_d, y_d)
    t.start()
    for _ in range(10):
        P.matvec(F_d, y_d)
    t.stop()
    b2b = t.ms() / 10
    if ref is None:
        ref = hist.copy()
    same = bool(np.array_equal(hist, ref))
    print("variant %2d: in-loop spmv avg %.4f (min %.4f) ms | solve %.2f ms profiled, %.2f ms bare | back-to-back %.4f ms | history bitwise equal: %s"
          % (v, np.median([r[0] for r in res]), min(r[1] for r in res), np.median([r[2] for r in res]), bare, b2b, same), flush=True)
L.dfl_tune(0, 4)
L.dfl_tune(1, 0)
P.close()
