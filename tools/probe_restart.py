"""Probe: convergence of PC_ILU0 (multicolor block-DILU) GMRES at M (default 203 = 50M tets) for full GMRES and GMRES(m)
restarts: relative residual every 40 iterations, time per solve.  args: M [pc=dilu|jacobi] then maxit:restart pairs"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dedflow_amd import api
from dedflow_amd.meshgen import kuhn_cube, synthetic_fields
M = int(sys.argv[1]) if len(sys.argv) > 1 else 203
pc = sys.argv[2] if len(sys.argv) > 2 else "dilu"
cfgs = [tuple(int(v) for v in c.split(":")) for c in (sys.argv[3:] or ["400:0", "800:200", "800:80"])]
mesh = kuhn_cube(M, jitter=0.2)
wg, dwg = synthetic_fields(mesh)
N = mesh.num_node
wg[3 * N:4 * N] = 0.0
L = api.lib()
P = api.Problem(mesh, maxit=120, atol=1e-30, rtol=1e-30)
L.KrylovSetPCType(P.ksp, {"dilu": api.PC_ILU0, "twolevel": api.PC_TWOLEVEL}.get(pc, api.PC_DECOMPOSITION))
if os.environ.get("DFL_AGG"):
    L.KrylovSetAggregateSize(P.ksp, int(os.environ["DFL_AGG"]))
wg_d, dwg_d = api.DeviceArray.from_numpy(wg), api.DeviceArray.from_numpy(0.1 * dwg)
F_d, x_d = api.DeviceArray(6 * N), api.DeviceArray(6 * N)
P.assemble_system(wg_d, dwg_d, F_d, want_J=False)
P.assemble_system(wg_d, dwg_d, None, want_J=True)
api.sync()
STRIDE = int(os.environ.get("DFL_PROBE_STRIDE", "40"))
import ctypes as C
ksp = C.cast(P.ksp, C.POINTER(C.c_int32))
for maxit, restart in cfgs:
    ksp[0] = maxit                      # Krylov.max_iter is the first field
    L.KrylovSetRestart(P.ksp, restart)
    x_d.zero()
    api.sync(); t = time.perf_counter()
    it, r0, hist, conv = P.solve(x_d, F_d)
    api.sync(); t = time.perf_counter() - t
    rel = hist / r0
    pts = ", ".join("%d: %.2e" % (k + 1, rel[k]) for k in range(STRIDE - 1, len(rel), STRIDE))
    if pc == "twolevel":
        na, cn, inner = C.c_int32(0), C.c_int32(0), C.c_int64(0)
        L.PCTwoLevelInfo(L.KrylovGetPC(P.ksp), C.byref(na), C.byref(cn), C.byref(inner))
        print("level-1 inner iterations so far: %d" % inner.value)
    print("maxit %d restart %d: %d its in %.2f s (%.1f ms/it), r0 %.3e; rel residual @ %s" % (maxit, restart, it, t, 1e3 * t / max(it, 1), r0, pts), flush=True)
P.close()
