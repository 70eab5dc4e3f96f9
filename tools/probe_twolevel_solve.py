"""One PC_TWOLEVEL solve to rtol 1e-4 on the bench system (for tools/trace_twolevel.sh: where does an outer iteration's time
go -- smoother, residual matvec, restriction, coarse solve, prolongation, FGMRES?).  Usage: python tools/probe_twolevel_solve.py [M]"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dedflow_amd import api
from dedflow_amd.meshgen import kuhn_cube, synthetic_fields
M = int(sys.argv[1]) if len(sys.argv) > 1 else 119
mesh = kuhn_cube(M, jitter=0.2)
wg, dwg = synthetic_fields(mesh)
L = api.lib()
P = api.Problem(mesh, maxit=100, atol=1e-12, rtol=1e-4, quiet=True)
N = P.N
wg_d, dwg_d = api.DeviceArray.from_numpy(wg), api.DeviceArray.from_numpy(dwg)
F_d, x_d = api.DeviceArray(6 * N), api.DeviceArray(6 * N)
P.assemble_system(wg_d, dwg_d, F_d, want_J=False)
P.assemble_system(wg_d, dwg_d, None, want_J=True)
if os.environ.get("DFL_AGG"):
    L.KrylovSetAggregateSize(P.ksp, int(os.environ["DFL_AGG"]))
L.KrylovSetPCType(P.ksp, api.PC_TWOLEVEL)
for rep in range(3):
    x_d.zero()
    api.sync(); t0 = time.perf_counter()
    it, r0, hist, conv = P.solve(x_d, F_d)
    api.sync(); dt = time.perf_counter() - t0
    print("solve %d: %d iterations, converged %s, %.1f ms (%.2f ms per iteration)" % (rep, it, conv, 1e3 * dt, 1e3 * dt / max(it, 1)))
pc = L.KrylovGetPC(P.ksp)
nagg, cnnz, inner = C.c_int32(0), C.c_int32(0), C.c_int64(0)
L.PCTwoLevelInfo(pc, C.byref(nagg), C.byref(cnnz), C.byref(inner))
print("aggregates %d, coarse nonzeros %d, inner iterations so far %d" % (nagg.value, cnnz.value, inner.value))
P.close()
