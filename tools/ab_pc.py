"""Time-to-tolerance of the Krylov solve with the reference's Jacobi tree vs PC_ILU0 (multicolor block-DILU)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dedflow_amd import api
from dedflow_amd.meshgen import kuhn_cube, synthetic_fields

M = int(sys.argv[1]) if len(sys.argv) > 1 else 119
rtol = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-6
maxit = int(sys.argv[3]) if len(sys.argv) > 3 else 300
mesh = kuhn_cube(M, jitter=0.2)
wg, dwg = synthetic_fields(mesh)
L = api.lib()
for name, pctype in (("jacobi", api.PC_DECOMPOSITION), ("dilu", api.PC_ILU0)):
    P = api.Problem(mesh, maxit=maxit, atol=0.0, rtol=rtol)
    L.KrylovSetCheckInterval(P.ksp, 5)
    L.KrylovSetPCType(P.ksp, pctype)
    wg_d, dwg_d = api.DeviceArray.from_numpy(wg), api.DeviceArray.from_numpy(dwg)
    F_d, x_d = api.DeviceArray(6 * P.N), api.DeviceArray(6 * P.N)
    P.assemble_system(wg_d, dwg_d, F_d, want_J=False)
    P.assemble_system(wg_d, dwg_d, None, want_J=True)
    res = []
    for rep in range(3):
        x_d.zero()
        api.sync(); t0 = time.perf_counter()
        it, r0, hist, conv = P.solve(x_d, F_d)
        api.sync(); res.append(time.perf_counter() - t0)
    nc = L.PCDILUGetColors(L.KrylovGetPC(P.ksp), None) if pctype == api.PC_ILU0 else 0
    print("%-7s M=%d: %3d iterations to rtol %.0e (converged=%s, final rel %.2e)  solve %.1f ms (min of 3)  %.3f ms/it  node colors %d"
          % (name, M, it, rtol, conv, hist[-1] / r0, 1e3 * min(res), 1e3 * min(res) / max(it, 1), nc), flush=True)
    P.close()
