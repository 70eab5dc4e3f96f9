"""Probe: repeated problem lifecycles with every preconditioner (Jacobi tree, PC_ILU0, PC_TWOLEVEL) and the DEM sweep:
pool in-use returns to the same level after close, free VRAM does not creep (host/pc_twolevel.c, FGMRES Z basis, patch
schedules, particle workspace)."""
import sys, ctypes as C, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dedflow_amd import api
from dedflow_amd.meshgen import kuhn_cube, synthetic_fields, dem_particles
L = api.lib()
L.DflDevicePoolStats.argtypes = [C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
def stats():
    r, u = C.c_int64(0), C.c_int64(0); L.DflDevicePoolStats(C.byref(r), C.byref(u)); return r.value, u.value
free0, tot = C.c_size_t(0), C.c_size_t(0)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 40
m = kuhn_cube(M, jitter=0.2)
wg, dwg = synthetic_fields(m)
N = m.num_node
wg[3 * N:4 * N] = 0.0
xp, vp_, R = dem_particles(20000, 0.006)
hist = []
for rep in range(6):
    pcs = [api.PC_DECOMPOSITION, api.PC_ILU0, api.PC_TWOLEVEL]
    P = api.Problem(m, maxit=60, atol=1e-12, rtol=1e-4, quiet=True)
    L.KrylovSetPCType(P.ksp, pcs[rep % 3])
    part = api.Particles(xp, vp_, R, dt=1e-4)
    st = [api.DeviceArray.from_numpy(a) for a in (wg, 0.1 * dwg, 0.1 * dwg)]
    F, dx = api.DeviceArray(6 * N), api.DeviceArray(6 * N)
    for s in range(3):
        P.time_step(st[0], st[1], st[2], F, dx, newton_maxit=2, particles=part, dem_substeps=3)
    api.sync()
    inuse = stats()
    part.close(); P.close()
    del st, F, dx
    api.hip().hipMemGetInfo(C.byref(free0), C.byref(tot))
    hist.append((stats()[1], free0.value))
    print("rep %d pc %d: pool in use during %d MiB, after close %d MiB; device free %d MiB" %
          (rep, pcs[rep % 3], inuse[1] >> 20, hist[-1][0] >> 20, hist[-1][1] >> 20), flush=True)
ok = all(h[0] == hist[0][0] for h in hist) and abs(hist[-1][1] - hist[2][1]) < (64 << 20)
print("LIFECYCLE_OK" if ok else "LIFECYCLE_LEAK", flush=True)
