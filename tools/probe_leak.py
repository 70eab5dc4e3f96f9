"""Probe: device memory over repeated problem lifecycles (pool in-use returns to zero, free VRAM stays constant)."""
import sys, ctypes as C
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dedflow_amd import api
from dedflow_amd.meshgen import kuhn_cube, synthetic_fields
L = api.lib()
L.Init(0, None)
L.DflDevicePoolStats.argtypes = [C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
def stats():
    r, u = C.c_int64(0), C.c_int64(0); L.DflDevicePoolStats(C.byref(r), C.byref(u)); return r.value, u.value
free0 = C.c_size_t(0); tot = C.c_size_t(0)
m = kuhn_cube(40, jitter=0.2)
wg, dwg = synthetic_fields(m)
for rep in range(4):
    P = api.Problem(m, maxit=30)
    wg_d, dwg_d = api.DeviceArray.from_numpy(wg), api.DeviceArray.from_numpy(dwg)
    F_d, x_d = api.DeviceArray(6 * P.N), api.DeviceArray(6 * P.N)
    P.assemble_system(wg_d, dwg_d, F_d, want_J=False)
    P.assemble_system(wg_d, dwg_d, None, want_J=True)
    P.solve(x_d, F_d)
    api.sync()
    inuse = stats()
    P.close()
    del wg_d, dwg_d, F_d, x_d
    api.hip().hipMemGetInfo(C.byref(free0), C.byref(tot))
    print("rep", rep, "pool in use during", inuse[1] >> 20, "MiB, after close", stats()[1] >> 20, "MiB; device free", free0.value >> 20, "MiB", flush=True)
