"""Probe: does the SpMV time depend on where the matrix lands in device memory?  Re-creates the 10M-tet problem several
times in one process with differently sized dummy allocations in front of it."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dedflow_amd import api
from dedflow_amd.meshgen import kuhn_cube, synthetic_fields
M = int(sys.argv[1]) if len(sys.argv) > 1 else 119
mesh = kuhn_cube(M, jitter=0.2)
wg, dwg = synthetic_fields(mesh)
for pad_mb in (0, 300, 1100, 2500, 0, 4097):
    pad = api.DeviceArray(pad_mb * 131072) if pad_mb else None
    P = api.Problem(mesh)
    wg_d, dwg_d = api.DeviceArray.from_numpy(wg), api.DeviceArray.from_numpy(dwg)
    P.assemble_system(wg_d, dwg_d, None, want_J=True)
    x = api.DeviceArray.from_numpy(np.random.default_rng(0).normal(size=6 * P.N))
    y = api.DeviceArray(6 * P.N)
    t = api.Timer(); res = []
    for rep in range(5):
        P.matvec(x, y)
        t.start()
        for _ in range(10):
            P.matvec(x, y)
        t.stop(); res.append(t.ms() / 10)
    import ctypes as C
    vptr = api.lib().MatrixFSBlockValues(P.J)
    print("pad %5d MB: SpMV %.4f ms (min %.4f)  val @ 0x%x  x @ 0x%x" % (pad_mb, float(np.median(res)), min(res), vptr, x.ptr), flush=True)
    P.close()
    del x, y, wg_d, dwg_d, pad
