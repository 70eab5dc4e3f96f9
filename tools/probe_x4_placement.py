"""Does the PLACEMENT of the 55-MB interleaved input copy (x4) matter to the matvec that gathers from it?  8 plain hipMalloc
blocks + 2 pool blocks as x4, the output in a pool block and in a heap block.  Usage: python tools/probe_x4_placement.py [M]"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dedflow_amd import api
from dedflow_amd.meshgen import kuhn_cube, synthetic_fields
M = int(sys.argv[1]) if len(sys.argv) > 1 else 119
mesh = kuhn_cube(M, jitter=0.2)
wg, dwg = synthetic_fields(mesh)
P = api.Problem(mesh)
L, H = api.lib(), api.hip()
vp = C.c_void_p
wg_d, dwg_d = api.DeviceArray.from_numpy(wg), api.DeviceArray.from_numpy(dwg)
P.assemble_system(wg_d, dwg_d, None, want_J=True)
N = P.N
x = api.DeviceArray.from_numpy(np.random.default_rng(0).normal(size=6 * N))
L.dfl_interleave4.argtypes = [C.c_int32, C.c_int32, C.c_int32, vp, vp, vp]
L.DflMatrixFSMatVecX4Range.argtypes = [C.POINTER(api.Matrix), vp, vp, C.c_int32, C.c_int32]
ys = {"y pool": api.DeviceArray(6 * N)}
yh = vp()
assert H.hipMalloc(C.byref(yh), 48 * N) == 0
ys["y heap"] = api.DeviceArray(6 * N, np.float64, ptr=yh.value)
cands = []
for k in range(8):
    p = vp()
    assert H.hipMalloc(C.byref(p), 32 * N) == 0
    cands.append(("heap %d" % k, p.value, None))
for k in range(2):
    a = api.DeviceArray(4 * N)
    cands.append(("pool %d" % k, a.ptr, a))
t = api.Timer()
for yname, y in ys.items():
    row = []
    for name, ptr, keep in cands:
        L.dfl_interleave4(0, N, N, x.ptr, ptr, L.DflStream())
        for _ in range(3):
            L.DflMatrixFSMatVecX4Range(P.J, ptr, y.ptr, 0, N)
        t.start()
        for _ in range(10):
            L.DflMatrixFSMatVecX4Range(P.J, ptr, y.ptr, 0, N)
        t.stop()
        row.append("%s %.4f" % (name, t.ms() / 10))
    print(yname + ": " + "  ".join(row))
t.start()
for _ in range(10):
    P.matvec(x, ys["y heap"])
t.stop()
print("MatrixMatVec (interleave + kernel, scratch of the matrix), y heap: %.4f ms" % (t.ms() / 10))
P.close()
