"""Probe: grouped timing (no host sync inside a group) of SpMV alone vs SpMV interleaved with another kernel."""
import sys, os, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dedflow_amd import api
from dedflow_amd.meshgen import kuhn_cube, synthetic_fields
M = int(sys.argv[1]) if len(sys.argv) > 1 else 119
mesh = kuhn_cube(M, jitter=0.2)
wg, dwg = synthetic_fields(mesh)
P = api.Problem(mesh)
L = api.lib()
wg_d, dwg_d = api.DeviceArray.from_numpy(wg), api.DeviceArray.from_numpy(dwg)
P.assemble_system(wg_d, dwg_d, None, want_J=True)
N = P.N; n4 = 4 * N
src = api.DeviceArray.from_numpy(np.random.default_rng(0).normal(size=6 * N))
x = api.DeviceArray.from_numpy(src.numpy()); y = api.DeviceArray(6 * N); z = api.DeviceArray(6 * N)
L.dfl_daxpy.argtypes = [C.c_int32, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]
REP = 10
def group(fn):
    t = api.Timer(); res = []
    for rep in range(5):
        fn()
        t.start()
        for _ in range(REP):
            fn()
        t.stop(); res.append(t.ms() / REP)
    return min(res)
spmv = lambda: P.matvec(x, y)
small = lambda: L.dfl_daxpy(n4, 0.0, src.ptr, z.ptr, None)
def both():
    P.matvec(x, y); L.dfl_daxpy(n4, 0.0, src.ptr, z.ptr, None)
def both_dep():
    P.matvec(x, y); L.dfl_daxpy(n4, 0.0, y.ptr, x.ptr, None)   # next SpMV input depends on this SpMV's output
a, b, c, d = group(spmv), group(small), group(both), group(both_dep)
print("SpMV x10 grouped           : %.4f ms per SpMV" % a)
print("daxpy(4N) x10 grouped      : %.4f ms" % b)
print("[SpMV, daxpy] x10          : %.4f ms per pair -> SpMV %.4f" % (c, c - b))
print("[SpMV, dependent daxpy] x10: %.4f ms per pair -> SpMV %.4f" % (d, d - b))
