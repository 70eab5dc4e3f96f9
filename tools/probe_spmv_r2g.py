"""Probe (round 2): SpMV time against the OFFSET of the output vector inside one large block (value array where the library
keeps it).  Coarse sweep: every 64 MiB over 8 GiB; fine sweep: every 2 MiB over the first 256 MiB; finer: every 64 KiB over
4 MiB.  Is the slow / fast split periodic in the address?"""
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
vp, i32, i64, f64 = C.c_void_p, C.c_int32, C.c_int64, C.c_double
L.dfl_bcsr_spmv.argtypes = [i32, vp, vp, vp, f64, vp, f64, vp, vp]
L.DflDeviceMalloc.restype = vp; L.DflDeviceMalloc.argtypes = [i64]
wg_d, dwg_d = api.DeviceArray.from_numpy(wg), api.DeviceArray.from_numpy(dwg)
P.assemble_system(wg_d, dwg_d, None, want_J=True)
N = P.N
rp_h, ci_h = P.pattern()
rp = api.DeviceArray.from_numpy(rp_h.astype(np.int32)); ci = api.DeviceArray.from_numpy(ci_h.astype(np.int32))
val = L.MatrixFSBlockValues(P.J)
x = api.DeviceArray.from_numpy(np.random.default_rng(0).normal(size=6 * N))
hip = api.hip()
hip.hipMalloc.argtypes = [C.POINTER(vp), C.c_size_t]
arena = vp(0)
AR = 9 << 30
assert hip.hipMalloc(C.byref(arena), AR) == 0
pool_arena = L.DflDeviceMalloc(9 << 30)
t = api.Timer()


def run(yptr):
    L.dfl_bcsr_spmv(N, rp.ptr, ci.ptr, val, 1.0, x.ptr, 0.0, yptr, None)
    res = []
    for g in range(3):
        t.start()
        for _ in range(5):
            L.dfl_bcsr_spmv(N, rp.ptr, ci.ptr, val, 1.0, x.ptr, 0.0, yptr, None)
        t.stop(); res.append(t.ms() / 5)
    return float(np.median(res))


print("value array at %#x, heap arena at %#x, pool arena at %#x" % (val, arena.value, pool_arena))
for name, base in (("heap arena", arena.value), ("pool arena", pool_arena)):
    for label, step, count in (("64 MiB", 64 << 20, 128), ("2 MiB", 2 << 20, 128), ("64 KiB", 64 << 10, 64)):
        ts = [run(base + k * step) for k in range(count)]
        slow = "".join("S" if v > 0.62 else "." for v in ts)
        print("%s, offsets k x %s: min %.4f max %.4f  %s" % (name, label, min(ts), max(ts), slow), flush=True)
P.close()
