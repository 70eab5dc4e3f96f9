"""Probe (round 2): SpMV time against the placement of the OUTPUT vector, with and without the store.
One process: 16 candidate output vectors (hipMalloc'ed with spacers in between, plus slices of one large block),
SpMV variants 4 (default), 5 (no store), 6 (nontemporal store); groups of 6 launches, median of 5 groups."""
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
L.dfl_tune.argtypes = [C.c_int, C.c_int]
L.dfl_bcsr_spmv.argtypes = [i32, vp, vp, vp, f64, vp, f64, vp, vp]
wg_d, dwg_d = api.DeviceArray.from_numpy(wg), api.DeviceArray.from_numpy(dwg)
P.assemble_system(wg_d, dwg_d, None, want_J=True)
N = P.N; n4 = 4 * N
rp_h, ci_h = P.pattern()
rp = api.DeviceArray.from_numpy(rp_h.astype(np.int32)); ci = api.DeviceArray.from_numpy(ci_h.astype(np.int32))
val = L.MatrixFSBlockValues(P.J)
x = api.DeviceArray.from_numpy(np.random.default_rng(0).normal(size=6 * N))
cands = []
keep = []
for i in range(5):
    cands.append(("hipMalloc #%d" % i, api.DeviceArray(6 * N)))
    keep.append(api.DeviceArray(1000003 * (i + 1)))   # spacer
big = api.DeviceArray(8 * 6 * N + 4096)
for i in range(5, 8):
    cands.append(("slice %d of one block" % i, big.view(i * (6 * N + 512), 6 * N)))
# the library's device pool (one 32 GiB hipMalloc made at Init): single vectors and slices of a Krylov-basis-sized block
L.DflDeviceMalloc.restype = vp; L.DflDeviceMalloc.argtypes = [i64]
for i in range(3):
    cands.append(("pool vector #%d" % i, api.DeviceArray(6 * N, np.float64, ptr=L.DflDeviceMalloc(8 * 6 * N))))
    keep.append(L.DflDeviceMalloc(8 * 1000003 * (i + 1)))
pq = L.DflDeviceMalloc(8 * 4 * N * 42)
for i in (0, 20, 40):
    cands.append(("pool basis column %d" % i, api.DeviceArray(6 * N, np.float64, ptr=pq + 8 * 4 * N * i)))
hq = api.DeviceArray(4 * N * 42)
for i in (0, 20, 40):
    cands.append(("hipMalloc basis column %d" % i, hq.view(4 * N * i, 4 * N)))
t = api.Timer()


def run(y, variant):
    global x
    L.dfl_tune(0, variant)
    res = []
    L.dfl_bcsr_spmv(N, rp.ptr, ci.ptr, val, 1.0, x.ptr, 0.0, y.ptr, None)
    for g in range(5):
        t.start()
        for _ in range(6):
            L.dfl_bcsr_spmv(N, rp.ptr, ci.ptr, val, 1.0, x.ptr, 0.0, y.ptr, None)
        t.stop(); res.append(t.ms() / 6)
    return float(np.median(res))


print("%-26s %16s %10s %10s %10s %10s" % ("output vector", "address", "default", "no store", "nt store", "lds store"))
for name, y in cands:
    print("%-26s %#16x %10.4f %10.4f %10.4f %10.4f" % (name, y.ptr, run(y, 4), run(y, 5), run(y, 6), run(y, 7)), flush=True)
ya, yb = api.DeviceArray(6 * N), api.DeviceArray(6 * N)
L.dfl_tune(0, 4); L.dfl_bcsr_spmv(N, rp.ptr, ci.ptr, val, 1.0, x.ptr, 0.0, ya.ptr, None)
L.dfl_tune(0, 7); L.dfl_bcsr_spmv(N, rp.ptr, ci.ptr, val, 1.0, x.ptr, 0.0, yb.ptr, None)
api.sync()
print("lds-store variant equals default bitwise:", np.array_equal(ya.numpy(), yb.numpy()))
L.dfl_tune(0, 4)
P.close()
