"""Probe: which operand's placement moves the SpMV time?  One 10M-tet problem; each operand in turn is replaced by copies
at other addresses (dummy allocations of varying size in front) and the launcher is timed directly."""
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
N, nnz1 = P.N, P.nnz1
spy = P.J.contents
rp_h, ci_h = P.pattern()
val = api.DeviceArray(16 * nnz1, np.float64, ptr=L.MatrixFSBlockValues(P.J), owner=False)
rp = api.DeviceArray.from_numpy(rp_h.astype(np.int32))
ci = api.DeviceArray.from_numpy(ci_h.astype(np.int32))
x = api.DeviceArray.from_numpy(np.random.default_rng(0).normal(size=6 * N))
y = api.DeviceArray(6 * N)
L.dfl_bcsr_spmv.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p]
L.dfl_bcsr_spmv.restype = None

def timeit(rp_, ci_, val_, x_, y_):
    t = api.Timer(); res = []
    for rep in range(4):
        L.dfl_bcsr_spmv(N, rp_.ptr, ci_.ptr, val_.ptr, 1.0, x_.ptr, 0.0, y_.ptr, None)
        t.start()
        for _ in range(10):
            L.dfl_bcsr_spmv(N, rp_.ptr, ci_.ptr, val_.ptr, 1.0, x_.ptr, 0.0, y_.ptr, None)
        t.stop(); res.append(t.ms() / 10)
    return min(res)

work = api.DeviceArray(8192)
out = api.DeviceArray(8)
L.dfl_dnrm2.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
def stream_read(v):
    t = api.Timer(); res = []
    for rep in range(4):
        t.start()
        for _ in range(5):
            L.dfl_dnrm2(16 * nnz1, v.ptr, out.ptr, work.ptr, None)
        t.stop(); res.append(t.ms() / 5)
    return 16.0 * nnz1 * 8 / min(res) / 1e6
print("plain streaming read of val (dfl_dnrm2): %.0f GB/s" % stream_read(val), flush=True)
print("baseline: %.4f ms  (val 0x%x ci 0x%x rp 0x%x x 0x%x y 0x%x)" % (timeit(rp, ci, val, x, y), val.ptr, ci.ptr, rp.ptr, x.ptr, y.ptr), flush=True)
pads = []
for trial, pad_mb in enumerate((3, 70, 513, 1201)):
    pads.append(api.DeviceArray(pad_mb * 131072))
    x2 = api.DeviceArray.from_numpy(x.numpy()); y2 = api.DeviceArray(6 * N)
    ci2 = api.DeviceArray.from_numpy(ci_h.astype(np.int32)); rp2 = api.DeviceArray.from_numpy(rp_h.astype(np.int32))
    print("trial %d: x' %.4f  y' %.4f  ci' %.4f  rp' %.4f  all four' %.4f" % (
        trial, timeit(rp, ci, val, x2, y), timeit(rp, ci, val, x, y2), timeit(rp, ci2, val, x, y), timeit(rp2, ci, val, x, y),
        timeit(rp2, ci2, val, x2, y2)), flush=True)
    val2 = api.DeviceArray(16 * nnz1)
    api.hip().hipMemcpy(C.c_void_p(val2.ptr), C.c_void_p(val.ptr), C.c_size_t(16 * nnz1 * 8), 3)
    print("         val' %.4f (0x%x)   val'+all %.4f   plain streaming read of val' %.0f GB/s" % (timeit(rp, ci, val2, x, y), val2.ptr, timeit(rp2, ci2, val2, x2, y2), stream_read(val2)), flush=True)
    print("         val' with x' %.4f  y' %.4f  ci' %.4f  rp' %.4f" % (timeit(rp, ci, val2, x2, y), timeit(rp, ci, val2, x, y2),
          timeit(rp, ci2, val2, x, y), timeit(rp2, ci, val2, x, y)), flush=True)
    pads.append(val2)
