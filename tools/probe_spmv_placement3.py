"""Probe: SpMV time with the value array placed inside one big slab allocated first thing in the process (pristine VRAM)."""
import sys, os, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dedflow_amd import api
from dedflow_amd.meshgen import kuhn_cube, synthetic_fields
api.lib()
slab = api.DeviceArray(24 * (1 << 27))          # 24 GiB of f64 slots, before anything else touches VRAM
M = int(sys.argv[1]) if len(sys.argv) > 1 else 119
mesh = kuhn_cube(M, jitter=0.2)
wg, dwg = synthetic_fields(mesh)
P = api.Problem(mesh)
L = api.lib()
wg_d, dwg_d = api.DeviceArray.from_numpy(wg), api.DeviceArray.from_numpy(dwg)
P.assemble_system(wg_d, dwg_d, None, want_J=True)
N, nnz1 = P.N, P.nnz1
rp_h, ci_h = P.pattern()
val = api.DeviceArray(16 * nnz1, np.float64, ptr=L.MatrixFSBlockValues(P.J), owner=False)
rp = api.DeviceArray.from_numpy(rp_h.astype(np.int32)); ci = api.DeviceArray.from_numpy(ci_h.astype(np.int32))
x = api.DeviceArray.from_numpy(np.random.default_rng(0).normal(size=6 * N)); y = api.DeviceArray(6 * N)
L.dfl_bcsr_spmv.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p]
def timeit(vptr):
    t = api.Timer(); res = []
    for rep in range(4):
        L.dfl_bcsr_spmv(N, rp.ptr, ci.ptr, vptr, 1.0, x.ptr, 0.0, y.ptr, None)
        t.start()
        for _ in range(10):
            L.dfl_bcsr_spmv(N, rp.ptr, ci.ptr, vptr, 1.0, x.ptr, 0.0, y.ptr, None)
        t.stop(); res.append(t.ms() / 10)
    return min(res)
print("library allocation: %.4f ms (0x%x)" % (timeit(val.ptr), val.ptr), flush=True)
nb = 16 * nnz1 * 8
for off_gb in (0, 4, 8, 12, 16, 20, 1.37, 9.11):
    dst = slab.ptr + int(off_gb * (1 << 30)) // 256 * 256
    api.hip().hipMemcpy(C.c_void_p(dst), C.c_void_p(val.ptr), C.c_size_t(nb), 3)
    print("slab + %5.2f GiB: %.4f ms" % (off_gb, timeit(dst)), flush=True)
for trial in range(3):
    v2 = api.DeviceArray(16 * nnz1)
    api.hip().hipMemcpy(C.c_void_p(v2.ptr), C.c_void_p(val.ptr), C.c_size_t(nb), 3)
    print("fresh hipMalloc #%d: %.4f ms (0x%x)" % (trial, timeit(v2.ptr), v2.ptr), flush=True)
    keep = v2 if trial == 0 else None
