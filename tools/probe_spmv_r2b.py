"""Probe (round 2): SpMV inside the REAL GMRES loop (in-library event profiler) next to the emulated Arnoldi step, with the
emulation's vectors taken (a) from hipMalloc and (b) from the library's device pool.  Run once per setting of
DFL_DEVICE_POOL_GB (unset = pool on, 0 = pool off)."""
import sys, os, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dedflow_amd import api
from dedflow_amd.meshgen import kuhn_cube, synthetic_fields
M = int(sys.argv[1]) if len(sys.argv) > 1 else 119
mesh = kuhn_cube(M, jitter=0.2)
wg, dwg = synthetic_fields(mesh)
its = 40
P = api.Problem(mesh, maxit=its, atol=0.0, rtol=0.0)
L = api.lib()
vp, i32, i64, f64 = C.c_void_p, C.c_int32, C.c_int64, C.c_double
L.DflDeviceMalloc.restype = vp; L.DflDeviceMalloc.argtypes = [i64]; L.DflDeviceFree.argtypes = [vp]
wg_d, dwg_d = api.DeviceArray.from_numpy(wg), api.DeviceArray.from_numpy(dwg)
N = P.N; n4 = 4 * N
F_d, x_d = api.DeviceArray(6 * N), api.DeviceArray(6 * N)
P.assemble_system(wg_d, dwg_d, F_d, want_J=False)
P.assemble_system(wg_d, dwg_d, None, want_J=True)
print("pool setting DFL_DEVICE_POOL_GB =", os.environ.get("DFL_DEVICE_POOL_GB", "(default: on)"))
val = L.MatrixFSBlockValues(P.J)
print("val @ %#x" % val)
TAGS = {"spmv": 0, "cgs_dots": 1, "cgs_update": 2, "pc_apply": 3}
for rep in range(3):
    x_d.zero()
    L.DflProfileEnable(1)
    P.solve(x_d, F_d)
    api.sync()
    line = []
    for name, tag in TAGS.items():
        tot, mn = C.c_double(0), C.c_double(0)
        cnt = L.DflProfileCollect(tag, C.byref(tot), C.byref(mn))
        line.append("%s avg %.4f min %.4f (n=%d)" % (name, tot.value / max(cnt, 1), mn.value, cnt))
    L.DflProfileEnable(0)
    print("real solve %d: " % rep + " | ".join(line), flush=True)

# emulation, vectors from hipMalloc and from the pool
L.dfl_cgs_work_size.restype = C.c_int64
L.dfl_pc_jacobi_setup.argtypes = [i32, vp, vp, vp, vp, vp, vp]
L.dfl_pc_jacobi_apply_scaled.argtypes = [i32, i32, vp, vp, vp, vp, vp, vp, vp]
L.dfl_cgs_dots.argtypes = [i32, i32, vp, i64, vp, vp, vp, vp]
L.dfl_cgs_update.argtypes = [i32, i32, vp, i64, vp, vp, vp, C.c_int, vp, vp]
L.dfl_bcsr_spmv.argtypes = [i32, vp, vp, vp, f64, vp, f64, vp, vp]
rp_h, ci_h = P.pattern()
rp = api.DeviceArray.from_numpy(rp_h.astype(np.int32)); ci = api.DeviceArray.from_numpy(ci_h.astype(np.int32))
h = api.DeviceArray(256); nrm = api.DeviceArray(8); nrm.upload(np.ones(8))
K = 40
work = api.DeviceArray(int(L.dfl_cgs_work_size(n4, K + 2)) + 16)
d33 = api.DeviceArray(9 * N); d1 = api.DeviceArray(N)
L.dfl_pc_jacobi_setup(N, rp.ptr, ci.ptr, val, d33.ptr, d1.ptr, None)
t = api.Timer()


def emulate(Qp, tmpp, label, fixed_y=None):
    col = lambda k: Qp + 8 * n4 * k
    res = []
    for i in range(14):
        k = 10 + i
        L.dfl_cgs_dots(n4, k, Qp, n4, col(k), h.ptr, work.ptr, None)
        L.dfl_cgs_update(n4, k, Qp, n4, h.ptr, col(k), nrm.ptr + 8, 1, work.ptr, None)
        L.dfl_pc_jacobi_apply_scaled(N, n4, d33.ptr, d1.ptr, col(k), nrm.ptr, col(k), tmpp, None)
        t.start()
        L.dfl_bcsr_spmv(N, rp.ptr, ci.ptr, val, 1.0, tmpp, 0.0, fixed_y if fixed_y else col(k + 1), None)
        t.stop(); res.append(t.ms())
    r = np.array(res[2:])
    print("emulated step, %-44s spmv median %.4f min %.4f max %.4f   (Q @ %#x, tmp @ %#x)" % (label, np.median(r), r.min(), r.max(), Qp, tmpp), flush=True)


init = np.random.default_rng(0).normal(size=n4 * (K + 2)) * 1e-3
Qh = api.DeviceArray(n4 * (K + 2)); Qh.upload(init); tmph = api.DeviceArray(12 * N)
emulate(Qh.ptr, tmph.ptr, "Q, tmp from hipMalloc")
Qp = L.DflDeviceMalloc(8 * n4 * (K + 2)); tmpp = L.DflDeviceMalloc(8 * 12 * N)
api._chk(api.hip().hipMemcpy(Qp, init.ctypes.data, init.nbytes, 1))
emulate(Qp, tmpp, "Q, tmp from the library allocator")
emulate(Qp, tmph.ptr, "Q from the library allocator, tmp hipMalloc")
emulate(Qh.ptr, tmpp, "Q hipMalloc, tmp from the library allocator")
yf = api.DeviceArray(6 * N)
emulate(Qh.ptr, tmph.ptr, "hipMalloc, y = one late hipMalloc'ed vector", fixed_y=yf.ptr)
yp = L.DflDeviceMalloc(8 * 6 * N)
emulate(Qh.ptr, tmph.ptr, "hipMalloc, y = one vector from the allocator", fixed_y=yp)
P.close()
