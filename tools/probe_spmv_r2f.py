"""Probe (round 2): is the SpMV placement effect about WHERE the output vector lies, or about where it lies RELATIVE to the
value array?  The 3.3 GB value array is used from the device pool (where the library keeps it) and from two plain hipMalloc
copies; the output vector from the pool and from plain hipMalloc blocks; every pair is timed (groups of 6 launches, median
of 5 groups, default SpMV variant and the no-store variant)."""
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
L.DflDeviceMalloc.restype = vp; L.DflDeviceMalloc.argtypes = [i64]
wg_d, dwg_d = api.DeviceArray.from_numpy(wg), api.DeviceArray.from_numpy(dwg)
P.assemble_system(wg_d, dwg_d, None, want_J=True)
N = P.N
rp_h, ci_h = P.pattern()
rp = api.DeviceArray.from_numpy(rp_h.astype(np.int32)); ci = api.DeviceArray.from_numpy(ci_h.astype(np.int32))
nval = P.nnz1 * 16
val_pool = L.MatrixFSBlockValues(P.J)
hip = api.hip()
hip.hipMemcpy.argtypes = [vp, vp, C.c_size_t, C.c_int]
hip.hipMalloc.argtypes = [C.POINTER(vp), C.c_size_t]
vals = [("val in the pool (library)", val_pool)]
keep = []
for i in range(int(os.environ.get('DFL_R2F_COPIES', '2'))):
    blk = api.DeviceArray(nval)
    hip.hipMemcpy(blk.ptr, val_pool, 8 * nval, 3)
    keep.append(blk)
    vals.append(("val in hipMalloc copy #%d" % i, blk.ptr))
    keep.append(api.DeviceArray(5000011 * (i + 1)))
second_pool = L.DflDeviceMalloc(8 * nval)
hip.hipMemcpy(second_pool, val_pool, 8 * nval, 3)
vals.append(("val as a 2nd pool block", second_pool))
x = api.DeviceArray.from_numpy(np.random.default_rng(0).normal(size=6 * N))
ys = [("y pool #0", api.DeviceArray(6 * N, np.float64, ptr=L.DflDeviceMalloc(8 * 6 * N))),
      ("y pool #1", api.DeviceArray(6 * N, np.float64, ptr=L.DflDeviceMalloc(8 * 6 * N)))]
for i in range(3):
    ys.append(("y hipMalloc #%d" % i, api.DeviceArray(6 * N)))
    keep.append(api.DeviceArray(1000003 * (i + 1)))
# other allocation kinds for the output vector (hipExtMallocWithFlags): fine-grained, uncached, physically contiguous
hip.hipExtMallocWithFlags.argtypes = [C.POINTER(vp), C.c_size_t, C.c_uint]
for nm, fl in (("y finegrained", 0x1), ("y uncached", 0x3), ("y contiguous", 0x4)):
    pp = vp(0)
    rc = hip.hipExtMallocWithFlags(C.byref(pp), 8 * 6 * N, fl)
    if rc == 0 and pp.value:
        hip.hipMemset(pp, 0, 8 * 6 * N)
        ys.append((nm, api.DeviceArray(6 * N, np.float64, ptr=pp.value)))
    else:
        print("hipExtMallocWithFlags(%#x) failed: %d" % (fl, rc))
# output vectors allocated behind big spacers (do "regions" follow the physical distance from the value array?)
spacers = []
for gb in (40, 40, 40, 40):
    sp = vp(0)
    if hip.hipMalloc(C.byref(sp), gb << 30) != 0:
        break
    spacers.append(sp)
    ys.append(("y after %d GB" % (40 * len(spacers)), api.DeviceArray(6 * N)))
t = api.Timer()


def run(v, y, variant):
    L.dfl_tune(0, variant)
    L.dfl_bcsr_spmv(N, rp.ptr, ci.ptr, v, 1.0, x.ptr, 0.0, y.ptr, None)
    res = []
    for g in range(5):
        t.start()
        for _ in range(6):
            L.dfl_bcsr_spmv(N, rp.ptr, ci.ptr, v, 1.0, x.ptr, 0.0, y.ptr, None)
        t.stop(); res.append(t.ms() / 6)
    return float(np.median(res))


print("%-28s %18s | " % ("value array", "address") + " | ".join("%-16s" % n for n, _ in ys) + " | no store")
for vn, v in vals:
    row = ["%.4f" % run(v, y, 4) for _, y in ys]
    print("%-28s %#18x | " % (vn, v) + " | ".join("%-16s" % r for r in row) + " | %.4f" % run(v, ys[0][1], 5), flush=True)
print("y addresses: " + ", ".join("%s %#x" % (n, y.ptr) for n, y in ys))
# the same with the INPUT vector in each class (value array where the library keeps it)
xs = [("x pool", api.DeviceArray(6 * N, np.float64, ptr=L.DflDeviceMalloc(8 * 6 * N))), ("x hipMalloc", api.DeviceArray(6 * N))]
hip.hipMemcpy(xs[0][1].ptr, x.ptr, 8 * 6 * N, 3)
hip.hipMemcpy(xs[1][1].ptr, x.ptr, 8 * 6 * N, 3)
x_keep = x
for xn, xv in xs:
    x = xv
    row = ["%.4f" % run(val_pool, y, 4) for _, y in ys[:5]]
    print("%-28s %#18x | " % (xn + " (val in the pool)", xv.ptr) + " | ".join("%-16s" % r for r in row), flush=True)
x = x_keep
L.dfl_tune(0, 4)
P.close()
