"""Probe: streaming write / read bandwidth into device-pool blocks and into plain hipMalloc blocks (hipMemsetAsync = write only,
dfl_dcopy = read + write, dfl_ddot-style read only via dfl_dnrm2), 1 GiB each, median of 7."""
import sys, os, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dedflow_amd import api
L = api.lib()
vp, i32, i64, f64 = C.c_void_p, C.c_int32, C.c_int64, C.c_double
L.Init(0, None)
L.DflDeviceMalloc.restype = vp; L.DflDeviceMalloc.argtypes = [i64]
hip = api.hip()
hip.hipMemsetAsync.argtypes = [vp, C.c_int, C.c_size_t, vp]
L.dfl_dcopy.argtypes = [i32, vp, vp, vp]
L.dfl_dnrm2.argtypes = [i32, vp, vp, vp, vp]
n = 1 << 27   # doubles = 1 GiB
bufs = []
for i in range(2):
    bufs.append(("pool #%d" % i, L.DflDeviceMalloc(8 * n)))
keep = []
for i in range(3):
    b = api.DeviceArray(n); keep.append(b)
    bufs.append(("hipMalloc #%d" % i, b.ptr))
    keep.append(api.DeviceArray(3000017 * (i + 1)))
src = api.DeviceArray(n)
work = api.DeviceArray(1 << 16); out = api.DeviceArray(8)
t = api.Timer()
def med(f):
    r = []
    f()
    for _ in range(7):
        t.start(); f(); t.stop(); r.append(t.ms())
    return float(np.median(r))
print("%-14s %18s %14s %14s %14s" % ("buffer", "address", "memset GB/s", "copy-into GB/s", "read GB/s"))
for name, p in bufs:
    w = med(lambda: hip.hipMemsetAsync(p, 0, 8 * n, None))
    c = med(lambda: L.dfl_dcopy(n // 2, src.ptr, p, None))   # 0.5 GiB read + 0.5 GiB write (int32 length limit)
    r = med(lambda: L.dfl_dnrm2(n // 2, p, out.ptr, work.ptr, None))
    print("%-14s %#18x %14.0f %14.0f %14.0f" % (name, p, 8 * n / w / 1e6, 8 * n / c / 1e6, 4 * n / r / 1e6), flush=True)
