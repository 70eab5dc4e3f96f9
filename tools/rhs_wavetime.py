"""Developer probe: cycles and patches per persistent wave of the lane-per-tet residual kernel (DFL_RHS_WTIME=1): is the
static, strided share of the patches finished at the same time by every wave?  Usage: python tools/rhs_wavetime.py [M]"""
import ctypes as C, os, sys
import numpy as np
os.environ["DFL_RHS_WTIME"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dedflow_amd import api
from dedflow_amd.meshgen import kuhn_cube, synthetic_fields
M = int(sys.argv[1]) if len(sys.argv) > 1 else 119
mesh = kuhn_cube(M, jitter=0.2)
wg, dwg = synthetic_fields(mesh)
L = api.lib()
P = api.Problem(mesh, schedule=4)
wg_d, dwg_d = api.DeviceArray.from_numpy(wg), api.DeviceArray.from_numpy(dwg)
F_d = api.DeviceArray(6 * P.N)
t = api.Timer()
for _ in range(3):
    P.assemble_tet(wg_d, dwg_d, F_d, want_J=False)
api.sync(); t.start()
for _ in range(10):
    P.assemble_tet(wg_d, dwg_d, F_d, want_J=False)
t.stop()
print("AssembleSystemTet(F): %.3f ms" % (t.ms() / 10))
buf = np.zeros(8192, np.uint64)
L.dfl_rhs_wtime_fetch.restype = C.c_int
L.dfl_rhs_wtime_fetch.argtypes = [C.c_void_p, C.c_int]
assert L.dfl_rhs_wtime_fetch(buf.ctypes.data, buf.size) == 8192
w = buf.reshape(-1, 2).astype(np.float64)
w = w[w[:, 1] > 0]
c = w[:, 0]
print("waves %d: cycles mean %.0f min %.0f max %.0f; patches per wave mean %.1f" % (len(c), c.mean(), c.min(), c.max(), w[:, 1].mean()))
wgidx = np.arange(len(c)) // 4
slot = (wgidx >> 3) // 32          # which of the co-resident workgroups of a CU (launch order)
for k in range(int(slot.max()) + 1):
    print("  workgroups launched %d-th on their CU: mean %.0f" % (k, c[slot == k].mean()))
print("percentiles 0/25/50/75/100:", np.percentile(c, [0, 25, 50, 75, 100]).round())
P.close()
# per workgroup (= CU) and per XCD (workgroup b runs on XCD b % 8): is the spread between XCDs (static ranges) or between CUs?
nw = 8 if os.environ.get("DFL_RHS_WPB", "8") != "4" else 4
g = c[: len(c) // nw * nw].reshape(-1, nw).max(axis=1)
print("per workgroup (slowest wave): mean %.0f min %.0f max %.0f" % (g.mean(), g.min(), g.max()))
for x in range(8):
    gx = g[x::8]
    print("  XCD %d: %3d workgroups, mean %.0f min %.0f max %.0f" % (x, len(gx), gx.mean(), gx.min(), gx.max()))
