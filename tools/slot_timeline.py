"""Developer probe: where a workgroup of the slot-owner J kernel spends the cycles of one patch.  The probe build stamps the
shader clock at eight points of the patch loop (dfl_tune_asm bit 2048; lane 0 of each wave of the first 32 workgroups, first
48 patches); this prints, per wave, the mean cycles between consecutive stamps.
  0 loop top  1 next patch's lists requested  2 phase 1 done  3 barrier passed  4 lists of the next patch arrived
  5 trips of pass 0 done  6 rows stored (both passes)  7 end-of-patch barrier passed
Usage: python tools/slot_timeline.py [M] [extra dfl_tune_asm bits]"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dedflow_amd import api
from dedflow_amd.meshgen import kuhn_cube, synthetic_fields

M = int(sys.argv[1]) if len(sys.argv) > 1 else 119
extra = int(sys.argv[2]) if len(sys.argv) > 2 else 0
mesh = kuhn_cube(M, jitter=0.2)
wg, dwg = synthetic_fields(mesh)
L = api.lib()
P = api.Problem(mesh, schedule=4)
wg_d, dwg_d = api.DeviceArray.from_numpy(wg), api.DeviceArray.from_numpy(dwg)
L.MatrixZero(P.J)
P.assemble_tet(wg_d, dwg_d, None, want_J=True)
api.sync()
L.dfl_tune_asm(256 | 2048 | extra)
for _ in range(3):
    P.assemble_tet(wg_d, dwg_d, None, want_J=True)
api.sync()
WG, IT, PT = 32, 48, 10
buf = np.zeros(WG * IT * 4 * PT, np.uint64)
L.dfl_slot_trace_fetch.restype = C.c_int
L.dfl_slot_trace_fetch.argtypes = [C.c_void_p, C.c_int]
n = L.dfl_slot_trace_fetch(buf.ctypes.data, buf.size)
assert n == buf.size, n
t = buf.reshape(WG, IT, 4, PT).astype(np.int64)
wt = np.zeros(4096, np.uint64)
L.dfl_slot_wgtime_fetch.restype = C.c_int
L.dfl_slot_wgtime_fetch.argtypes = [C.c_void_p, C.c_int]
L.dfl_slot_wgtime_fetch(wt.ctypes.data, wt.size)
wt = wt.reshape(-1, 2).astype(np.float64)
wt = wt[wt[:, 1] > 0]
print("workgroups %d: cycles in the patch loop mean %.0f  min %.0f  max %.0f; patches per workgroup mean %.1f max %.0f; cycles per patch mean %.0f"
      % (len(wt), wt[:, 0].mean(), wt[:, 0].min(), wt[:, 0].max(), wt[:, 1].mean(), wt[:, 1].max(), (wt[:, 0] / wt[:, 1]).mean()))
if os.environ.get("SLOT_WGTIME_OUT"):
    np.save(os.environ["SLOT_WGTIME_OUT"], wt)
by_xcd = [wt[i::8, 0].mean() for i in range(8)]
print("mean cycles by XCD (blockIdx % 8):", " ".join("%.0f" % v for v in by_xcd))
L.dfl_tune_asm(0)
names = ["top->lists requested", "phase 1 (gather + record)", "barrier 1", "wait for next lists", "pass 0 trips", "exchange + stores (+ pass 1)",
         "end barrier", "(next top)"]
sel = t[:, 8:40]                                   # steady state
per_patch = (sel[:, 1:, :, 0] - sel[:, :-1, :, 0]).astype(np.float64)
print("cycles per patch per workgroup (loop top to loop top): mean %.0f  median %.0f" % (per_patch.mean(), np.median(per_patch)))
for w in range(4):
    d = [np.mean(sel[:, :, w, k + 1] - sel[:, :, w, k]) for k in range(7)]
    print("wave %d: " % w + "  ".join("%s %.0f" % (names[k], d[k]) for k in range(7)))
P.close()
