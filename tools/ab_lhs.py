"""A/B of the Jacobian-assembly schedules (1 = compact colors, 2 = tet patches, 3 = row-owner node patches, 4 = slot-owner
node patches) at M (default 119); extra args are mode:leaf:cap[:tetcap] tuples.  AB_OVERWRITE=1 times the
overwrite form AssembleSystem uses for schedules 3 and 4 (val = assembled rows: no MatrixZero, no read of the old values)
instead of MatrixZero + accumulate; the first config (schedule 1) is then skipped."""
import ctypes as C, sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dedflow_amd import api
from dedflow_amd.meshgen import kuhn_cube, synthetic_fields

M = int(sys.argv[1]) if len(sys.argv) > 1 else 119
def parse(c):
    v = [int(k) for k in c.split(":")]
    return (v[0], v[1], v[2], v[3] if len(v) > 3 else 208)
configs = [(1, 0, 0, 0)] + [parse(c) for c in sys.argv[2:]] if len(sys.argv) > 2 else [(1, 0, 0, 0), (3, 16, 255, 0), (4, 16, 255, 208)]
mesh = kuhn_cube(M, jitter=0.2)
wg, dwg = synthetic_fields(mesh)
L = api.lib()
OVERWRITE = os.environ.get("AB_OVERWRITE") == "1"
if OVERWRITE:
    L.DflAssembleSystemTetBeta.restype = None
    L.DflAssembleSystemTetBeta.argtypes = [C.POINTER(api.Mesh3D), C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(api.Matrix), C.c_double]
    configs = [c for c in configs if c[0] >= 3]
ref = None
for mode, leaf, cap, tcap in configs:
    if mode == 2:
        L.DflSetPatchParameters(leaf, cap)
    if mode == 3:
        L.DflSetRowPatchParameters(leaf, cap)
    if mode == 4:
        L.DflSetSlotPatchParameters(leaf, cap, tcap)
    t0 = time.perf_counter()
    P = api.Problem(mesh, schedule=mode)
    wg_d, dwg_d = api.DeviceArray.from_numpy(wg), api.DeviceArray.from_numpy(dwg)
    L.MatrixZero(P.J)
    P.assemble_tet(wg_d, dwg_d, None, want_J=True)   # builds the schedule on first use
    api.sync()
    t_setup = time.perf_counter() - t0
    t = api.Timer()
    res = []
    for rep in range(9 if OVERWRITE else 5):
        if not OVERWRITE:
            L.MatrixZero(P.J)
        api.sync()
        t.start()
        if OVERWRITE:
            L.DflAssembleSystemTetBeta(P.mesh, wg_d.ptr, dwg_d.ptr, None, P.J, 0.0)
        else:
            P.assemble_tet(wg_d, dwg_d, None, want_J=True)
        t.stop()
        res.append(t.ms())
    L.MatrixZero(P.J)
    P.assemble_tet(wg_d, dwg_d, None, want_J=True)
    api.sync()
    v = P.block_values().numpy()
    if ref is None:
        ref = v
    err = np.abs(v - ref).max() / np.abs(ref).max()
    ms = float(np.median(res))
    print("mode %d leaf %d cap %d tcap %d: J assembly median %.3f ms (min %.3f)  %.3g DOF/s  setup %.1f s  rel diff vs mode 1: %.2e" %
          (mode, leaf, cap, tcap, ms, min(res), 4.0 * P.N / (ms * 1e-3), t_setup, err), flush=True)
    P.close()
