"""Developer probe: time split of the patch LHS kernel (dfl_tune_asm bits: 1 skip element loop, 2 skip flush, 4 skip LDS atomics)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dedflow_amd import api
from dedflow_amd.meshgen import kuhn_cube, synthetic_fields
M = int(sys.argv[1]) if len(sys.argv) > 1 else 119
leaf, cap = (int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "64:320").split(":"))
mesh = kuhn_cube(M, jitter=0.2)
wg, dwg = synthetic_fields(mesh)
L = api.lib()
L.DflSetPatchParameters(leaf, cap)
P = api.Problem(mesh, schedule=2)
wg_d, dwg_d = api.DeviceArray.from_numpy(wg), api.DeviceArray.from_numpy(dwg)
L.MatrixZero(P.J)
P.assemble_tet(wg_d, dwg_d, None, want_J=True)
api.sync()
for dbg in (0, 1, 2, 4, 6, 3):
    L.dfl_tune_asm(dbg)
    t = api.Timer(); res = []
    for rep in range(5):
        api.sync(); t.start()
        P.assemble_tet(wg_d, dwg_d, None, want_J=True)
        t.stop(); res.append(t.ms())
    print("dbg %d: %.3f ms" % (dbg, float(np.median(res))), flush=True)
