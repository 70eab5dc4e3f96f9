"""Developer probe: time split of the patch LHS kernel (dfl_tune_asm bits: 1 skip element loop, 2 skip flush, 4 skip LDS atomics (mode 2), 8 write-only flush (mode 3)).
Schedule 4 (slot owner): 1 skip phase 2, 2 skip phase 1, 4 skip the block evaluation, 8 skip the store, 16 EARLY build, 32 every lane reads
tet record 0 (no LDS bank conflicts); any bit but 16 selects the probe build, which always overwrites (256 = probe build, nothing skipped).
Usage: dbg_patch.py M mode:leaf:cap[:tetcap] [comma-separated dfl_tune_asm values]"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dedflow_amd import api
from dedflow_amd.meshgen import kuhn_cube, synthetic_fields
M = int(sys.argv[1]) if len(sys.argv) > 1 else 119
cfg = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "2:64:320").split(":")]
mode, leaf, cap = cfg[:3]
mesh = kuhn_cube(M, jitter=0.2)
wg, dwg = synthetic_fields(mesh)
L = api.lib()
if mode == 4:
    L.DflSetSlotPatchParameters(leaf, cap, cfg[3] if len(cfg) > 3 else 208)
else:
    (L.DflSetPatchParameters if mode == 2 else L.DflSetRowPatchParameters)(leaf, cap)
P = api.Problem(mesh, schedule=mode)
wg_d, dwg_d = api.DeviceArray.from_numpy(wg), api.DeviceArray.from_numpy(dwg)
L.MatrixZero(P.J)
P.assemble_tet(wg_d, dwg_d, None, want_J=True)
api.sync()
REPS = int(os.environ.get('DFL_DBG_REPS', 20))
seq = ((256,) if mode == 4 else (0,)) if len(sys.argv) > 3 and sys.argv[3] == 'pmc' else tuple(int(v) for v in sys.argv[3].split(',')) if len(sys.argv) > 3 else ((256, 272, 257, 258, 259, 260, 262, 264, 268, 288, 256) if mode == 4 else (0, 8, 2, 10, 1, 3, 0))
for dbg in seq:
    L.dfl_tune_asm(dbg)
    for rep in range(3):
        P.assemble_tet(wg_d, dwg_d, None, want_J=True)
    t = api.Timer()
    api.sync(); t.start()
    for rep in range(REPS):
        P.assemble_tet(wg_d, dwg_d, None, want_J=True)
    t.stop()
    print("dbg %d: %.3f ms" % (dbg, t.ms() / REPS), flush=True)
