"""A/B of the residual-assembly paths at M (default 119): schedule 1 (colored launches), patch form (schedule 3, args
leaf:node_cap) and wave-per-patch form (schedule 4, args w:tets:nodes)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dedflow_amd import api
from dedflow_amd.meshgen import kuhn_cube, synthetic_fields

M = int(sys.argv[1]) if len(sys.argv) > 1 else 119
def parse(c):
    if c.startswith("w:"):
        v = [int(k) for k in c[2:].split(":")]
        return (4, v[0], v[1], v[2] if len(v) > 2 else 0)   # w:tets:nodes[:dfl_tune_asm bits] (64:64 -> lane-per-tet kernel; 32 = the 4-lane wave kernel, 64 = 2 waves/SIMD build)
    v = [int(k) for k in c.split(":")]
    return (3, v[0], v[1], v[2] if len(v) > 2 else 0)
cfgs = [(1, 0, 0, 0)] + [parse(c) for c in (sys.argv[2:] or ["64:64", "w:32:48", "w:64:64", "w:16:32"])]
mesh = kuhn_cube(M, jitter=0.2)
wg, dwg = synthetic_fields(mesh)
L = api.lib()
ref = None
for mode, leaf, cap, dbg in cfgs:
    L.dfl_tune_asm(dbg)
    if mode == 3:
        L.DflSetRhsPatchParameters(leaf, cap)
    if mode == 4:
        L.DflSetRhsWaveParameters(leaf, cap)
    P = api.Problem(mesh, schedule=mode)
    wg_d, dwg_d = api.DeviceArray.from_numpy(wg), api.DeviceArray.from_numpy(dwg)
    F_d = api.DeviceArray(6 * P.N)
    for rep in range(3):
        P.assemble_tet(wg_d, dwg_d, F_d, want_J=False)
    api.sync()
    t = api.Timer(); res = []
    for rep in range(7):
        api.sync(); t.start()
        P.assemble_tet(wg_d, dwg_d, F_d, want_J=False)
        t.stop(); res.append(t.ms())
    F_d2 = api.DeviceArray(6 * P.N)
    P.assemble_tet(wg_d, dwg_d, F_d2, want_J=False)
    api.sync()
    v = F_d2.numpy()
    if ref is None:
        ref = v
    print("mode %d leaf %d cap %d dbg %d: F assembly median %.3f ms (min %.3f)  rel diff vs mode 1: %.2e" %
          (mode, leaf, cap, dbg, float(np.median(res)), min(res), np.abs(v - ref).max() / np.abs(ref).max()), flush=True)
    P.close()
