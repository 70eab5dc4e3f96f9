"""Time of one PC_ILU0 (multicolor block-DILU) setup and apply next to one SpMV on the same matrix."""
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
wg_d, dwg_d = api.DeviceArray.from_numpy(wg), api.DeviceArray.from_numpy(dwg)
P.assemble_system(wg_d, dwg_d, None, want_J=True)
pc = L.PCCreateDILU(P.J)
x = api.DeviceArray.from_numpy(np.random.default_rng(0).normal(size=6 * P.N)); y = api.DeviceArray(6 * P.N)
t = api.Timer()
def grp(fn, n=10):
    fn(); t.start()
    for _ in range(n): fn()
    t.stop(); return t.ms() / n
L.PCSetup(pc)
print("node colors:", L.PCDILUGetColors(pc, None))
print("setup  %.3f ms" % grp(lambda: L.PCSetup(pc), 5))
print("apply  %.3f ms" % grp(lambda: L.PCApply(pc, x.ptr, y.ptr)))
print("SpMV   %.3f ms" % grp(lambda: P.matvec(x, y)))
L.PCDestroy(pc); P.close()
