"""Probe: SpMV time per launch over long un-instrumented bursts (does sustained load change the clocks?)."""
import sys, os, ctypes as C, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dedflow_amd import api
from dedflow_amd.meshgen import kuhn_cube, synthetic_fields
M = int(sys.argv[1]) if len(sys.argv) > 1 else 119
mesh = kuhn_cube(M, jitter=0.2)
wg, dwg = synthetic_fields(mesh)
P = api.Problem(mesh)
wg_d, dwg_d = api.DeviceArray.from_numpy(wg), api.DeviceArray.from_numpy(dwg)
P.assemble_system(wg_d, dwg_d, None, want_J=True)
x = api.DeviceArray.from_numpy(np.random.default_rng(0).normal(size=6 * P.N)); y = api.DeviceArray(6 * P.N)
t = api.Timer()
for burst in (10, 10, 100, 1000, 1000, 10):
    P.matvec(x, y); api.sync(); time.sleep(0.2)
    t.start()
    for _ in range(burst): P.matvec(x, y)
    t.stop()
    print("burst of %4d launches: %.4f ms per SpMV" % (burst, t.ms() / burst), flush=True)
# bursts chained without idle: 20 groups of 50, each group timed
res = []
for g in range(20):
    t.start()
    for _ in range(50): P.matvec(x, y)
    t.stop(); res.append(round(t.ms() / 50, 4))
print("chained groups of 50:", res)
