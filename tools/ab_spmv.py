"""A/B timing of block-CSR SpMV kernel variants (dfl_tune key 0) on the 10M-tet matrix,
interleaved in one process (cdna_hip_programming.md rule 24)."""
import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dedflow_amd import api
from dedflow_amd.meshgen import kuhn_cube, synthetic_fields

M = int(sys.argv[1]) if len(sys.argv) > 1 else 119
mesh = kuhn_cube(M, jitter=0.2)
wg, dwg = synthetic_fields(mesh)
P = api.Problem(mesh)
L = api.lib()
L.dfl_tune.argtypes = [C.c_int, C.c_int]
wg_d, dwg_d = api.DeviceArray.from_numpy(wg), api.DeviceArray.from_numpy(dwg)
P.assemble_system(wg_d, dwg_d, None, want_J=True)
x = api.DeviceArray.from_numpy(np.random.default_rng(0).normal(size=6 * P.N))
y = api.DeviceArray(6 * P.N)
nbytes = 132.0 * P.nnz1 + 4.0 * (P.N + 1) + 64.0 * P.N
t = api.Timer()
VARS = [int(a) for a in os.environ["AB_SPMV_VARIANTS"].split(",")] if os.environ.get("AB_SPMV_VARIANTS") else list(range(5))
NV = len(VARS)
res = {v: [] for v in VARS}
ref = None
for rep in range(6):
    for v in VARS:
        L.dfl_tune(0, v)
        P.matvec(x, y)
        t.start()
        for _ in range(10):
            P.matvec(x, y)
        t.stop()
        res[v].append(t.ms() / 10)
for v in VARS:
    a = np.array(res[v][1:])
    print("variant %d (0-4: NT=%d U=%d; 4 = default XCD slabs; 12/13/14 = persistent, 8/16/4 workgroups per CU): median %.4f ms  min %.4f  -> %.0f GB/s (%.3f of 8 TB/s)" %
          (v, 1 if v == 4 else v & 1, 2 if v < 2 else 4, np.median(a), a.min(), nbytes / np.median(a) / 1e6, nbytes / np.median(a) / 1e6 / 8000))
L.dfl_tune(0, 4); P.matvec(x, y); api.sync(); y4 = y.numpy().copy()
for v in VARS:
    L.dfl_tune(0, v); y.zero(); P.matvec(x, y); api.sync()
    print("variant %d: max |y - y(default)| = %.3e" % (v, np.abs(y.numpy() - y4).max()))
L.dfl_tune(0, 4)
