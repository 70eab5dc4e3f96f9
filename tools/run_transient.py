"""Transient loop in the shape of BASELINE config 5 on ONE GPU: M-cube mesh, PC_ILU0 (multicolor block-DILU) or the Jacobi tree,
`steps` generalized-alpha time steps through DflTimeStep; prints per-step time, Newton/GMRES work and the device-memory footprint.
  python tools/run_transient.py [M=203] [steps=5] [pc=dilu|jacobi|twolevel] [newton=2] [gmres_maxit=120] [restart=0]"""
import sys, os, time, json, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dedflow_amd import api
from dedflow_amd.meshgen import kuhn_cube, synthetic_fields

M = int(sys.argv[1]) if len(sys.argv) > 1 else 203
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
pc = sys.argv[3] if len(sys.argv) > 3 else "dilu"
newton = int(sys.argv[4]) if len(sys.argv) > 4 else 2
maxit = int(sys.argv[5]) if len(sys.argv) > 5 else 120
restart = int(sys.argv[6]) if len(sys.argv) > 6 else 0
t0 = time.perf_counter()
mesh = kuhn_cube(M, jitter=0.2)
wg, dwg = synthetic_fields(mesh)
N = mesh.num_node
wg[3 * N:4 * N] = 0.0
L = api.lib()
P = api.Problem(mesh, maxit=maxit, atol=1e-12, rtol=1e-4, quiet=True)
L.KrylovSetPCType(P.ksp, {"dilu": api.PC_ILU0, "twolevel": api.PC_TWOLEVEL}.get(pc, api.PC_DECOMPOSITION))
if os.environ.get("DFL_AGG"):
    L.KrylovSetAggregateSize(P.ksp, int(os.environ["DFL_AGG"]))
L.KrylovSetRestart(P.ksp, restart)
if os.environ.get("DFL_TRANSIENT_VERBOSE") == "1":
    L.KrylovSetVerbose(P.ksp, 1)
st = [api.DeviceArray.from_numpy(a) for a in (wg, 0.1 * dwg, 0.1 * dwg)]
F, dx = api.DeviceArray(6 * N), api.DeviceArray(6 * N)
api.sync()
print("setup %.1f s: %d tets, %d nodes, nnz1 %d" % (time.perf_counter() - t0, mesh.num_tet, N, P.nnz1), flush=True)
L.DflDevicePoolStats.argtypes = [C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
times = []
for s in range(steps):
    api.sync(); t = time.perf_counter()
    it, rn, r0 = P.time_step(st[0], st[1], st[2], F, dx, newton_maxit=newton)
    api.sync(); times.append(time.perf_counter() - t)
    stt = L.KrylovGetStats(P.ksp).contents
    print("step %d: %.1f ms, %d Newton iterations, last GMRES %d its (converged=%d), solves so far %d (converged %d, %d iterations), |R_u| %.3e -> %.3e" %
          (s + 1, 1e3 * times[-1], it, stt.iterations, stt.converged, stt.total_solves, stt.total_converged, stt.total_iterations, r0[0], rn[0]), flush=True)
r, u = C.c_int64(0), C.c_int64(0)
L.DflDevicePoolStats(C.byref(r), C.byref(u))
free_b, tot_b = C.c_size_t(0), C.c_size_t(0)
api.hip().hipMemGetInfo(C.byref(free_b), C.byref(tot_b))
print(json.dumps({"M": M, "tets": mesh.num_tet, "pc": pc, "steps": steps, "ms_per_step_mean": 1e3 * float(np.mean(times[1:] or times)),
                  "pool_reserved_GiB": r.value / 2**30, "pool_in_use_GiB": u.value / 2**30,
                  "device_used_GiB": (tot_b.value - free_b.value) / 2**30}))
P.close()
