"""Probe: compute time of ONE rank's share of the 10M-tet step at an 8-way partition (rank 0's local mesh incl. its halo
layer), run through the partitioned code path (owned rows, interior/boundary split, RCCL communicator of world size 1 with
an empty halo plan).  No inter-GPU latency is in it: t_1gpu / this = the ceiling of the 8-GPU strong-scaling factor.
Launch: python -m torch.distributed.run --nproc-per-node 1 --master-addr 127.0.0.1 tools/probe_rank_local.py [M] [parts] [slot-patch leaf]"""
import sys, os, time, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("PROBE_EARLY_INIT") == "1":   # reserve the library's device pool BEFORE torch / RCCL allocate anything
    from dedflow_amd import api as _api
    _api.hip().hipSetDevice(0)
    _api.lib().Init(0, None)
import torch, torch.distributed as dist
from dedflow_amd import api, dist as D, dist_bench
from dedflow_amd.meshgen import kuhn_cube, synthetic_fields

M = int(sys.argv[1]) if len(sys.argv) > 1 else 119
parts = int(sys.argv[2]) if len(sys.argv) > 2 else 8
if len(sys.argv) > 3:   # J-assembly patch size (nodes per slot-owner patch), default 7
    api.lib().DflSetSlotPatchParameters(int(sys.argv[3]), 200, 128)
its = 40
torch.cuda.set_device(0)
device = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1)
mesh = kuhn_cube(M, jitter=0.2)
wg, dwg = synthetic_fields(mesh)
epart = D.partition_rcb(mesh, parts)
owner = D.node_owner(mesh, epart, parts)
for r in ((0,) * int(os.environ["PROBE_REPEAT"]) if os.environ.get("PROBE_REPEAT") else (0, parts // 2)):
    lm = D.build_local(mesh, epart, owner, r, parts)
    P = api.Problem(lm.mesh, maxit=its, atol=0.0, rtol=0.0, quiet=True)
    L = api.lib()
    L.MatrixFSSetOwnedRows.argtypes = [C.c_void_p, C.c_int32]
    L.MatrixFSSetOwnedRows(C.cast(P.J, C.c_void_p), lm.n_owned)

    class Plan:   # empty halo plan on a 1-rank communicator, real owned / interior counts
        rank, world = 0, 1
        n_local, n_owned, n_interior = lm.l2g_node.size, lm.n_owned, lm.n_interior
        send_splits, recv_splits = [0], [0]
        send_all = torch.zeros(0, dtype=torch.int64, device=device)
        recv_all = torch.zeros(0, dtype=torch.int64, device=device)
        def exchange(self, x): pass
    comm = D.RcclSolverComm(Plan(), dist, device)
    comm.install(P.ksp)
    L.KrylovSetFusedNorm(P.ksp, 0 if os.environ.get("DFL_FUSED_NORM") == "0" else 1)   # as bench.py --gpus N does
    alloc = D.RawPointerViews(device)
    n, no, Ng = P.N, lm.n_owned, mesh.num_node
    wg_t, wg_p = dist_bench.device_vector(alloc, torch, device, 6 * n, D.localize_vector(wg, lm, Ng))
    dwg_t, dwg_p = dist_bench.device_vector(alloc, torch, device, 6 * n, D.localize_vector(dwg, lm, Ng))
    F_t, F_p = dist_bench.device_vector(alloc, torch, device, 6 * n)
    x_t, x_p = dist_bench.device_vector(alloc, torch, device, 6 * n)
    Pp = dist_bench._Ptr
    def step():
        P.assemble_system(Pp(wg_p), Pp(dwg_p), Pp(F_p), want_J=False)
        P.assemble_system(Pp(wg_p), Pp(dwg_p), None, want_J=True)
        F_t[3 * no:3 * n].zero_(); F_t[3 * n + no:4 * n].zero_(); x_t.zero_()
        return P.solve(Pp(x_p), Pp(F_p))
    if os.environ.get("PROBE_SLEEP"):
        torch.cuda.synchronize(); time.sleep(float(os.environ["PROBE_SLEEP"]))
    if os.environ.get("PROBE_WAIT") == "1":     # the driver's background wipe of what the set-up freed
        L.DflWaitDeviceMemoryQuiet.restype = C.c_double
        L.DflWaitDeviceMemoryQuiet.argtypes = [C.c_double]
        torch.cuda.synchronize()
        print("waited %.2f s for the device memory to be quiet" % L.DflWaitDeviceMemoryQuiet(30.0))
    for _ in range(3): step()
    K = 10
    batches = []
    for _ in range(5):   # 5 batches of 10 steps: the median batch is reported, the spread beside it
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(K): step()
        torch.cuda.synchronize(); batches.append(1e3 * (time.perf_counter() - t0) / K)
    ms = float(np.median(batches))
    if os.environ.get("PROBE_TAGS") == "1":      # in-library event timing per kernel class over 5 more steps (as bench.py does)
        L.DflProfileEnable.argtypes = [C.c_int]
        L.DflProfileCollect.argtypes = [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.DflProfileCollect.restype = C.c_int
        L.DflProfileEnable(1)
        for _ in range(5): step()
        torch.cuda.synchronize()
        for name, tag in (("spmv", 0), ("cgs_dots", 1), ("cgs_update", 2), ("pc_apply", 3), ("asm_lhs", 4), ("asm_rhs", 5)):
            tot, mn = C.c_double(), C.c_double()
            cnt = L.DflProfileCollect(tag, C.byref(tot), C.byref(mn))
            if cnt: print("   %-10s n=%4d avg %.4f ms  min %.4f ms" % (name, cnt, tot.value / cnt, mn.value))
        L.DflProfileEnable(0)
    print("rank %d of %d: %d local tets (%d owned nodes, %d interior), %d collectives/step on a 1-rank communicator: %.2f ms per step"
          % (r, parts, P.T, no, lm.n_interior, (comm.n_allreduce + comm.n_halo) // (5 * K + 3), ms) +
          " (batches %s)" % ", ".join("%.2f" % b for b in batches), flush=True)
    P.close()
dist.destroy_process_group()
