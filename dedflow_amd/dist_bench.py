"""bench.py's N>1 leg: the same 10M-tet step, element-partitioned over WORLD_SIZE ranks
(strong scaling).  Launched by torch.distributed.run, one rank per GPU, RCCL ("nccl")."""
from __future__ import annotations

import ctypes as C
import json
import os
import sys
import time

import numpy as np


def distribute_problem(M, jitter, rank, world, dist):
    """The mesh is generated and partitioned ONCE, on rank 0, and every rank is shipped its own piece: the local mesh (local
    numbering, halo layer included), its maps and the localized state vectors.  No other rank ever holds the global mesh
    (8 ranks each building the 10M-tet mesh and partitioning it cost 8x the host memory and time of one).  The pieces
    travel as pickled objects over a CPU (gloo) group, so nothing of the setup touches RCCL.  DFL_DIST_SETUP=replicated:
    every rank builds and partitions the global mesh itself (the round-1 behaviour), for A/B.
    Returns (LocalMesh, wg_local, dwg_local, num_node_global, num_tet_global)."""
    from . import dist as D
    from .meshgen import kuhn_cube, synthetic_fields

    def pieces(ranks):
        mesh = kuhn_cube(M, jitter=jitter)
        wg, dwg = synthetic_fields(mesh)
        epart = D.partition_rcb(mesh, world)
        owner = D.node_owner(mesh, epart, world)
        out = []
        for r in ranks:
            lm = D.build_local(mesh, epart, owner, r, world)
            out.append((lm, D.localize_vector(wg, lm, mesh.num_node), D.localize_vector(dwg, lm, mesh.num_node),
                        mesh.num_node, mesh.num_tet))
        return out

    if world == 1 or os.environ.get("DFL_DIST_SETUP") == "replicated":
        return pieces([rank])[0]
    cpu = dist.new_group(backend="gloo") if dist.get_backend() != "gloo" else None
    objs = pieces(range(world)) if rank == 0 else None
    mine = [None]
    dist.scatter_object_list(mine, objs, src=0, group=cpu)
    return mine[0]


def setup_rank(lm, rank, world, device, dist, its, staged):
    """Build the local problem on this rank's GPU and wire the communicator."""
    from . import api
    from . import dist as D
    # device buffers stay in the C layer's own pool (host/runtime.c); the torch.distributed fallback views them by pointer
    alloc = D.TorchDeviceAllocator(device) if os.environ.get("DFL_TORCH_ALLOCATOR") == "1" else D.RawPointerViews(device)
    P = api.Problem(lm.mesh, maxit=its, atol=0.0, rtol=0.0, quiet=True)
    L = api.lib()
    L.MatrixFSSetOwnedRows.argtypes = [C.c_void_p, C.c_int32]
    L.MatrixFSSetOwnedRows(C.cast(P.J, C.c_void_p), lm.n_owned)
    plan = D.HaloPlan(lm, dist, device, staged)
    comm = None
    if not staged and os.environ.get("DFL_COMM", "rccl") == "rccl":
        # C-level RCCL communicator (no Python in the GMRES iteration); checked against torch.distributed on
        # this very plan before it is trusted, otherwise the torch.distributed callbacks stay in place
        try:
            cand = D.RcclSolverComm(plan, dist, device)
            if cand.verify():
                comm = cand
            elif rank == 0:
                print("dedflow: RCCL communicator failed verification, using torch.distributed callbacks", file=sys.stderr)
        except Exception as exc:  # noqa: BLE001
            print("dedflow: RCCL communicator unavailable (%r), using torch.distributed callbacks" % (exc,), file=sys.stderr)
    if comm is None:
        comm = D.DistSolverComm(plan, alloc, dist)
    comm.install(P.ksp)
    return lm, alloc, P, plan, comm


def device_vector(alloc, torch, device, n, init=None):
    """f64 device vector owned by torch (kept alive in alloc.blocks; comm callbacks view it by pointer)."""
    t = torch.zeros(8 * n + 16, dtype=torch.uint8, device=device)
    p = t.data_ptr()
    alloc.blocks[p] = t
    alloc.bases = sorted(alloc.blocks)
    v = t[:8 * n].view(torch.float64)
    if init is not None:
        v.copy_(torch.from_numpy(np.ascontiguousarray(init)))
    return v, p


class _Ptr:
    def __init__(self, p):
        self.ptr = p


def run(args, rank, world, local_rank):
    import torch
    import torch.distributed as dist
    from . import api
    from . import dist as D
    from .meshgen import kuhn_cube, synthetic_fields

    backend = os.environ.get("DFL_BACKEND", "nccl")
    if "DFL_FORCE_DEVICE" in os.environ:  # rehearsal of several ranks on one GPU (gloo only)
        local_rank = int(os.environ["DFL_FORCE_DEVICE"])
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world == 1:   # DFL_FORCE_DIST=1 rehearsal without a launcher: loopback rendezvous on a free port
        import socket
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
    dist.init_process_group(backend=backend, rank=rank, world_size=world)
    staged = backend != "nccl"

    t_setup = time.perf_counter()
    its = args.gmres_its
    lm, wg_l, dwg_l, Ng, Tg = distribute_problem(args.M, args.jitter, rank, world, dist)
    lm, alloc, P, plan, comm = setup_rank(lm, rank, world, device, dist, its, staged)
    n, no = P.N, lm.n_owned
    wg_t, wg_p = device_vector(alloc, torch, device, 6 * n, wg_l)
    dwg_t, dwg_p = device_vector(alloc, torch, device, 6 * n, dwg_l)
    F_t, F_p = device_vector(alloc, torch, device, 6 * n)
    x_t, x_p = device_vector(alloc, torch, device, 6 * n)
    del wg_l, dwg_l
    torch.cuda.synchronize()
    t_setup = time.perf_counter() - t_setup
    L = api.lib()

    def step():
        P.assemble_system(_Ptr(wg_p), _Ptr(dwg_p), _Ptr(F_p), want_J=False)
        P.assemble_system(_Ptr(wg_p), _Ptr(dwg_p), None, want_J=True)
        F_t[3 * no:3 * n].zero_()          # ghost entries of b are partial sums: not ours
        F_t[3 * n + no:4 * n].zero_()
        x_t.zero_()
        return P.solve(_Ptr(x_p), _Ptr(F_p))

    # Krylov work-space placement, as in the one-GPU line (bench.py --placement): the explicit calibration
    # before the warm-up, every rank for itself on its own GPU -- it contains no collective, and the library skips it below
    # 2^20 unknowns per rank (the 8-way share of the 10M-tet mesh).  Not when several ranks share one card (rehearsals).
    placement = getattr(args, "placement", "default")
    t_cal = 0.0
    if placement == "calibrate" and "DFL_FORCE_DEVICE" not in os.environ:
        L.DflKrylovCalibratePlacement.argtypes = [C.c_void_p, C.POINTER(api.Matrix), C.c_int64]
        L.DflKrylovCalibratePlacement.restype = None
        P.assemble_system(_Ptr(wg_p), _Ptr(dwg_p), None, want_J=True)     # (real values in the matrix the loop is timed on)
        tc = time.perf_counter()
        L.DflKrylovCalibratePlacement(P.ksp, P.J, int(getattr(args, "placement_cap_gb", 0.0) * 2 ** 30))
        torch.cuda.synchronize()
        t_cal = time.perf_counter() - tc
    dist.barrier()

    # one all-reduce per Arnoldi step (h and w.w together, KrylovSetFusedNorm) unless DFL_FUSED_NORM=0; the warm-up steps
    # double as the check: if any rank saw heavy cancellation in the Pythagorean norm, fall back to the explicit norm
    fused = os.environ.get("DFL_FUSED_NORM", "1") != "0"
    L.KrylovSetFusedNorm(P.ksp, 1 if fused else 0)
    for _ in range(args.warmup):
        step()
        if fused:
            flag = torch.tensor([float(L.KrylovGetStats(P.ksp).contents.fused_norm_cancelled)], dtype=torch.float64,
                                device="cpu" if staged else device)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            if flag.item() > 0:
                fused = False
                L.KrylovSetFusedNorm(P.ksp, 0)
    # timed region: EXACTLY --steps steps between barrier + synchronize on both sides, no instrumentation inside (the
    # in-library hipEvent profiler puts two event packets around every kernel)
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        it, r0, hist, _ = step()
    torch.cuda.synchronize()
    dist.barrier()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cpu" if staged else device)
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    t_total = float(dt.item())
    ms_per_step = 1e3 * t_total / args.steps
    # separate, untimed pass with the profiler on: rank 0's SpMV durations for the roofline figure
    L.DflProfileEnable(1)
    for _ in range(min(args.steps, 3)):
        step()
    torch.cuda.synchronize()

    # rank 0's local SpMV against the HBM roofline (same per-unit bytes as the 1-GPU line).  A partitioned matvec is the
    # interior rows on the library stream (while the halo is in flight) plus the boundary rows; with the C-level RCCL
    # communicator the boundary rows run on the halo stream beside the interior launch and are not inside the library
    # stream's event pairs: then the figure is the INTERIOR launch alone against the bytes of the interior rows
    tot, mn = C.c_double(0), C.c_double(0)
    n_spmv = L.DflProfileCollect(0, C.byref(tot), C.byref(mn))
    L.DflProfileEnable(0)
    rp, _ = P.pattern()
    split = 0 < lm.n_interior <= no
    side_rows = split and type(comm).__name__ == "RcclSolverComm" and os.environ.get("DFL_NO_SIDE_BOUNDARY_ROWS") is None
    nrow = lm.n_interior if side_rows else no
    spmv_bytes = 132.0 * float(rp[nrow]) + 4.0 * (nrow + 1) + 64.0 * nrow
    n_matvec = n_spmv if (side_rows or not split) else n_spmv // 2
    roofline = None
    if n_matvec and tot.value > 0:
        gbps = spmv_bytes * n_matvec / (tot.value * 1e-3) / 1e9
        what = ("interior rows (boundary rows overlap them on the halo stream)" if side_rows else
                "owned rows, interior + boundary launch" if split else "owned rows")
        roofline = {"kernel": "spmv (rank 0, %s)" % what, "bound": "hbm",
                    "achieved": gbps, "peak": 8000.0, "unit": "GB/s", "frac": gbps / 8000.0, "traffic": None,
                    "algorithmic_bytes_per_launch": spmv_bytes, "avg_launch_ms": tot.value / n_matvec}

    # ---- time to the reference's tolerance (rtol 1e-4, main.c:406) on the partitioned system: PC_TWOLEVEL (aggregates per
    # rank, replicated Galerkin coarse problem, rank-local DILU smoothing; FGMRES) -- outside the timed steps; every rank takes
    # part (the preconditioner's setup and application are collective)
    to_rtol = None
    n_extra_solves = 0
    n_steps_counted = args.steps + args.warmup + min(args.steps, 3)
    coll_allreduce, coll_halo = comm.n_allreduce // n_steps_counted, comm.n_halo // n_steps_counted   # before the extra leg
    if getattr(args, "solve_to_rtol", 1):
        ksp_i = C.cast(P.ksp, C.POINTER(C.c_int32))
        ksp_f = C.cast(P.ksp, C.POINTER(C.c_double))
        ksp_i[0] = 100
        ksp_f[1], ksp_f[2] = 1e-12, 1e-4
        L.KrylovSetFusedNorm(P.ksp, 0)
        L.KrylovSetPCType(P.ksp, api.PC_TWOLEVEL)
        res = []
        for rep in range(2):              # the first solve builds the aggregates and the replicated coarse matrix
            x_t.zero_()
            torch.cuda.synchronize(); dist.barrier(); tw = time.perf_counter()
            it3, r03, hist3, conv3 = P.solve(_Ptr(x_p), _Ptr(F_p))
            torch.cuda.synchronize(); dist.barrier(); res.append(time.perf_counter() - tw)
            n_extra_solves += 1
        pc = L.KrylovGetPC(P.ksp)
        is_tl = bool(pc) and C.cast(pc, C.POINTER(C.c_int))[0] == api.PC_TWOLEVEL
        nagg, cnnz, inner = C.c_int32(0), C.c_int32(0), C.c_int64(0)
        if is_tl:
            L.PCTwoLevelInfo(pc, C.byref(nagg), C.byref(cnnz), C.byref(inner))
        to_rtol = {"pc": "PC_TWOLEVEL on the partitioned matrix (rank-local block-DILU smoother + replicated aggregation coarse level, FGMRES)"
                         if is_tl else "PC_ILU0 fall-back (PC_TWOLEVEL unavailable)",
                   "rtol": 1e-4, "iterations": it3, "converged": bool(conv3), "ms": 1e3 * res[-1],
                   "first_solve_incl_hierarchy_build_ms": 1e3 * res[0], "coarse_nodes": nagg.value, "coarse_nnz": cnnz.value,
                   "relative_residual": float(hist3[-1] / r03) if len(hist3) else None, "dofs_per_s": 4.0 * Ng / res[-1],
                   "collectives_per_application": "1 halo exchange + 1 all-reduce of 4 x coarse_nodes doubles (+ the outer solver's own)"}
        L.KrylovSetPCType(P.ksp, api.PC_DECOMPOSITION)
        ksp_i[0] = its
        ksp_f[1], ksp_f[2] = 0.0, 0.0

    stats = torch.tensor([float(P.T), float(no), float(plan.bytes_per_exchange), float(P.num_color)], dtype=torch.float64,
                         device="cpu" if staged else device)
    gathered = [torch.zeros_like(stats) for _ in range(world)]
    dist.all_gather(gathered, stats)
    if rank == 0:
        per_rank = [[float(v) for v in g.tolist()] for g in gathered]
        out = {
            "metric": "assembled DOFs/s + Krylov-SpMV GB/s (%HBM peak), 10M-tet mesh",
            "value": 4.0 * Ng / (ms_per_step * 1e-3), "unit": "DOF/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"Kuhn cube M={args.M}: {Tg} tets, {Ng} nodes; step = AssembleSystem(F) + AssembleSystem(J) + "
                                   f"Jacobi-PC GMRES x{its} iterations; RCB element partition over {world} ranks, "
                                   f"one halo layer assembled redundantly, halo exchange + all-reduce on {backend}; mesh generated and "
                                   f"partitioned once on rank 0, local pieces scattered",
                       "gmres_its": its, "parallelism": f"dd{world}",
                       "placement": (f"explicit DflKrylovCalibratePlacement per rank before the warm-up ({t_cal:.1f} s on rank 0)"
                                     if t_cal > 0 else "library default (bounded pick inside the first solve)")},
            "per_rank": {"local_tets": [r[0] for r in per_rank], "owned_nodes": [r[1] for r in per_rank],
                         "halo_send_bytes": [r[2] for r in per_rank], "colors": [r[3] for r in per_rank]},
            "redundant_assembly_fraction": sum(r[0] for r in per_rank) / Tg - 1.0,
            "communicator": type(comm).__name__, "fused_norm_allreduce": bool(fused),
            "collectives_per_step": {"allreduce": coll_allreduce, "halo_exchange": coll_halo},
            "roofline": roofline, "cpu_baseline": None, "setup_s": t_setup, "solve_to_rtol": to_rtol,
            "gmres_residual_drop": float(hist[-1] / r0) if len(hist) else None,
        }
        sys.stdout.flush()
        fd = getattr(args, "_stdout_fd", None)   # bench.py keeps stdout for this one line (library chatter goes to stderr)
        if fd is None:
            print(json.dumps(out))
        else:
            os.write(fd, (json.dumps(out) + "\n").encode())
    dist.barrier()
    dist.destroy_process_group()
