"""Worker for the multi-rank tests (launched by torch.distributed.run, gloo backend).

mode "cpu": the distributed ALGORITHM (RCB partition, owner-first numbering, halo plan,
            owned-rows SpMV, all-reduced CGS) with the CPU oracle as local compute --
            runs in the GPU-less container.
mode "gpu": the PRODUCT path -- C host GMRES + HIP kernels per rank with the DflComm
            callbacks on torch.distributed -- several ranks sharing cuda:0 (gloo with host
            staging; RCCL refuses two ranks on one device).
Both compare against the global single-domain oracle solve.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def global_reference(mesh, its):
    from dedflow_amd.meshgen import synthetic_fields
    from oracle import orc
    S = orc.System(mesh)
    wg, dwg = synthetic_fields(mesh)
    F, vals = S.assemble_system(wg, dwg, True, True)
    x, hist, r0, it = S.gmres(vals, F, maxit=its, atol=0.0, rtol=0.0)
    return S, wg, dwg, F, vals, x, hist, r0


def run_cpu(rank, world, M, its):
    import torch
    import torch.distributed as dist
    from dedflow_amd import dist as D
    from dedflow_amd.meshgen import kuhn_cube
    from oracle import orc
    mesh = kuhn_cube(M, jitter=0.2)
    Sg, wg, dwg, Fg, valsg, xg_, histg, r0g = global_reference(mesh, its)
    epart = D.partition_rcb(mesh, world)
    assert np.bincount(epart, minlength=world).min() > 0
    owner = D.node_owner(mesh, epart, world)
    lm = D.build_local(mesh, epart, owner, rank, world)
    n, no, Ng = lm.l2g_node.size, lm.n_owned, mesh.num_node
    # bench.py's setup (mesh generated and partitioned once on rank 0, pieces scattered) hands this rank the same piece
    from dedflow_amd import dist_bench
    lm_s, wg_s, dwg_s, Ng_s, Tg_s = dist_bench.distribute_problem(M, 0.2, rank, world, dist)
    assert (Ng_s, Tg_s) == (mesh.num_node, mesh.num_tet) and lm_s.n_owned == no and lm_s.n_interior == lm.n_interior
    assert np.array_equal(lm_s.l2g_node, lm.l2g_node) and np.array_equal(lm_s.mesh.ien, lm.mesh.ien)
    assert np.array_equal(lm_s.mesh.xg, lm.mesh.xg) and np.array_equal(lm_s.ghost_owner, lm.ghost_owner)
    assert np.array_equal(wg_s, D.localize_vector(wg, lm, Ng)) and np.array_equal(dwg_s, D.localize_vector(dwg, lm, Ng))
    # ownership is a partition of the nodes
    cnt = torch.tensor([float(no)], dtype=torch.float64)
    dist.all_reduce(cnt)
    assert int(cnt.item()) == Ng
    S = orc.System(lm.mesh)
    F, vals = S.assemble_system(D.localize_vector(wg, lm, Ng), D.localize_vector(dwg, lm, Ng), True, True)
    plan = D.HaloPlan(lm, dist, torch.device("cpu"), False)
    own4 = np.concatenate([np.arange(3 * no), 3 * n + np.arange(no)])
    ghost4 = np.concatenate([np.arange(3 * no, 3 * n), 3 * n + np.arange(no, n)])
    gidx4 = np.concatenate([(3 * lm.l2g_node[:no, None] + np.arange(3)).reshape(-1), 3 * Ng + lm.l2g_node[:no]])

    # owned rows of the locally assembled F / matrix are complete (zero assembly communication)
    assert np.abs(F[own4] - Fg[gidx4]).max() <= 1e-10 * np.abs(Fg).max()

    def halo(v):
        t = torch.from_numpy(v[:4 * n])
        plan.exchange(t)

    def spmv_owned(x):
        y = S.matvec(vals, x)
        y[ghost4] = 0.0
        return y

    xr = np.random.default_rng(3).normal(size=6 * Ng)
    xl = D.localize_vector(xr, lm, Ng)
    xl[ghost4] = 1e30  # must be overwritten by the exchange
    halo(xl)
    assert np.array_equal(xl[ghost4], D.localize_vector(xr, lm, Ng)[ghost4])
    yl = spmv_owned(xl)
    yg = Sg.matvec(valsg, xr)
    assert np.abs(yl[own4] - yg[gidx4]).max() <= 1e-10 * np.abs(yg).max()

    # distributed right-preconditioned GMRES (mirror of dedflow_amd/host/solver.c)
    d33, d1 = S.pc_setup(vals)

    def pc(v):
        with np.errstate(all="ignore"):
            z = S.pc_apply(d33, d1, v)
        z[ghost4] = 0.0
        z[4 * n:] = 0.0
        return z

    def allsum(a):
        t = torch.from_numpy(np.atleast_1d(np.asarray(a, dtype=np.float64)).copy())
        dist.all_reduce(t)
        return t.numpy()

    b = F.copy()
    b[ghost4] = 0.0
    Q = np.zeros((its + 1, 6 * n))
    H = np.zeros((its + 1, its))
    r0 = np.sqrt(allsum(b @ b)[0])
    Q[0] = b / r0
    beta = np.zeros(its + 1)
    beta[0] = r0
    gv = np.zeros((its, 2))
    hist = []
    for k in range(its):
        z = pc(Q[k])
        halo(z)
        w = spmv_owned(z)
        h = allsum(Q[:k + 1] @ w)
        w = w - h @ Q[:k + 1]
        nr = np.sqrt(allsum(w @ w)[0])
        H[:k + 1, k] = h
        H[k + 1, k] = nr
        Q[k + 1] = w / nr
        for i in range(k):
            c, s = gv[i]
            a, bb = H[i, k], H[i + 1, k]
            H[i, k], H[i + 1, k] = c * a + s * bb, c * bb - s * a
        r = np.hypot(H[k, k], H[k + 1, k])
        c, s = H[k, k] / r, H[k + 1, k] / r
        gv[k] = (c, s)
        H[k, k], H[k + 1, k] = r, 0.0
        beta[k + 1] = -s * beta[k]
        beta[k] = c * beta[k]
        hist.append(abs(beta[k + 1]))
    hist = np.array(hist)
    assert abs(r0 - r0g) <= 1e-12 * r0g
    assert np.abs(hist - histg).max() <= 1e-8 * r0g, np.abs(hist - histg).max() / r0g
    y = np.linalg.solve(np.triu(H[:its, :its]), beta[:its])
    x = pc(y @ Q[:its])
    assert np.abs(x[own4] - xg_[gidx4]).max() <= 1e-6 * np.abs(xg_).max()
    if rank == 0:
        print("DIST_CPU_OK", world, no, n, len(plan.neighbours))


def run_gpu(rank, world, M, its):
    import torch
    import torch.distributed as dist
    from dedflow_amd import api, dist_bench
    from dedflow_amd import dist as D
    from dedflow_amd.meshgen import kuhn_cube
    mesh = kuhn_cube(M, jitter=0.2)
    Sg, wg, dwg, Fg, valsg, xg_, histg, r0g = global_reference(mesh, its)
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    # the bench's own setup path: mesh generated + partitioned once on rank 0, pieces scattered over a CPU group
    lm, wg_l, dwg_l, Ng_s, Tg_s = dist_bench.distribute_problem(M, 0.2, rank, world, dist)
    assert (Ng_s, Tg_s) == (mesh.num_node, mesh.num_tet)
    epart = D.partition_rcb(mesh, world)
    lm_ref = D.build_local(mesh, epart, D.node_owner(mesh, epart, world), rank, world)   # what a replicated build gives
    assert np.array_equal(lm.l2g_node, lm_ref.l2g_node) and np.array_equal(lm.mesh.ien, lm_ref.mesh.ien)
    assert lm.n_owned == lm_ref.n_owned and lm.n_interior == lm_ref.n_interior
    assert np.array_equal(wg_l, D.localize_vector(wg, lm_ref, mesh.num_node))
    lm, alloc, P, plan, comm = dist_bench.setup_rank(lm, rank, world, device, dist, its, True)
    fused = os.environ.get("DFL_FUSED_NORM") == "1"
    if fused:  # one all-reduce per Arnoldi step (norm from w.w - sum h^2)
        api.lib().KrylovSetFusedNorm(P.ksp, 1)
    pipelined = os.environ.get("DFL_PIPELINED") == "1"
    if pipelined:  # p(1)-pipelined GMRES: the reduction of a step overlaps the matvec of the next (looser history tolerance)
        api.lib().KrylovSetPipelined(P.ksp, 1)
    htol = 1e-6 if pipelined else 1e-8
    Ng, n, no = mesh.num_node, P.N, lm.n_owned
    wg_t, wg_p = dist_bench.device_vector(alloc, torch, device, 6 * n, D.localize_vector(wg, lm, Ng))
    dwg_t, dwg_p = dist_bench.device_vector(alloc, torch, device, 6 * n, D.localize_vector(dwg, lm, Ng))
    F_t, F_p = dist_bench.device_vector(alloc, torch, device, 6 * n)
    x_t, x_p = dist_bench.device_vector(alloc, torch, device, 6 * n)
    Pp = dist_bench._Ptr
    P.assemble_system(Pp(wg_p), Pp(dwg_p), Pp(F_p), want_J=False)
    P.assemble_system(Pp(wg_p), Pp(dwg_p), None, want_J=True)
    F_t[3 * no:3 * n].zero_()
    F_t[3 * n + no:4 * n].zero_()
    it, r0, hist, _ = P.solve(Pp(x_p), Pp(F_p))
    torch.cuda.synchronize()
    own4 = np.concatenate([np.arange(3 * no), 3 * n + np.arange(no)])
    gidx4 = np.concatenate([(3 * lm.l2g_node[:no, None] + np.arange(3)).reshape(-1), 3 * Ng + lm.l2g_node[:no]])
    Fl = F_t.cpu().numpy()
    assert np.abs(Fl[own4] - Fg[gidx4]).max() <= 1e-10 * np.abs(Fg).max()
    assert it == its and abs(r0 - r0g) <= 1e-12 * r0g
    assert np.abs(hist - histg).max() <= htol * r0g, np.abs(hist - histg).max() / r0g
    if rank == 0 and (os.environ.get("DFL_PRINT_DEV") or pipelined):
        print("HIST_DEV pipelined=%d max|hist-oracle|/r0 = %.3e" % (pipelined, np.abs(hist - histg).max() / r0g))
    if rank == 0 and os.environ.get("DFL_PRINT_DEV"):
        print("HIST_DEV fused=%d max|hist-oracle|/r0 = %.3e" % (fused, np.abs(hist - histg).max() / r0g))
    xl = x_t.cpu().numpy()
    assert np.abs(xl[own4] - xg_[gidx4]).max() <= 1e-6 * np.abs(xg_).max()
    # KrylovSolve itself clears the ghost rows of the residual (partial sums that belong to other ranks -- the local dots
    # run over all rows): the same solve from an F whose ghost rows the caller did NOT zero gives the same history
    P.assemble_system(Pp(wg_p), Pp(dwg_p), Pp(F_p), want_J=False)
    assert float(F_t[3 * no:3 * n].abs().max()) > 0.0
    x_t.zero_()
    it2, r02, hist2, _ = P.solve(Pp(x_p), Pp(F_p))
    torch.cuda.synchronize()
    assert it2 == its and r02 == r0 and np.array_equal(hist2, hist), (r02, r0, np.abs(hist2 - hist).max())
    dist.barrier()
    if fused or pipelined:
        st = api.lib().KrylovGetStats(P.ksp).contents
        assert not st.fused_norm_cancelled
        assert comm.n_allreduce <= 2 * (its + 2), comm.n_allreduce   # two solves: one per iteration + the initial residual each
    if rank == 0:
        print("DIST_GPU_OK", world, comm.n_allreduce, comm.n_halo)
    P.close()


def run_gpu_rccl(rank, world, M, its):
    """The C-level RCCL communicator on the nccl backend.  A GPU box has one GPU and RCCL refuses two ranks on one
    device, so this runs with world_size 1: bootstrap (unique id, ncclCommInitRank), in-place ncclAllReduce from the
    C GMRES loop, empty halo plan, verification against torch.distributed, and the solve equal to the plain
    single-GPU solve.  The multi-rank send/recv group itself is exercised only by bench.py --gpus N."""
    import torch
    import torch.distributed as dist
    from dedflow_amd import api, dist_bench
    from dedflow_amd import dist as D
    from dedflow_amd.meshgen import kuhn_cube
    assert dist.get_backend() == "nccl"
    mesh = kuhn_cube(M, jitter=0.2)
    Sg, wg, dwg, Fg, valsg, xg_, histg, r0g = global_reference(mesh, its)
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    os.environ["DFL_COMM"] = "rccl"
    lm, _, _, _, _ = dist_bench.distribute_problem(M, 0.2, rank, world, dist)
    lm, alloc, P, plan, comm = dist_bench.setup_rank(lm, rank, world, device, dist, its, False)
    assert type(comm).__name__ == "RcclSolverComm", type(comm).__name__
    Ng, n, no = mesh.num_node, P.N, lm.n_owned
    wg_t, wg_p = dist_bench.device_vector(alloc, torch, device, 6 * n, D.localize_vector(wg, lm, Ng))
    dwg_t, dwg_p = dist_bench.device_vector(alloc, torch, device, 6 * n, D.localize_vector(dwg, lm, Ng))
    F_t, F_p = dist_bench.device_vector(alloc, torch, device, 6 * n)
    x_t, x_p = dist_bench.device_vector(alloc, torch, device, 6 * n)
    Pp = dist_bench._Ptr
    P.assemble_system(Pp(wg_p), Pp(dwg_p), Pp(F_p), want_J=False)
    P.assemble_system(Pp(wg_p), Pp(dwg_p), None, want_J=True)
    it, r0, hist, _ = P.solve(Pp(x_p), Pp(F_p))
    torch.cuda.synchronize()
    assert it == its and abs(r0 - r0g) <= 1e-12 * r0g
    assert np.abs(hist - histg).max() <= 1e-8 * r0g
    assert comm.n_allreduce >= its and comm.n_halo >= its
    # p(1)-pipelined GMRES with the reduction on its own stream (the communicator is stream-ordered): same system
    api.lib().KrylovSetPipelined(P.ksp, 1)
    x_t.zero_()
    itp, r0p, histp, _ = P.solve(Pp(x_p), Pp(F_p))
    torch.cuda.synchronize()
    assert itp == its and abs(r0p - r0g) <= 1e-12 * r0g
    assert np.abs(histp - histg).max() <= 1e-6 * r0g, np.abs(histp - histg).max() / r0g
    xl = x_t.cpu().numpy()
    assert np.abs(xl[:4 * n] - xg_[:4 * Ng]).max() <= 1e-6 * np.abs(xg_).max()
    api.lib().KrylovSetPipelined(P.ksp, 0)
    print("DIST_RCCL_OK", world, comm.n_allreduce, comm.n_halo, "pipelined history dev %.2e" % (np.abs(histp - histg).max() / r0g))
    P.close()


def run_gpu_step(rank, world, M, its):
    """One coupled-free time step (predictor, <=2 Newton iterations with distributed GMRES, corrector) on the
    partitioned mesh against the single-domain oracle driver."""
    import torch
    import torch.distributed as dist
    from dedflow_amd import dist_bench
    from dedflow_amd import dist as D
    from dedflow_amd.meshgen import kuhn_cube, synthetic_fields
    from oracle import orc
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from ref_driver import time_step
    mesh = kuhn_cube(M, jitter=0.2)
    Sg = orc.System(mesh)
    Ng = mesh.num_node
    wg0, dw0 = synthetic_fields(mesh)
    wgold = wg0.copy()
    wgold[3 * Ng:4 * Ng] = 0.0
    dwgold = 0.1 * dw0
    dwg = dwgold.copy()
    it_o, rn_o, ri_o, wgold_o, dwgold_o, _ = time_step(Sg, wgold, dwgold, dwg, maxit=2)
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    lm, _, _, _, _ = dist_bench.distribute_problem(M, 0.2, rank, world, dist)
    lm, alloc, P, plan, comm = dist_bench.setup_rank(lm, rank, world, device, dist, 120, True)
    from dedflow_amd import api
    api.lib().KrylovDestroy(P.ksp)  # reference solver settings for the driver: GMRES(120), atol 1e-12, rtol 1e-4
    P.ksp = api.lib().KrylovCreateGMRES(120, 1e-12, 1e-4, None)
    api.lib().KrylovSetVerbose(P.ksp, 0)
    comm.install(P.ksp)
    n, no = P.N, lm.n_owned
    vecs = [dist_bench.device_vector(alloc, torch, device, 6 * n, D.localize_vector(v, lm, Ng)) for v in (wgold, dwgold, dwg)]
    F_t, F_p = dist_bench.device_vector(alloc, torch, device, 6 * n)
    dx_t, dx_p = dist_bench.device_vector(alloc, torch, device, 6 * n)
    Pp = dist_bench._Ptr
    it, rn, ri = P.time_step(Pp(vecs[0][1]), Pp(vecs[1][1]), Pp(vecs[2][1]), Pp(F_p), Pp(dx_p), newton_maxit=2)
    torch.cuda.synchronize()
    assert it == it_o
    assert np.allclose(ri, ri_o, rtol=1e-9, atol=1e-10 * ri_o.max()), (ri, ri_o)
    assert np.allclose(rn, rn_o, rtol=1e-5, atol=1e-8 * ri_o.max()), (rn, rn_o)
    # every local node (owned AND ghost copies) carries the global state
    for (t, _), ref in zip(vecs[:2], (wgold_o, dwgold_o)):
        loc = t.cpu().numpy()
        exp = D.localize_vector(ref, lm, Ng)
        assert np.abs(loc - exp).max() <= 1e-7 * np.abs(ref).max(), np.abs(loc - exp).max()
    dist.barrier()
    if rank == 0:
        print("DIST_STEP_OK", world, it)
    P.close()


def run_gpu_transient_twolevel(rank, world, M, its):
    """BASELINE config 5 in small: a TRANSIENT (its = number of time steps, 2 Newton iterations each) on the partitioned mesh
    with PC_TWOLEVEL under FGMRES, against the same transient on the whole mesh in one process (product path, same
    preconditioner, both solved to rtol 1e-8 so that the states are comparable): Newton iteration counts equal, every solve
    converged, initial Newton residuals and the states after the last step equal at solver accuracy."""
    import torch
    import torch.distributed as dist
    from dedflow_amd import api, dist_bench
    from dedflow_amd import dist as D
    from dedflow_amd.meshgen import kuhn_cube, synthetic_fields
    L = api.lib()
    mesh = kuhn_cube(M, jitter=0.2)
    Ng = mesh.num_node
    wg0, dw0 = synthetic_fields(mesh)
    wgold = wg0.copy()
    wgold[3 * Ng:4 * Ng] = 0.0
    dwgold = 0.1 * dw0
    dwg = dwgold.copy()
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)

    def new_solver(P, comm=None):
        L.KrylovDestroy(P.ksp)
        P.ksp = L.KrylovCreateGMRES(200, 1e-14, 1e-8, None)
        L.KrylovSetVerbose(P.ksp, 0)
        L.KrylovSetAggregateSize(P.ksp, 27)
        L.KrylovSetPCType(P.ksp, api.PC_TWOLEVEL)      # (the mesh reaches the solver through SolveFlowSystem)
        if comm is not None:
            comm.install(P.ksp)

    def stats(P):
        st = L.KrylovGetStats(P.ksp).contents
        return st.total_solves, st.total_converged, st.total_iterations

    # ---- the whole mesh in this process ----
    Pg = api.Problem(mesh, maxit=200, atol=0.0, rtol=1e-8)
    new_solver(Pg)
    g = [api.DeviceArray.from_numpy(v) for v in (wgold, dwgold, dwg)]
    Fg, dxg = api.DeviceArray(6 * Ng), api.DeviceArray(6 * Ng)
    ref = []
    for k in range(its):
        it, rn, ri = Pg.time_step(g[0], g[1], g[2], Fg, dxg, newton_maxit=2)
        api.sync()
        ref.append((it, rn, ri))
    ref_state = (g[0].numpy(), g[1].numpy())
    sg = stats(Pg)
    assert sg[0] == sg[1] == 2 * its, sg
    Pg.close()

    # ---- the partition ----
    lm, _, _, _, _ = dist_bench.distribute_problem(M, 0.2, rank, world, dist)
    lm, alloc, P, plan, comm = dist_bench.setup_rank(lm, rank, world, device, dist, 200, True)
    new_solver(P, comm)
    n = P.N
    vecs = [dist_bench.device_vector(alloc, torch, device, 6 * n, D.localize_vector(v, lm, Ng)) for v in (wgold, dwgold, dwg)]
    F_t, F_p = dist_bench.device_vector(alloc, torch, device, 6 * n)
    dx_t, dx_p = dist_bench.device_vector(alloc, torch, device, 6 * n)
    Pp = dist_bench._Ptr
    for k in range(its):
        it, rn, ri = P.time_step(Pp(vecs[0][1]), Pp(vecs[1][1]), Pp(vecs[2][1]), Pp(F_p), Pp(dx_p), newton_maxit=2)
        torch.cuda.synchronize()
        it_o, rn_o, ri_o = ref[k]
        assert it == it_o, (k, it, it_o)
        assert np.allclose(ri, ri_o, rtol=1e-6, atol=1e-9 * ref[0][2].max()), (k, ri, ri_o)
        assert np.allclose(rn, rn_o, rtol=1e-3, atol=1e-7 * ref[0][2].max()), (k, rn, rn_o)
    sp_ = stats(P)
    assert sp_[0] == sp_[1] == 2 * its, sp_
    assert abs(sp_[2] - sg[2]) <= max(4, int(0.25 * sg[2])), (sp_, sg)       # total iterations within 25 %
    for (t, _), r in zip(vecs[:2], ref_state):
        loc = t.cpu().numpy()
        exp = D.localize_vector(r, lm, Ng)
        assert np.abs(loc - exp).max() <= 1e-6 * np.abs(r).max(), np.abs(loc - exp).max() / np.abs(r).max()
    dist.barrier()
    if rank == 0:
        print("DIST_TRANSIENT_TWOLEVEL_OK", world, its, sp_[2], sg[2])
    P.close()


def run_gpu_twolevel(rank, world, M, its):
    """PC_TWOLEVEL on the element-partitioned matrix (aggregates per rank over owned nodes, replicated Galerkin coarse
    problem, rank-local DILU smoothing) against the same preconditioner on the whole mesh in one process: iteration count to
    rtol 1e-4 within 20 %, solution at solver accuracy (both at rtol 1e-8), coarse matrix = P^T A P of the GLOBAL matrix."""
    import ctypes as C
    import torch
    import torch.distributed as dist
    from dedflow_amd import api, dist_bench
    from dedflow_amd import dist as D
    from dedflow_amd.meshgen import kuhn_cube, synthetic_fields
    L = api.lib()
    mesh = kuhn_cube(M, jitter=0.2)
    wg, dwg = synthetic_fields(mesh)
    Ng = mesh.num_node
    wg[3 * Ng:4 * Ng] = 0.0
    dwg = 0.1 * dwg
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    agg_size = 27

    def new_solver(P, rtol, comm=None):
        L.KrylovDestroy(P.ksp)
        P.ksp = L.KrylovCreateGMRES(200, 0.0, rtol, None)
        L.KrylovSetVerbose(P.ksp, 0)
        L.KrylovSetMesh(P.ksp, P.mesh)
        L.KrylovSetCheckInterval(P.ksp, 1)
        L.KrylovSetAggregateSize(P.ksp, agg_size)
        L.KrylovSetPCType(P.ksp, api.PC_TWOLEVEL)
        if comm is not None:
            comm.install(P.ksp)

    # ---- the whole mesh in this process ----
    Pg = api.Problem(mesh, maxit=200, atol=0.0, rtol=1e-4)
    wg_d, dwg_d = api.DeviceArray.from_numpy(wg), api.DeviceArray.from_numpy(dwg)
    Fg_d, xg_d = api.DeviceArray(6 * Ng), api.DeviceArray(6 * Ng)
    Pg.assemble_system(wg_d, dwg_d, Fg_d, want_J=False)
    Pg.assemble_system(wg_d, dwg_d, None, want_J=True)
    ref = {}
    for rtol in (1e-4, 1e-8):
        new_solver(Pg, rtol)
        xg_d.zero()
        it, r0, hist, conv = Pg.solve(xg_d, Fg_d)
        assert conv, (rtol, it)
        ref[rtol] = (it, r0, xg_d.numpy())
    rpg, cig = Pg.pattern()
    valg = Pg.block_values().numpy().reshape(-1, 4, 4)
    Pg.close()

    # ---- the partition ----
    lm, wg_l, dwg_l, Ng_s, Tg_s = dist_bench.distribute_problem(M, 0.2, rank, world, dist)
    lm, alloc, P, plan, comm = dist_bench.setup_rank(lm, rank, world, device, dist, 200, True)
    n, no = P.N, lm.n_owned
    wg_t, wg_p = dist_bench.device_vector(alloc, torch, device, 6 * n, D.localize_vector(wg, lm, Ng))
    dwg_t, dwg_p = dist_bench.device_vector(alloc, torch, device, 6 * n, D.localize_vector(dwg, lm, Ng))
    F_t, F_p = dist_bench.device_vector(alloc, torch, device, 6 * n)
    x_t, x_p = dist_bench.device_vector(alloc, torch, device, 6 * n)
    Pp = dist_bench._Ptr
    P.assemble_system(Pp(wg_p), Pp(dwg_p), Pp(F_p), want_J=False)
    P.assemble_system(Pp(wg_p), Pp(dwg_p), None, want_J=True)
    own4 = np.concatenate([np.arange(3 * no), 3 * n + np.arange(no)])
    gidx4 = np.concatenate([(3 * lm.l2g_node[:no, None] + np.arange(3)).reshape(-1), 3 * Ng + lm.l2g_node[:no]])
    got = {}
    for rtol in (1e-4, 1e-8):
        new_solver(P, rtol, comm)
        x_t.zero_()
        it, r0, hist, conv = P.solve(Pp(x_p), Pp(F_p))
        torch.cuda.synchronize()
        assert conv, (rtol, it)
        pc = L.KrylovGetPC(P.ksp)
        assert pc and C.cast(pc, C.POINTER(C.c_int))[0] == api.PC_TWOLEVEL   # not the PC_ILU0 fall-back
        got[rtol] = (it, r0, x_t.cpu().numpy())
    it1, r01, _ = ref[1e-4]
    itp, r0p, _ = got[1e-4]
    assert abs(r0p - r01) <= 1e-12 * r01
    assert abs(itp - it1) <= max(2, int(round(0.2 * it1))), (itp, it1)
    xr, xp = ref[1e-8][2], got[1e-8][2]
    assert np.abs(xp[own4] - xr[gidx4]).max() <= 1e-6 * np.abs(xr).max(), np.abs(xp[own4] - xr[gidx4]).max() / np.abs(xr).max()
    # the replicated coarse matrix equals P^T A P of the GLOBAL matrix with the partition's aggregates, on every rank
    import scipy.sparse as sp
    pc = L.KrylovGetPC(P.ksp)
    nagg, cnnz, inner = C.c_int32(0), C.c_int32(0), C.c_int64(0)
    L.PCTwoLevelInfo(pc, C.byref(nagg), C.byref(cnnz), C.byref(inner))
    Nc = nagg.value
    agg_l = api.d2h(L.PCTwoLevelAggregates(pc), n, np.int32)
    agg_g = torch.full((Ng,), -1.0, dtype=torch.float64)
    agg_g[torch.from_numpy(lm.l2g_node[:no])] = torch.from_numpy(agg_l[:no].astype(np.float64))
    cnt = torch.zeros(Ng, dtype=torch.float64)
    cnt[torch.from_numpy(lm.l2g_node[:no])] = 1.0
    agg_g = torch.where(agg_g < 0, torch.zeros_like(agg_g), agg_g)
    dist.all_reduce(agg_g)
    dist.all_reduce(cnt)
    assert bool((cnt == 1.0).all())
    agg_g = agg_g.numpy().astype(np.int64)
    assert agg_g.min() == 0 and agg_g.max() == Nc - 1 and np.bincount(agg_g).max() <= agg_size
    # ghost nodes carry their owner's aggregate
    assert np.array_equal(agg_l, agg_g[lm.l2g_node])
    A = sp.bsr_matrix((valg, cig, rpg), shape=(4 * Ng, 4 * Ng)).tocsr()
    Pm = sp.csr_matrix((np.ones(4 * Ng), (np.arange(4 * Ng), np.repeat(agg_g, 4) * 4 + np.tile(np.arange(4), Ng))), shape=(4 * Ng, 4 * Nc))
    Ac_ref = (Pm.T @ A @ Pm).tocsr()
    Ac = L.PCTwoLevelCoarseMatrix(pc)
    fs = C.cast(Ac.contents.data, C.POINTER(api.MatrixFS)).contents
    spy = fs.spy1x1.contents
    assert spy.num_row == Nc
    crp, cci = api.d2h(spy.row_ptr, Nc + 1, np.int32), api.d2h(spy.col_ind, spy.nnz, np.int32)
    cval = api.d2h(L.MatrixFSBlockValues(Ac), spy.nnz * 16, np.float64).reshape(-1, 4, 4)
    Ac_dev = sp.bsr_matrix((cval, cci, crp), shape=(4 * Nc, 4 * Nc)).tocsr()
    assert abs(Ac_dev - Ac_ref).max() <= 1e-10 * abs(Ac_ref).max(), abs(Ac_dev - Ac_ref).max() / abs(Ac_ref).max()
    dist.barrier()
    if rank == 0:
        print("DIST_TWOLEVEL_OK", world, "iterations 1-rank %d, partitioned %d; aggregates %d" % (it1, itp, Nc))
    P.close()


if __name__ == "__main__":
    import torch.distributed as dist
    mode, M, its = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    if mode == "gpu_rccl":
        import torch
        torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl" if mode == "gpu_rccl" else "gloo", rank=rank, world_size=world)
    {"cpu": run_cpu, "gpu": run_gpu, "gpu_step": run_gpu_step, "gpu_rccl": run_gpu_rccl, "gpu_twolevel": run_gpu_twolevel,
     "gpu_transient_twolevel": run_gpu_transient_twolevel}[mode](rank, world, M, its)
    dist.barrier()
    dist.destroy_process_group()
