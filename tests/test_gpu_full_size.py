"""Full-size checks (BASELINE.json configs 1 and 2/3 sizes) through size-independent properties --
the oracle would take minutes here, so nothing in this file calls it:
  coloring validity, batch structure, pattern size formula, SpMV linearity, Dirichlet rows,
  zero phi/T residual tail, non-increasing GMRES recurrence residual that equals the TRUE residual
  ||b - A x|| (one extra matvec), agreement of the two execution schedules."""
import numpy as np
import pytest

from dedflow_amd.meshgen import kuhn_cube, synthetic_fields

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api():
    from dedflow_amd import api as A
    A.lib()
    return A


def _nnz1(M):
    E = 3 * M * (M + 1) ** 2 + 3 * M * M * (M + 1) + M ** 3
    return (M + 1) ** 3 + 2 * E


import os

# 55 = 1M tets (configs 1, 3), 119 = 10M tets (config 2), 203 = 50M tets (config 5: ~1 min and ~60 GB of HBM on one GPU);
# DFL_FULLSIZE_M adds one more size on demand
_SIZES = [55, 119, 203] + ([int(os.environ["DFL_FULLSIZE_M"])] if os.environ.get("DFL_FULLSIZE_M") else [])


@pytest.mark.parametrize("M", _SIZES)
def test_full_size_properties(api, M):
    m = kuhn_cube(M, jitter=0.2)
    wg, dwg = synthetic_fields(m)
    its = 40
    P = api.Problem(m, maxit=its, atol=0.0, rtol=0.0)
    try:
        N, T = P.N, P.T
        assert P.nnz1 == _nnz1(M)  # SURVEY.md section 8 table (2 559 646 at M=55, 25 575 838 at M=119)
        rp, ci = P.pattern()
        assert rp[0] == 0 and rp[-1] == P.nnz1 and np.all(np.diff(rp) > 0) and np.all(np.diff(rp) <= 64)
        # --- coloring: valid (no node shared inside a color), complete, batches ascending inside a color
        color = P.color()
        nc = P.num_color
        assert color.min() == 0 and color.max() == nc - 1
        key = color.astype(np.int64)[:, None] * N + m.ien.reshape(-1, 4).astype(np.int64)
        assert np.unique(key.reshape(-1)).size == 4 * T, "two same-colored tets share a node"
        del key
        off, ind = P.batch_offset(), P.batch_ind()
        assert off[0] == 0 and off[-1] == T and np.array_equal(np.bincount(color, minlength=nc), np.diff(off))
        assert np.array_equal(color[ind], np.repeat(np.arange(nc), np.diff(off)))
        inner = np.ones(T, bool)
        inner[off[1:-1]] = False
        assert np.all(np.diff(ind)[inner[1:]] > 0)  # ascending element ids inside each color (stable copy_if)
        # --- assembly + Dirichlet
        wg_d, dwg_d = api.DeviceArray.from_numpy(wg), api.DeviceArray.from_numpy(dwg)
        F_d, x_d, y_d = api.DeviceArray(6 * N), api.DeviceArray(6 * N), api.DeviceArray(6 * N)
        P.assemble_system(wg_d, dwg_d, F_d, want_J=False)
        P.assemble_system(wg_d, dwg_d, None, want_J=True)
        F = F_d.numpy()
        assert np.all(np.isfinite(F)) and np.all(F[4 * N:] == 0.0)
        for group, bctype in api.REFERENCE_BCS:
            bn = m.bound_node[m.bound_node_offset[group]:m.bound_node_offset[group + 1]]
            for ic, t in enumerate(bctype):
                if t == 1:
                    assert np.all(F[bn * 3 + ic] == 0.0)
        # unit Dirichlet rows: A e_r picks column r; (A x)_r == x_r for BC rows, any x
        rng = np.random.default_rng(0)
        x = rng.normal(size=6 * N)
        x_d.upload(x)
        P.matvec(x_d, y_d)
        y = y_d.numpy()
        bn = m.bound_node[m.bound_node_offset[0]:m.bound_node_offset[1]]
        for ic in range(3):
            assert np.array_equal(y[bn * 3 + ic], x[bn * 3 + ic])
        # --- SpMV linearity
        z = rng.normal(size=6 * N)
        z_d, t_d = api.DeviceArray.from_numpy(z), api.DeviceArray(6 * N)
        P.matvec(z_d, t_d)
        Az = t_d.numpy()
        c_d = api.DeviceArray.from_numpy(2.0 * x - 3.0 * z)
        P.matvec(c_d, t_d)
        lin = t_d.numpy()[:4 * N] - (2.0 * y[:4 * N] - 3.0 * Az[:4 * N])
        assert np.abs(lin).max() <= 1e-11 * np.abs(y).max()
        # --- GMRES: recurrence residual non-increasing and equal to the true residual
        x_d.zero()
        it, r0, hist, _ = P.solve(x_d, F_d)
        assert it == its and np.all(np.diff(hist) <= 1e-12 * r0)
        P.matvec(x_d, y_d)
        true = np.linalg.norm(F[:4 * N] - y_d.numpy()[:4 * N])
        assert abs(true - hist[-1]) <= 1e-8 * r0, (true, hist[-1], r0)
        assert hist[-1] < 0.5 * r0
        vals_default = P.block_values().numpy()
    finally:
        P.close()
    if M == 55:  # the reference-order schedule gives the same matrix up to summation order
        P0 = api.Problem(m, schedule=0)
        try:
            wg_d, dwg_d = api.DeviceArray.from_numpy(wg), api.DeviceArray.from_numpy(dwg)
            P0.assemble_system(wg_d, dwg_d, None, want_J=True)
            v0 = P0.block_values().numpy()
            assert np.abs(v0 - vals_default).max() <= 1e-12 * np.abs(v0).max()
        finally:
            P0.close()


def test_config5_transient_50M_tets(api):
    """BASELINE config 5 on one GPU: 50M-tet mesh, transient loop.  Three generalized-alpha steps through DflTimeStep
    (src/main.c:535-565), two Newton iterations each: every linear solve converges to the reference's tolerance (rtol
    1e-4, main.c:406), the Newton residuals are finite and drop, and the device pool is flat after the first step (no
    per-step allocation).
    Preconditioner: PC_TWOLEVEL = PC_ILU0 (multicolor block-DILU) as the smoother + an aggregation coarse-grid
    correction (host/pc_twolevel.c), FGMRES outside.  PC_ILU0 alone is not enough at this size: ~410 full-GMRES
    iterations for the first solve, GMRES(80) stalls at 7e-3, and GMRES(320) under a cap of 1280 still missed 1e-4 on
    the third step (tools/probe_restart.py, profiles/r02_restart_M203.txt); with the coarse level the solves take
    40-80 iterations.  GMRES(80) under a cap of 240 keeps the two bases (Q and Z, 272 MB per column) at 44 GB."""
    import ctypes as C
    M = int(os.environ.get("DFL_CONFIG5_M", "203"))
    steps = 3
    m = kuhn_cube(M, jitter=0.2)
    wg, dwg = synthetic_fields(m)
    N = m.num_node
    wg[3 * N:4 * N] = 0.0   # main.c:118: the pressure slot of the state vector is zero
    L = api.lib()
    P = api.Problem(m, maxit=240, atol=1e-12, rtol=1e-4)
    try:
        L.KrylovSetPCType(P.ksp, api.PC_TWOLEVEL)
        L.KrylovSetRestart(P.ksp, 80)
        st = [api.DeviceArray.from_numpy(a) for a in (wg, 0.1 * dwg, 0.1 * dwg)]
        F, dx = api.DeviceArray(6 * N), api.DeviceArray(6 * N)
        used = []
        for s in range(steps):
            it, rn, r0 = P.time_step(st[0], st[1], st[2], F, dx, newton_maxit=2)
            api.sync()
            stats = L.KrylovGetStats(P.ksp).contents
            assert it == 2 and np.all(np.isfinite(rn)) and np.all(np.isfinite(r0))
            assert rn[0] < r0[0] and rn[1] < r0[1], (s, r0, rn)          # momentum and continuity residuals drop
            assert stats.total_solves == stats.total_converged == 2 * (s + 1), (s, stats.total_solves, stats.total_converged, stats.iterations)
            assert stats.total_iterations <= 100 * stats.total_solves, stats.total_iterations
            r, u = C.c_int64(0), C.c_int64(0)
            L.DflDevicePoolStats(C.byref(r), C.byref(u))
            used.append((r.value, u.value))
        assert used[1] == used[2] == used[0], used
        assert np.all(np.isfinite(st[0].numpy()))
    finally:
        P.close()


def test_cg_on_spd_csr_matrix(api):
    """KrylovCreateCG is a stub in the reference (krylov.c:42-51); here: PCG checked against scipy on an SPD
    matrix (graph Laplacian + I) stored in a plain MAT_TYPE_CSR matrix over the nodal pattern."""
    import ctypes as C
    import scipy.sparse as sp
    import scipy.sparse.linalg as spl
    m = kuhn_cube(10, jitter=0.2)
    P = api.Problem(m)
    L = api.lib()
    try:
        rp, ci = P.pattern()
        N = P.N
        val = -np.ones(ci.size)
        rows = np.repeat(np.arange(N), np.diff(rp))
        val[rows == ci] = (np.diff(rp) - 1.0) + 1.0  # degree + 1: strictly diagonally dominant => SPD
        A = L.MatrixCreateTypeCSR(P.spy1x1, None)
        L.MatrixZero(A)  # allocates the value array
        csr = C.cast(A.contents.data, C.POINTER(api.MatrixCSR)).contents
        api.DeviceArray(ci.size, np.float64, ptr=csr.val).upload(val)
        b = np.random.default_rng(1).normal(size=N)
        b_d, x_d = api.DeviceArray.from_numpy(b), api.DeviceArray(N)
        ksp = L.KrylovCreateCG(200, 0.0, 1e-12, None)
        L.KrylovSetVerbose(ksp, 0)
        L.KrylovSolve(ksp, A, x_d.ptr, b_d.ptr)
        api.sync()
        st = L.KrylovGetStats(ksp).contents
        ref = spl.spsolve(sp.csr_matrix((val, ci, rp), shape=(N, N)).tocsc(), b)
        assert st.converged and st.iterations < 200
        assert np.abs(x_d.numpy() - ref).max() <= 1e-9 * np.abs(ref).max()
        L.KrylovDestroy(ksp)
        L.MatrixDestroy(A)
    finally:
        P.close()
