"""PC_TWOLEVEL (build-defined: block-DILU smoothing + aggregation coarse-grid correction, FGMRES outside): the pieces
against scipy (Galerkin coarse matrix = P^T A P, restriction / prolongation adjoint), the solve against the direct
solution of the same system, and the property that justifies it -- iteration counts that do not grow with the mesh."""
import ctypes as C

import numpy as np
import pytest

from dedflow_amd.meshgen import kuhn_cube, synthetic_fields

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api():
    from dedflow_amd import api as A
    A.lib()  # raises if the HIP library is missing: no fallback
    return A


def _setup(api, M, maxit=200, rtol=1e-8):
    m = kuhn_cube(M, jitter=0.2)
    wg, dwg = synthetic_fields(m)
    N = m.num_node
    wg[3 * N:4 * N] = 0.0
    P = api.Problem(m, maxit=maxit, atol=0.0, rtol=rtol)
    wg_d, dwg_d = api.DeviceArray.from_numpy(wg), api.DeviceArray.from_numpy(0.1 * dwg)
    F_d = api.DeviceArray(6 * N)
    P.assemble_system(wg_d, dwg_d, F_d, want_J=False)
    P.assemble_system(wg_d, dwg_d, None, want_J=True)
    return m, P, F_d


def _scipy_matrix(P, vals):
    import scipy.sparse as sp
    N = P.N
    rp, ci = P.pattern()
    blocks = P.block_values().numpy().reshape(-1, 4, 4)
    A = sp.bsr_matrix((blocks, ci, rp), shape=(4 * N, 4 * N)).tocsr()   # node-block ordering (node*4 + comp)
    return A


def test_twolevel_pieces_and_solve(api, oracle_lib):
    import scipy.sparse as sp
    import scipy.sparse.linalg as spl
    m, P, F_d = _setup(api, 10)
    L = api.lib()
    N = P.N
    try:
        L.KrylovSetAggregateSize(P.ksp, 27)
        L.KrylovSetPCType(P.ksp, api.PC_TWOLEVEL)
        x_d = api.DeviceArray(6 * N)
        it, r0, hist, conv = P.solve(x_d, F_d)
        assert conv and it <= 60, (it, conv)
        pc = L.KrylovGetPC(P.ksp)
        nagg, cnnz, inner = C.c_int32(0), C.c_int32(0), C.c_int64(0)
        L.PCTwoLevelInfo(pc, C.byref(nagg), C.byref(cnnz), C.byref(inner))
        Nc = nagg.value
        assert 0 < Nc < N and inner.value > 0
        agg = api.d2h(L.PCTwoLevelAggregates(pc), N, np.int32)
        assert agg.min() == 0 and agg.max() == Nc - 1 and np.bincount(agg).max() <= 27
        # Galerkin coarse matrix = P^T A P (block sums)
        A = _scipy_matrix(P, None)
        Pm = sp.csr_matrix((np.ones(4 * N), (np.arange(4 * N), np.repeat(agg, 4) * 4 + np.tile(np.arange(4), N))), shape=(4 * N, 4 * Nc))
        Ac_ref = (Pm.T @ A @ Pm).tocsr()
        Ac = L.PCTwoLevelCoarseMatrix(pc)
        fs = C.cast(Ac.contents.data, C.POINTER(api.MatrixFS)).contents
        spy = fs.spy1x1.contents
        crp, cci = api.d2h(spy.row_ptr, Nc + 1, np.int32), api.d2h(spy.col_ind, spy.nnz, np.int32)
        cval = api.d2h(L.MatrixFSBlockValues(Ac), spy.nnz * 16, np.float64).reshape(-1, 4, 4)
        Ac_dev = sp.bsr_matrix((cval, cci, crp), shape=(4 * Nc, 4 * Nc)).tocsr()
        assert abs(Ac_dev - Ac_ref).max() <= 1e-12 * abs(Ac_ref).max()
        # the solve: same solution as a direct solve of the assembled system (global [u | p] ordering on the device)
        perm = np.empty(4 * N, np.int64)
        for c in range(3):
            perm[np.arange(N) * 4 + c] = 3 * np.arange(N) + c
        perm[np.arange(N) * 4 + 3] = 3 * N + np.arange(N)
        b = F_d.numpy()[:4 * N][perm]
        xref = spl.spsolve(A.tocsc(), b)
        xdev = x_d.numpy()[:4 * N][perm]
        assert np.abs(xdev - xref).max() <= 1e-6 * np.abs(xref).max()
        # the recurrence residual of FGMRES is the true residual
        y_d = api.DeviceArray(6 * N)
        P.matvec(x_d, y_d)
        true = np.linalg.norm(F_d.numpy()[:4 * N] - y_d.numpy()[:4 * N])
        assert abs(true - hist[-1]) <= 1e-6 * r0, (true, hist[-1], r0)
    finally:
        P.close()


def test_twolevel_iterations_do_not_grow_with_the_mesh(api):
    """rtol 1e-4 (main.c:406): the DILU-preconditioned solve needs more iterations on the finer mesh, the two-level solve
    about the same number -- and far fewer."""
    L = api.lib()
    counts = {}
    for M in (16, 32):
        for pc in (api.PC_ILU0, api.PC_TWOLEVEL):
            m, P, F_d = _setup(api, M, maxit=300, rtol=1e-4)
            try:
                L.KrylovSetCheckInterval(P.ksp, 1)
                L.KrylovSetPCType(P.ksp, pc)
                x_d = api.DeviceArray(6 * P.N)
                it, r0, hist, conv = P.solve(x_d, F_d)
                assert conv
                counts[(M, pc)] = it
            finally:
                P.close()
    assert counts[(32, api.PC_TWOLEVEL)] <= counts[(16, api.PC_TWOLEVEL)] + 6, counts
    assert counts[(32, api.PC_ILU0)] >= counts[(16, api.PC_ILU0)] + 10, counts
    assert counts[(32, api.PC_TWOLEVEL)] * 2 < counts[(32, api.PC_ILU0)], counts


def test_lifecycles_do_not_leak(api):
    """Six problem lifecycles cycling through the three preconditioners, each with a particle context and three coupled time
    steps: the device pool returns to empty and free VRAM does not creep (tools/probe_lifecycle_twolevel.py, own process)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "probe_lifecycle_twolevel.py"), "20"], capture_output=True,
                       text=True, timeout=300, cwd=root)
    assert r.returncode == 0 and "LIFECYCLE_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
