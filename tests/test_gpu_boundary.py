"""Level-B boundary: the reference's own entry points driven the way a host that keeps the reference's .c files would
drive them -- the live LHS scatter MatrixAddElemValueBlockedBatched (src/matrix.c:574-592,819-831) fed with element
blocks from the CPU oracle, and the whole matrix / preconditioner / Krylov chain on a MatrixFS that keeps the reference's
storage (four row-expanded CSR value arrays: zero rows, diagonal extraction, unfused PC tree, 4 x scalar SpMV, GMRES).
Tolerance as everywhere else: 1e-10 of the largest magnitude; integer structure bit-exact."""
import ctypes as C

import numpy as np
import pytest

from dedflow_amd.meshgen import kuhn_cube, synthetic_fields

pytestmark = pytest.mark.gpu
RTOL = 1e-10


@pytest.fixture(scope="module")
def api():
    from dedflow_amd import api as A
    A.lib()  # raises if the HIP library is missing: no fallback
    return A


def close(a, b, rtol=RTOL):
    scale = max(float(np.abs(b).max()), 1e-300)
    err = float(np.abs(a - b).max()) / scale
    return err <= rtol, err


def _oracle_elem_J(orc, m, wg, dwg):
    eJ = np.empty((m.num_tet, 576))
    for e in range(m.num_tet):
        eJ[e] = orc.elem_tensors(m.xg, m.ien[4 * e:4 * e + 4], m.num_node, wg, dwg)[1]
    return eJ


def _scatter_by_color(api, P, eJ, S):
    """The reference's color loop (assemble.cu:1559-1738): one MatrixAddElemValueBlockedBatched call per color batch."""
    L = api.lib()
    L.MatrixZero(P.J)
    ien_d = P.mesh.contents.device.contents.ien
    bind = P.mesh.contents.batch_ind
    for c in range(S.num_color):
        lo, hi = int(S.batch_offset[c]), int(S.batch_offset[c + 1])
        if hi == lo:
            continue
        val = api.DeviceArray.from_numpy(np.ascontiguousarray(eJ[S.batch_ind[lo:hi]]).reshape(-1))
        L.MatrixAddElemValueBlockedBatched(P.J, 4, hi - lo, bind + 4 * lo, ien_d, 6, 6, val.ptr, 6, 36, None)
        api.sync()


@pytest.mark.parametrize("reference_layout", [False, True])
def test_add_elem_value_blocked_batched_assembles_the_oracle_matrix(api, oracle_lib, reference_layout):
    m = kuhn_cube(4, jitter=0.2)
    wg, dwg = synthetic_fields(m)
    S = oracle_lib.System(m)
    vals = S.new_values()
    S.assemble_tet(wg, dwg, None, vals)
    eJ = _oracle_elem_J(oracle_lib, m, wg, dwg)
    P = api.Problem(m, reference_layout=reference_layout)
    try:
        assert np.array_equal(P.batch_ind(), S.batch_ind)
        _scatter_by_color(api, P, eJ, S)
        assert (api.lib().MatrixFSBlockValues(P.J) is None) == reference_layout
        for g, o in zip(P.export_values(), vals):
            ok, err = close(g, o)
            assert ok, err
        # a masked batch leaves masked elements out: mask the first element of color 0
        L = api.lib()
        L.MatrixZero(P.J)
        lo, hi = int(S.batch_offset[0]), int(S.batch_offset[1])
        mask = np.ones(hi - lo, np.int32)
        mask[0] = 0
        val = api.DeviceArray.from_numpy(np.ascontiguousarray(eJ[S.batch_ind[lo:hi]]).reshape(-1))
        mask_d = api.DeviceArray.from_numpy(mask)
        L.MatrixAddElemValueBlockedBatched(P.J, 4, hi - lo, P.mesh.contents.batch_ind + 4 * lo, P.mesh.contents.device.contents.ien,
                                           6, 6, val.ptr, 6, 36, mask_d.ptr)
        api.sync()
        total_masked = sum(float(np.abs(g).sum()) for g in P.export_values())
        L.MatrixZero(P.J)
        L.MatrixAddElemValueBlockedBatched(P.J, 4, hi - lo, P.mesh.contents.batch_ind + 4 * lo, P.mesh.contents.device.contents.ien,
                                           6, 6, val.ptr, 6, 36, None)
        api.sync()
        total_full = sum(float(np.abs(g).sum()) for g in P.export_values())
        assert total_masked < total_full
    finally:
        P.close()


def test_reference_layout_matrix_chain_matches_the_oracle(api, oracle_lib):
    """Non-block MatrixFS: scatter, Dirichlet rows through GetRowFromNode + MatrixZeroRow + GetNodeFromRow
    (dirichlet.c:47-61), MatrixGetDiag on the sub-matrices (MatrixGetDiagBlockGPU / MatrixCSRGetDiagGPU), the generic PC
    tree (PCJacobi bs=3 / bs=1 / PCNone x2, pc.c:44-147), scal + 4 x scalar SpMV (matrix.c:471-497) and GMRES."""
    m = kuhn_cube(4, jitter=0.2)
    wg, dwg = synthetic_fields(m)
    S = oracle_lib.System(m)
    F, vals = S.assemble_system(wg, dwg, True, True)
    # element blocks + face terms: the oracle's tet-only values scattered through the reference entry point, faces added on
    # the host into the same arrays afterwards (AssembleSystemTetFace needs block mode), then the Dirichlet rows on the GPU
    vt = S.new_values()
    S.assemble_tet(wg, dwg, None, vt)
    eJ = _oracle_elem_J(oracle_lib, m, wg, dwg)
    P = api.Problem(m, reference_layout=True, maxit=60, atol=0.0, rtol=0.0)
    L = api.lib()
    try:
        _scatter_by_color(api, P, eJ, S)
        vf = S.new_values()
        S.assemble_face(wg, dwg, None, vf)
        for slot, add in zip((0, 1, 4, 5), vf):
            csr = C.cast(P.fs.mat[slot].contents.data, C.POINTER(api.MatrixCSR)).contents
            cur = api.d2h(csr.val, add.size, np.float64)
            api.DeviceArray(add.size, np.float64, ptr=csr.val).upload(cur + add)
        for bc in P.bcs:
            L.DirichletApplyMat(bc, P.J)
        api.sync()
        got = P.export_values()
        for g, o in zip(got, vals):
            ok, err = close(g, o)
            assert ok, err
        # boundary node lists survive the node -> row -> node round trip of dirichlet.c:56-58
        for bc, (group, _) in zip(P.bcs, api.REFERENCE_BCS):
            nb = int(m.bound_node_offset[group + 1] - m.bound_node_offset[group])
            assert np.array_equal(api.d2h(bc.contents.buffer, nb, np.int32), S.bnodes(group))
        # matvec / AMVPBY
        x = np.random.default_rng(3).normal(size=6 * S.N)
        y0 = np.random.default_rng(4).normal(size=6 * S.N)
        x_d, y_d = api.DeviceArray.from_numpy(x), api.DeviceArray.from_numpy(y0)
        L.MatrixAMVPBY(P.J, 0.7, x_d.ptr, -1.3, y_d.ptr)
        api.sync()
        yo = y0.copy()
        S.amvpby(vals, 0.7, x, -1.3, yo)
        ok, err = close(y_d.numpy()[:4 * S.N], yo[:4 * S.N])
        assert ok, err
        assert np.array_equal(y_d.numpy()[4 * S.N:], y0[4 * S.N:])   # Q5: the phi / T tail is untouched
        # the generic PC nodes one by one (pc.c:44-114): block Jacobi on A00 (inv(D)^T, Q7), point Jacobi on A11
        d33, d1 = S.pc_setup(vals)
        zo = S.pc_apply(d33, d1, x)
        pc0, pc1 = L.PCCreateJacobi(P.fs.mat[0], 3, None), L.PCCreateJacobi(P.fs.mat[5], 1, None)
        z_d = api.DeviceArray(6 * S.N)
        L.PCSetup(pc0); L.PCSetup(pc1)
        L.PCApply(pc0, x_d.ptr, z_d.ptr)
        L.PCApply(pc1, x_d.ptr + 8 * 3 * S.N, z_d.ptr + 8 * 3 * S.N)
        api.sync()
        ok, err = close(z_d.numpy()[:4 * S.N], zo[:4 * S.N], 1e-9)   # closed-form vs pivoted-LU 3x3 inverse (Q8)
        assert ok, err
        L.PCDestroy(pc0); L.PCDestroy(pc1)
        # GMRES (the PC tree KrylovSolve builds, krylov.c:439-453)
        F_d, xs_d = api.DeviceArray.from_numpy(F), api.DeviceArray(6 * S.N)
        it, r0, hist, _ = P.solve(xs_d, F_d)
        xo, ho, r0o, ito = S.gmres(vals, F, maxit=60, atol=0.0, rtol=0.0)
        assert it == ito == 60
        assert abs(r0 - r0o) <= 1e-12 * r0o
        k = np.arange(1, it + 1)
        assert np.all(np.abs(hist - ho) <= 1e-10 * r0o * np.maximum(1.0, k / 10.0))
        ok, err = close(xs_d.numpy(), xo, 1e-8)
        assert ok, err
    finally:
        P.close()


def test_block_mode_zero_row_and_submatrix_views(api, oracle_lib):
    """MatrixZeroRow on the block-mode matrix applies the rows (it used to print and return); operations on the
    sub-matrix views act on the assembled values, not on a private zero array."""
    m = kuhn_cube(4, jitter=0.2)
    wg, dwg = synthetic_fields(m)
    S = oracle_lib.System(m)
    _, vals = S.assemble_system(wg, dwg, False, True)
    P = api.Problem(m)
    L = api.lib()
    try:
        wg_d, dwg_d = api.DeviceArray.from_numpy(wg), api.DeviceArray.from_numpy(dwg)
        L.MatrixZero(P.J)
        P.assemble_tet(wg_d, dwg_d, None, want_J=True)
        P.assemble_face(wg_d, dwg_d, None, want_J=True)
        for (group, bctype) in api.REFERENCE_BCS:            # dirichlet.c:54-59 by hand: rows node*3+ic, MatrixZeroRow
            bn = S.bnodes(group)
            for ic, t in enumerate(bctype):
                if t:
                    rows = api.DeviceArray.from_numpy((bn * 3 + ic).astype(np.int32))
                    L.MatrixZeroRow(P.J, bn.size, rows.ptr, 0, 1.0)
        api.sync()
        for g, o in zip(P.export_values(), vals):
            ok, err = close(g, o)
            assert ok, err
        # y = A00 x through the VIEW of the velocity block
        x = np.random.default_rng(5).normal(size=3 * S.N)
        x_d, y_d = api.DeviceArray.from_numpy(x), api.DeviceArray(3 * S.N)
        L.MatrixMatVec(P.fs.mat[0], x_d.ptr, y_d.ptr)
        api.sync()
        import scipy.sparse as sp
        A00 = sp.csr_matrix((vals[0], S.ci33, S.rp33), shape=(3 * S.N, 3 * S.N))
        ok, err = close(y_d.numpy(), A00 @ x)
        assert ok, err
    finally:
        P.close()


def test_masked_matvec(api, oracle_lib):
    """y = lm .* (A (rm .* x)) (MatrixMatVecWithMask, matrix.c:167-204,499-525) on both storage modes."""
    m = kuhn_cube(3, jitter=0.2)
    wg, dwg = synthetic_fields(m)
    S = oracle_lib.System(m)
    _, vals = S.assemble_system(wg, dwg, False, True)
    rng = np.random.default_rng(6)
    x = rng.normal(size=6 * S.N)
    lm = (rng.random(6 * S.N) > 0.3).astype(np.float64)
    rm = (rng.random(6 * S.N) > 0.3).astype(np.float64)
    want = np.zeros(6 * S.N)
    S.amvpby(vals, 1.0, rm * x, 0.0, want)
    want *= lm
    P = api.Problem(m)
    try:
        wg_d, dwg_d = api.DeviceArray.from_numpy(wg), api.DeviceArray.from_numpy(dwg)
        P.assemble_system(wg_d, dwg_d, None, want_J=True)
        x_d, y_d = api.DeviceArray.from_numpy(x), api.DeviceArray(6 * S.N)
        lm_d, rm_d = api.DeviceArray.from_numpy(lm), api.DeviceArray.from_numpy(rm)
        api.lib().MatrixMatVecWithMask(P.J, x_d.ptr, y_d.ptr, lm_d.ptr, rm_d.ptr)
        api.sync()
        ok, err = close(y_d.numpy()[:4 * S.N], want[:4 * S.N])
        assert ok, err
    finally:
        P.close()


def test_vec_kernels(api):
    """VecAXPY / VecPointwiseMult / Div / Inv (src/vec.cu:14-76), odd lengths and unaligned views included.  One rounding
    of slack for the fused multiply-add of the axpy and the device division."""
    ulp = 4.0 * np.finfo(np.float64).eps
    L = api.lib()
    rng = np.random.default_rng(7)
    for n, off in ((1001, 0), (4096, 1), (7, 3), (1, 0)):
        x = rng.normal(size=n + off)
        y = rng.normal(size=n + off) + 3.0
        x_d, y_d, z_d = api.DeviceArray.from_numpy(x), api.DeviceArray.from_numpy(y), api.DeviceArray(n + off)
        xv, yv, zv = x_d.view(off, n), y_d.view(off, n), z_d.view(off, n)
        L.VecAXPY(0.37, xv.ptr, yv.ptr, n)
        api.sync()
        assert np.all(np.abs(yv.numpy() - (0.37 * x[off:] + y[off:])) <= ulp * np.abs(y[off:]).max())
        y2 = yv.numpy()
        L.VecPointwiseMult(xv.ptr, yv.ptr, zv.ptr, n)
        api.sync()
        assert np.array_equal(zv.numpy(), x[off:] * y2)
        L.VecPointwiseDiv(xv.ptr, yv.ptr, zv.ptr, n)
        api.sync()
        assert np.all(np.abs(zv.numpy() - x[off:] / y2) <= ulp * np.abs(x[off:] / y2))
        L.VecPointwiseInv(yv.ptr, n)
        api.sync()
        assert np.all(np.abs(yv.numpy() - 1.0 / y2) <= ulp * np.abs(1.0 / y2))
