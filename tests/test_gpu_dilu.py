"""PC_ILU0 = multicolor block-DILU (host/pc_dilu.c, csrc/k_dilu.hip).  Build-defined (the reference only declares the
enum value), so the checker is a dense numpy restatement of the formula written here:
    E_i = A_ii - sum_{j~i, color(j)<color(i)} A_ij E_j^-1 A_ji ;  M = (E+L) E^-1 (E+U) ;  z = M^-1 r
on the oracle-assembled system.  Parity unpinned by construction (no reference implementation exists)."""
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


def node_dofs(i, N):
    return np.array([3 * i, 3 * i + 1, 3 * i + 2, 3 * N + i])


def dilu_numpy(A, rp, ci, N, color, r):
    A = A.toarray()
    blk = lambda i, j: A[np.ix_(node_dofs(i, N), node_dofs(j, N))]
    order = np.argsort(color, kind="stable")
    Einv = np.zeros((N, 4, 4))
    for i in order:
        E = blk(i, i).copy()
        for j in ci[rp[i]:rp[i + 1]]:
            if j != i and color[j] < color[i]:
                E -= blk(i, j) @ Einv[j] @ blk(j, i)
        Einv[i] = np.linalg.inv(E)
    z = np.zeros(4 * N)
    for i in order:  # forward
        acc = r[node_dofs(i, N)].copy()
        for j in ci[rp[i]:rp[i + 1]]:
            if color[j] < color[i]:
                acc -= blk(i, j) @ z[node_dofs(j, N)]
        z[node_dofs(i, N)] = Einv[i] @ acc
    for i in order[::-1]:  # backward
        acc = np.zeros(4)
        for j in ci[rp[i]:rp[i + 1]]:
            if color[j] > color[i]:
                acc += blk(i, j) @ z[node_dofs(j, N)]
        z[node_dofs(i, N)] -= Einv[i] @ acc
    return Einv, z


def test_dilu_setup_and_apply_match_dense_restatement(api, oracle_lib):
    m = kuhn_cube(4, jitter=0.2)
    S = oracle_lib.System(m)
    wg, dwg = synthetic_fields(m)
    _, vals = S.assemble_system(wg, dwg, False, True)
    A = S.to_scipy(vals)
    N = S.N
    P = api.Problem(m)
    L = api.lib()
    try:
        wg_d, dwg_d = api.DeviceArray.from_numpy(wg), api.DeviceArray.from_numpy(dwg)
        P.assemble_system(wg_d, dwg_d, None, want_J=True)
        pc = L.PCCreateDILU(P.J)
        assert pc
        L.PCSetup(pc)
        color = np.zeros(N, np.uint8)
        nc = L.PCDILUGetColors(pc, color.ctypes.data)
        rp, ci = P.pattern()
        # a valid coloring of the nodal graph, all rows colored
        assert nc == int(color.max()) + 1 and nc <= 20
        for i in range(N):
            nb = ci[rp[i]:rp[i + 1]]
            assert np.all(color[nb[nb != i]] != color[i])
        rng = np.random.default_rng(3)
        r = rng.normal(size=6 * N)
        Einv_ref, z_ref = dilu_numpy(A, rp, ci, N, color.astype(int), r[:4 * N])
        Einv = api.DeviceArray(16 * N, np.float64, ptr=L.PCDILUGetInverseBlocks(pc)).numpy().reshape(N, 4, 4)
        assert np.abs(Einv - Einv_ref).max() <= 1e-9 * np.abs(Einv_ref).max()
        r_d, z_d = api.DeviceArray.from_numpy(r), api.DeviceArray(6 * N)
        L.PCApply(pc, r_d.ptr, z_d.ptr)
        api.sync()
        z = z_d.numpy()
        assert np.abs(z[:4 * N] - z_ref).max() <= 1e-10 * np.abs(z_ref).max()
        assert np.array_equal(z[4 * N:], r[4 * N:])      # phi / T sections: identity (PCNone)
        L.PCDestroy(pc)
    finally:
        P.close()


def test_gmres_with_dilu_needs_fewer_iterations_than_jacobi(api, oracle_lib):
    import scipy.sparse.linalg as spla
    m = kuhn_cube(10, jitter=0.2)
    S = oracle_lib.System(m)
    wg, dwg = synthetic_fields(m)
    F, vals = S.assemble_system(wg, dwg, True, True)
    A = S.to_scipy(vals).tocsc()
    N = S.N
    x_ref = spla.spsolve(A, F[:4 * N])
    its = {}
    for name, pctype in (("jacobi", api.PC_DECOMPOSITION), ("dilu", api.PC_ILU0)):
        P = api.Problem(m, maxit=400, atol=0.0, rtol=1e-9)
        L = api.lib()
        try:
            L.KrylovSetCheckInterval(P.ksp, 1)
            L.KrylovSetPCType(P.ksp, pctype)
            wg_d, dwg_d = api.DeviceArray.from_numpy(wg), api.DeviceArray.from_numpy(dwg)
            F_d, x_d = api.DeviceArray(6 * N), api.DeviceArray(6 * N)
            P.assemble_system(wg_d, dwg_d, F_d, want_J=False)
            P.assemble_system(wg_d, dwg_d, None, want_J=True)
            it, r0, hist, conv = P.solve(x_d, F_d)
            api.sync()
            assert conv, (name, it, hist[-1] / r0)
            x = x_d.numpy()[:4 * N]
            assert np.abs(x - x_ref).max() <= 1e-6 * np.abs(x_ref).max(), name
            its[name] = it
        finally:
            P.close()
    assert its["dilu"] < 0.6 * its["jacobi"], its
