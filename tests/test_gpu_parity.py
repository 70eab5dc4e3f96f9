"""GPU parity tests: the HIP path (through the C ABI / C host API in
dedflow_amd/libdedflow.so) against the CPU oracle on the same seeded inputs, and
against the committed golden fixtures.

Bars (BASELINE.json north_star): indexing / coloring / batching / patterns bit-exact;
assembled values, F and Krylov residuals within 1e-10 relative (fp64).  "Relative"
is taken against the largest magnitude of the array being compared (entries that
cancel to ~0 cannot be compared entry-relative).
The oracle itself is "parity unpinned" w.r.t. the real reference (no runnable
reference, no fixtures -- SURVEY.md F7).
"""
import ctypes as C
import os

import numpy as np
import pytest

from dedflow_amd.meshgen import kuhn_cube, single_tet, synthetic_fields

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
RTOL = 1e-10


def close(a, b, rtol=RTOL):
    scale = max(np.abs(b).max(), 1e-300)
    err = np.abs(a - b).max() / scale
    return err <= rtol, err


@pytest.fixture(scope="module")
def api():
    from dedflow_amd import api as A
    A.lib()  # raises if the HIP library is missing: no fallback
    return A


@pytest.fixture(scope="module", params=[4, 12])
def case(request, api, oracle_lib):
    m = kuhn_cube(request.param, jitter=0.2)
    S = oracle_lib.System(m)
    P = api.Problem(m)
    wg, dwg = synthetic_fields(m)
    yield m, S, P, wg, dwg
    P.close()


def test_library_exports_and_loads(api):
    assert api.lib().dfl_abi_version() == 1


def test_xorwow_priorities_bit_exact(api, oracle_lib):
    n = 3 * 4096 + 123
    d = api.DeviceArray(n, np.int32)
    api.lib().GenerateRandomColor(d.ptr, n, 256)
    api.sync()
    assert np.array_equal(d.numpy(), oracle_lib.priorities(n))


def test_pattern_bit_exact(case):
    _, S, P, _, _ = case
    rp, ci = P.pattern()
    assert np.array_equal(rp, S.rp11) and np.array_equal(ci, S.ci11)
    for attr, (orp, oci) in ((P.spy3x3, (S.rp33, S.ci33)), (P.spy3x1, (S.rp31, S.ci31)), (P.spy1x3, (S.rp13, S.ci13))):
        rp, ci = P.pattern(attr)
        assert np.array_equal(rp, orp) and np.array_equal(ci, oci)  # incl. the Q3-fixed last entry


def test_coloring_and_batches_bit_exact(case):
    m, S, P, _, _ = case
    assert S.num_ties == 0  # tie-free => index tie-break == reference '<' (Q1)
    assert P.num_color == S.num_color
    assert np.array_equal(P.color(), S.color)
    assert np.array_equal(P.batch_offset(), S.batch_offset)
    assert np.array_equal(P.batch_ind(), S.batch_ind)


def test_assemble_tet_matches_oracle(case, api):
    m, S, P, wg, dwg = case
    dwg_d, wg_d = api.DeviceArray.from_numpy(dwg), api.DeviceArray.from_numpy(wg)
    F_d = api.DeviceArray(6 * S.N)
    api.lib().MatrixZero(P.J)
    P.assemble_tet(wg_d, dwg_d, F_d, want_J=True)
    api.sync()
    F = np.zeros(6 * S.N)
    vals = S.new_values()
    S.assemble_tet(wg, dwg, F, vals)
    ok, err = close(F_d.numpy(), F)
    assert ok, f"F rel err {err:.3e}"
    for name, g, o in zip(("A00", "A01", "A10", "A11"), P.export_values(), vals):
        ok, err = close(g, o)
        assert ok, f"{name} rel err {err:.3e}"


def test_assemble_face_matches_oracle(case, api):
    m, S, P, wg, dwg = case
    dwg_d, wg_d = api.DeviceArray.from_numpy(dwg), api.DeviceArray.from_numpy(wg)
    F_d = api.DeviceArray(6 * S.N)
    api.lib().MatrixZero(P.J)
    P.assemble_face(wg_d, dwg_d, F_d, want_J=True)
    api.sync()
    F = np.zeros(6 * S.N)
    vals = S.new_values()
    S.assemble_face(wg, dwg, F, vals)
    assert np.abs(F).max() > 0
    ok, err = close(F_d.numpy(), F)
    assert ok, f"F rel err {err:.3e}"
    for name, g, o in zip(("A00", "A01", "A10", "A11"), P.export_values(), vals):
        ok, err = close(g, o)
        assert ok, f"{name} rel err {err:.3e}"


def _assembled(case, api):
    m, S, P, wg, dwg = case
    dwg_d, wg_d = api.DeviceArray.from_numpy(dwg), api.DeviceArray.from_numpy(wg)
    F_d = api.DeviceArray(6 * S.N)
    P.assemble_system(wg_d, dwg_d, F_d, want_J=False)
    P.assemble_system(wg_d, dwg_d, None, want_J=True)
    api.sync()
    F, vals = S.assemble_system(wg, dwg, True, True)
    return F_d, F, vals


def test_assemble_system_with_dirichlet(case, api):
    m, S, P, wg, dwg = case
    F_d, F, vals = _assembled(case, api)
    Fg = F_d.numpy()
    ok, err = close(Fg, F)
    assert ok, f"F rel err {err:.3e}"
    assert np.all(Fg[4 * S.N:] == 0.0)
    gv = P.export_values()
    for name, g, o in zip(("A00", "A01", "A10", "A11"), gv, vals):
        ok, err = close(g, o)
        assert ok, f"{name} rel err {err:.3e}"
    # Dirichlet rows are exactly unit rows / zero RHS (bit-exact, no tolerance)
    A = S.to_scipy(gv).tocsr()
    for group, bctype in api.REFERENCE_BCS:
        for ic, t in enumerate(bctype):
            if t != 1:
                continue
            for node in S.bnodes(group)[::7]:
                r = node * 3 + ic
                row = A.getrow(r)
                assert row[0, r] == 1.0 and row.nnz - (row.data == 0).sum() == 1
                assert Fg[r] == 0.0


def test_spmv_matches_oracle(case, api):
    m, S, P, wg, dwg = case
    _, _, vals = _assembled(case, api)
    x = np.random.default_rng(0).normal(size=6 * S.N)
    x_d = api.DeviceArray.from_numpy(x)
    y_d = api.DeviceArray.from_numpy(np.full(6 * S.N, 7.0))
    P.matvec(x_d, y_d)
    api.sync()
    y = S.matvec(vals, x)
    yg = y_d.numpy()
    ok, err = close(yg[:4 * S.N], y[:4 * S.N])
    assert ok, f"SpMV rel err {err:.3e}"
    assert np.all(yg[4 * S.N:] == 7.0)  # [4N,6N) untouched (Q5)
    # alpha/beta form: y = -A x + y  (krylov.c:116)
    y0 = np.random.default_rng(1).normal(size=6 * S.N)
    y_d.upload(y0)
    api.lib().MatrixAMVPBY(P.J, -1.0, x_d.ptr, 1.0, y_d.ptr)
    api.sync()
    ref = y0.copy()
    S.amvpby(vals, -1.0, x, 1.0, ref)
    ok, err = close(y_d.numpy()[:4 * S.N], ref[:4 * S.N])
    assert ok, f"AMVPBY rel err {err:.3e}"


def test_jacobi_pc_matches_oracle(case, api):
    m, S, P, wg, dwg = case
    _, _, vals = _assembled(case, api)
    N = S.N
    d33, d1 = api.DeviceArray(9 * N), api.DeviceArray(N)
    a = P.spy1x1.contents
    api.lib().dfl_pc_jacobi_setup(N, a.row_ptr, a.col_ind, P.block_values().ptr, d33.ptr, d1.ptr, None)
    r = np.random.default_rng(2).normal(size=6 * N)
    r_d, z_d = api.DeviceArray.from_numpy(r), api.DeviceArray(6 * N)
    api.lib().dfl_pc_jacobi_apply(N, 6 * N, d33.ptr, d1.ptr, r_d.ptr, z_d.ptr, None)
    api.sync()
    o33, o1 = S.pc_setup(vals)
    ok, err = close(d33.numpy(), o33, 1e-9)  # closed-form vs pivoted-LU inverse
    assert ok, f"dinv33 rel err {err:.3e}"
    z = S.pc_apply(o33, o1, r)
    ok, err = close(z_d.numpy(), z, 1e-9)
    assert ok, f"PC apply rel err {err:.3e}"


def test_cgs_kernels(api):
    L = api.lib()
    rng = np.random.default_rng(3)
    n, k = 10007 * 2, 37
    Q = rng.normal(size=(k, n))
    w = rng.normal(size=n)
    Qd, wd = api.DeviceArray.from_numpy(Q.reshape(-1)), api.DeviceArray.from_numpy(w)
    h, nrm = api.DeviceArray(k), api.DeviceArray(1)
    work = api.DeviceArray(int(L.dfl_cgs_work_size(n, k)) + 2048)
    L.dfl_cgs_dots(n, k, Qd.ptr, n, wd.ptr, h.ptr, work.ptr, None)
    L.dfl_cgs_update(n, k, Qd.ptr, n, h.ptr, wd.ptr, nrm.ptr, 1, work.ptr, None)
    api.sync()
    href = Q @ w
    wref = w - Q.T @ href
    assert np.allclose(h.numpy(), href, rtol=1e-12, atol=1e-12 * np.abs(href).max())
    assert np.allclose(wd.numpy(), wref, rtol=0, atol=1e-12 * np.abs(wref).max())
    assert np.isclose(nrm.numpy()[0], np.linalg.norm(wref), rtol=1e-13)


def test_gmres_matches_oracle(case, api):
    m, S, P, wg, dwg = case
    F_d, F, vals = _assembled(case, api)
    x_d = api.DeviceArray(6 * S.N)
    it, r0, hist, conv = P.solve(x_d, F_d)
    xo, ho, r0o, ito = S.gmres(vals, F)
    assert it == ito
    assert abs(r0 - r0o) <= 1e-12 * r0o
    # residual history: tolerance grows with the iteration count (CGS reduction order, SURVEY hard part vii)
    k = np.arange(1, it + 1)
    assert np.all(np.abs(hist - ho) <= 1e-10 * r0o * np.maximum(1.0, k / 10.0)), np.abs(hist - ho).max() / r0o
    xg = x_d.numpy()
    ok, err = close(xg[:4 * S.N], xo[:4 * S.N], 1e-8)
    assert ok, f"solution rel err {err:.3e}"
    assert np.all(xg[4 * S.N:] == 0.0)


def test_gmres_nonzero_initial_guess_matches_oracle(case, api):
    """x0 != 0 takes the r0 = b - A x0 path (x0 = 0 skips that matvec: r0 = b exactly); both against the oracle."""
    m, S, P, wg, dwg = case
    F_d, F, vals = _assembled(case, api)
    rng = np.random.default_rng(17)
    x0 = np.zeros(6 * S.N)
    x0[:4 * S.N] = 1e-3 * rng.normal(size=4 * S.N)
    x_d = api.DeviceArray.from_numpy(x0)
    it, r0, hist, conv = P.solve(x_d, F_d)
    xo, ho, r0o, ito = S.gmres(vals, F, x0=x0)
    assert it == ito and abs(r0 - r0o) <= 1e-12 * r0o
    k = np.arange(1, it + 1)
    assert np.all(np.abs(hist - ho) <= 1e-10 * r0o * np.maximum(1.0, k / 10.0)), np.abs(hist - ho).max() / r0o
    ok, err = close(x_d.numpy()[:4 * S.N], xo[:4 * S.N], 1e-8)
    assert ok, f"solution rel err {err:.3e}"


@pytest.mark.parametrize("name", ["cube_M4", "cube_M12"])
def test_golden_fixtures(name, api):
    """HIP path against the committed fixtures only (no oracle call)."""
    g = np.load(os.path.join(GOLD, name + ".npz"))
    m = kuhn_cube(int(g["M"]), jitter=float(g["jitter"]))
    P = api.Problem(m)
    try:
        assert np.array_equal(P.color(), g["color"])
        assert np.array_equal(P.batch_offset(), g["batch_offset"])
        assert np.array_equal(P.batch_ind(), g["batch_ind"])
        rp, ci = P.pattern()
        assert np.array_equal(rp, g["row_ptr"]) and np.array_equal(ci, g["col_ind"])
        wg, dwg = synthetic_fields(m)
        wg_d, dwg_d = api.DeviceArray.from_numpy(wg), api.DeviceArray.from_numpy(dwg)
        F_d = api.DeviceArray(6 * P.N)
        P.assemble_system(wg_d, dwg_d, F_d, want_J=False)
        P.assemble_system(wg_d, dwg_d, None, want_J=True)
        api.sync()
        ok, err = close(F_d.numpy(), g["F"])
        assert ok, err
        vals = P.export_values()
        for k, v in zip(("A00", "A01", "A10", "A11"), vals):
            if name == "cube_M4":
                ok, err = close(v, g[k])
            else:
                ok, err = close(v[::37], g[k + "_sample"], 1e-10 * np.abs(v).max() / max(np.abs(g[k + "_sample"]).max(), 1e-300))
                assert np.isclose(np.dot(v, v), g[k + "_sum"][2], rtol=1e-9)
            assert ok, (k, err)
        x_d = api.DeviceArray(6 * P.N)
        it, r0, hist, _ = P.solve(x_d, F_d)
        assert it == int(g["gmres_it"])
        assert np.all(np.abs(hist - g["gmres_hist"]) <= 1e-9 * float(g["gmres_r0"]))
    finally:
        P.close()


def test_reference_color_schedule_matches_compact_schedule(api, oracle_lib):
    """Schedule 0 launches the reference's JPL color batches one to one (reference summation
    order); schedule 1 (default) executes a compact re-coloring.  Same coloring artefacts, same
    values up to summation order."""
    m = kuhn_cube(8, jitter=0.2)
    S = oracle_lib.System(m)
    wg, dwg = synthetic_fields(m)
    F, vals = S.assemble_system(wg, dwg, True, True)
    out = []
    for mode in (0, 1, 2, 3, 4):
        P = api.Problem(m, schedule=mode)
        try:
            assert np.array_equal(P.color(), S.color) and np.array_equal(P.batch_ind(), S.batch_ind)
            wg_d, dwg_d = api.DeviceArray.from_numpy(wg), api.DeviceArray.from_numpy(dwg)
            F_d = api.DeviceArray(6 * S.N)
            P.assemble_system(wg_d, dwg_d, F_d, want_J=False)
            P.assemble_system(wg_d, dwg_d, None, want_J=True)
            api.sync()
            out.append((F_d.numpy(), P.export_values()))
        finally:
            P.close()
    for Fg, vg in out:
        ok, err = close(Fg, F)
        assert ok, err
        for g, o in zip(vg, vals):
            ok, err = close(g, o)
            assert ok, err
    # the two schedules agree to rounding, far inside the parity bar
    assert np.abs(out[0][0] - out[1][0]).max() <= 1e-13 * np.abs(F).max()


def test_row_owner_schedule_add_and_overwrite(api):
    """Schedule 3: AssembleSystemTet keeps the reference's additive contract (assemble twice = 2x), while
    AssembleSystem (zero + assemble, src/main.c:44-52) overwrites whatever J held before; both across patch sizes."""
    m = kuhn_cube(7, jitter=0.2)
    wg, dwg = synthetic_fields(m)
    L = api.lib()
    ref = None
    for leaf, cap in ((16, 255), (3, 40), (64, 1023)):
        L.DflSetRowPatchParameters(leaf, cap)
        P = api.Problem(m, schedule=3)
        try:
            wg_d, dwg_d = api.DeviceArray.from_numpy(wg), api.DeviceArray.from_numpy(dwg)
            L.MatrixZero(P.J)
            P.assemble_tet(wg_d, dwg_d, None, want_J=True)
            api.sync()
            v1 = P.block_values().numpy().copy()
            P.assemble_tet(wg_d, dwg_d, None, want_J=True)
            api.sync()
            v2 = P.block_values().numpy().copy()
            assert np.abs(v2 - 2.0 * v1).max() <= 1e-13 * np.abs(v1).max()
            P.assemble_system(wg_d, dwg_d, None, want_J=True)   # J holds 2x: must be overwritten, then faces + BC rows
            api.sync()
            v3 = P.block_values().numpy().copy()
            L.MatrixZero(P.J)
            P.assemble_system(wg_d, dwg_d, None, want_J=True)
            api.sync()
            v4 = P.block_values().numpy()
            assert np.abs(v3 - v4).max() <= 1e-13 * np.abs(v4).max()
            if ref is None:
                ref = v1
            assert np.abs(v1 - ref).max() <= 1e-13 * np.abs(ref).max()
        finally:
            P.close()
    L.DflSetRowPatchParameters(16, 255)


def test_slot_owner_schedule_add_overwrite_and_bitwise_reproducible(api, oracle_lib):
    """Schedule 4 (default): every nodal nonzero is summed by its owner lanes in a fixed order.  AssembleSystemTet keeps
    the reference's additive contract, AssembleSystem overwrites, two runs agree BITWISE (like the reference's colored
    scatter), and the values match the oracle for every patch size (incl. 1-node patches and tet-capped patches)."""
    m = kuhn_cube(7, jitter=0.2)
    wg, dwg = synthetic_fields(m)
    S = oracle_lib.System(m)
    _, vals = S.assemble_system(wg, dwg, False, True)
    L = api.lib()
    try:
        for leaf, cap, tcap in ((16, 255, 208), (1, 40, 64), (5, 60, 40), (64, 1023, 500), (12, 192, 150)):
            L.DflSetSlotPatchParameters(leaf, cap, tcap)
            P = api.Problem(m, schedule=4)
            try:
                wg_d, dwg_d = api.DeviceArray.from_numpy(wg), api.DeviceArray.from_numpy(dwg)
                L.MatrixZero(P.J)
                P.assemble_tet(wg_d, dwg_d, None, want_J=True)
                api.sync()
                v1 = P.block_values().numpy().copy()
                P.assemble_tet(wg_d, dwg_d, None, want_J=True)
                api.sync()
                v2 = P.block_values().numpy().copy()
                assert np.abs(v2 - 2.0 * v1).max() <= 1e-13 * np.abs(v1).max()
                P.assemble_system(wg_d, dwg_d, None, want_J=True)   # J holds 2x: must be overwritten, then faces + BC rows
                api.sync()
                v3 = P.block_values().numpy().copy()
                L.MatrixZero(P.J)
                P.assemble_system(wg_d, dwg_d, None, want_J=True)
                api.sync()
                v4 = P.block_values().numpy().copy()
                assert np.array_equal(v3, v4), "slot-owner assembly is not bitwise reproducible"
                for g, o in zip(P.export_values(), vals):
                    ok, err = close(g, o)
                    assert ok, (leaf, cap, tcap, err)
            finally:
                P.close()
    finally:
        L.DflSetSlotPatchParameters(16, 255, 208)


def test_patch_residual_is_reproducible_and_patch_size_independent(api, oracle_lib):
    """Schedules 2/3 assemble F by spatial tet patches with a fixed summation order: bitwise equal run to run,
    equal to the oracle within the parity bar for every patch size (incl. 1-tet patches and ragged tails)."""
    m = kuhn_cube(7, jitter=0.2)
    S = oracle_lib.System(m)
    wg, dwg = synthetic_fields(m)
    F, _ = S.assemble_system(wg, dwg, True, False)
    L = api.lib()
    try:
        for leaf, cap in ((64, 64), (1, 4), (7, 20), (64, 96)):
            L.DflSetRhsPatchParameters(leaf, cap)
            P = api.Problem(m, schedule=3)
            try:
                wg_d, dwg_d = api.DeviceArray.from_numpy(wg), api.DeviceArray.from_numpy(dwg)
                runs = []
                for rep in range(2):
                    F_d = api.DeviceArray(6 * S.N)
                    P.assemble_system(wg_d, dwg_d, F_d, want_J=False)
                    api.sync()
                    runs.append(F_d.numpy().copy())
                assert np.array_equal(runs[0], runs[1])
                ok, err = close(runs[0], F)
                assert ok, (leaf, cap, err)
            finally:
                P.close()
    finally:
        L.DflSetRhsPatchParameters(64, 64)


def test_two_problems_with_different_configurations_coexist(api, oracle_lib):
    """The assembly configuration (schedule, weak-BC face group, patch shapes) and the Newton work vectors are per mesh:
    two problems alive in one process, assembled alternately, each match the oracle for ITS configuration."""
    ma, mb = kuhn_cube(5, jitter=0.2), kuhn_cube(4, jitter=0.1)
    Sa, Sb = oracle_lib.System(ma), oracle_lib.System(mb)
    wa, dwa = synthetic_fields(ma)
    wb, dwb = synthetic_fields(mb)
    L = api.lib()
    Pa = api.Problem(ma, schedule=1)                # compact colors, reference face group 4
    L.DflSetWeakBCGroup(1)
    Pb = api.Problem(mb, schedule=4)                # slot-owner patches, weak-BC faces on group 1
    L.DflSetWeakBCGroup(4)
    try:
        da = [api.DeviceArray.from_numpy(v) for v in (wa, dwa)]
        db = [api.DeviceArray.from_numpy(v) for v in (wb, dwb)]
        Fa, Fb = api.DeviceArray(6 * Pa.N), api.DeviceArray(6 * Pb.N)
        for rep in range(2):                         # alternate: a's call must not disturb b's cached schedules and vice versa
            Pa.assemble_system(da[0], da[1], Fa, want_J=True)
            Pb.assemble_system(db[0], db[1], Fb, want_J=True)
        api.sync()
        Fo, vo = Sa.assemble_system(wa, dwa, True, True)
        ok, err = close(Fa.numpy(), Fo)
        assert ok, err
        for g, o in zip(Pa.export_values(), vo):
            ok, err = close(g, o)
            assert ok, err
        Fo = np.zeros(6 * Sb.N)
        vo = Sb.new_values()
        Sb.assemble_tet(wb, dwb, Fo, vo)
        Sb.assemble_face(wb, dwb, Fo, vo, group=1)
        Fo[4 * Sb.N:] = 0.0
        for group, bctype in api.REFERENCE_BCS:
            bt, bn = np.asarray(bctype, np.int32), Sb.bnodes(group)
            oracle_lib.lib().orc_dirichlet_vec(oracle_lib._p(Fo), C.c_int(bn.size), oracle_lib._p(bn), C.c_int(3), oracle_lib._p(bt))
            oracle_lib.lib().orc_dirichlet_mat(C.c_int(bn.size), oracle_lib._p(bn), C.c_int(3), oracle_lib._p(bt), C.c_int(Sb.N),
                                               oracle_lib._p(Sb.rp33), oracle_lib._p(Sb.ci33), oracle_lib._p(vo[0]),
                                               oracle_lib._p(Sb.rp31), oracle_lib._p(Sb.ci31), oracle_lib._p(vo[1]))
        ok, err = close(Fb.numpy(), Fo)
        assert ok, err
        for g, o in zip(Pb.export_values(), vo):
            ok, err = close(g, o)
            assert ok, err
    finally:
        Pa.close()
        Pb.close()


def test_wave_residual_is_reproducible_and_shape_independent(api, oracle_lib):
    """Schedule 4 assembles F with one wave per spatial patch (padded layout, no workgroup barriers): bitwise equal run to
    run, equal to the oracle within the parity bar for every supported patch shape and kernel build -- the default
    lane-per-tet kernel on 64-tet patches (2 waves per SIMD), its 1-wave build (dfl_tune_asm 64), the 4-lanes-per-tet wave
    kernel on the same patches (32) and on the smaller shapes.  dfl_tune(2, 8) caps the persistent kernel at 8 workgroups so
    that every wave walks several patches through the pipelined loop (prefetch of the next patch's records and lists)."""
    m = kuhn_cube(12, jitter=0.2)   # 10368 tets = 162 full patches of 64 on 32 persistent waves
    S = oracle_lib.System(m)
    wg, dwg = synthetic_fields(m)
    F, _ = S.assemble_system(wg, dwg, True, False)
    L = api.lib()
    L.dfl_tune.argtypes = [C.c_int, C.c_int]
    try:
        for tets, nodes, bits, cap in ((64, 64, 0, 8), (64, 64, 0, 0), (64, 64, 64, 8), (64, 64, 32, 0), (32, 48, 0, 0), (16, 32, 0, 0)):
            L.DflSetRhsWaveParameters(tets, nodes)
            L.dfl_tune_asm(bits)
            L.dfl_tune(2, cap)
            P = api.Problem(m, schedule=4)
            try:
                wg_d, dwg_d = api.DeviceArray.from_numpy(wg), api.DeviceArray.from_numpy(dwg)
                runs = []
                for rep in range(2):
                    F_d = api.DeviceArray(6 * S.N)
                    P.assemble_system(wg_d, dwg_d, F_d, want_J=False)
                    api.sync()
                    runs.append(F_d.numpy().copy())
                assert np.array_equal(runs[0], runs[1])
                ok, err = close(runs[0], F)
                assert ok, (tets, nodes, bits, cap, err)
            finally:
                P.close()
    finally:
        L.DflSetRhsWaveParameters(64, 64)
        L.dfl_tune_asm(0)
        L.dfl_tune(2, 0)


def test_geometry_cache_follows_moved_nodes(api, oracle_lib):
    """The element-geometry cache is per mesh; after the node coordinates change, DflMeshGeometryChanged makes the next
    assembly agree with the oracle on the moved mesh (and without it the stale cache is what one gets)."""
    import copy
    m = kuhn_cube(5, jitter=0.2)
    wg, dwg = synthetic_fields(m)
    P = api.Problem(m)
    L = api.lib()
    try:
        wg_d, dwg_d = api.DeviceArray.from_numpy(wg), api.DeviceArray.from_numpy(dwg)
        F_d = api.DeviceArray(6 * P.N)
        P.assemble_system(wg_d, dwg_d, F_d, want_J=True)
        m2 = copy.deepcopy(m)
        m2.xg = np.ascontiguousarray((m.xg.reshape(-1, 3) * np.array([1.0, 1.1, 0.95])).reshape(-1))
        S2 = oracle_lib.System(m2)
        F2, vals2 = S2.assemble_system(wg, dwg, True, True)
        xg_dev = api.DeviceArray(3 * P.N, np.float64, ptr=P.mesh.contents.device.contents.xg, owner=False)
        xg_dev.upload(m2.xg)
        L.DflMeshGeometryChanged(P.mesh)
        F_d.zero()
        P.assemble_system(wg_d, dwg_d, F_d, want_J=False)
        P.assemble_system(wg_d, dwg_d, None, want_J=True)
        api.sync()
        ok, err = close(F_d.numpy(), F2)
        assert ok, err
        for g, o in zip(P.export_values(), vals2):
            ok, err = close(g, o)
            assert ok, err
    finally:
        P.close()


def test_restarted_gmres_matches_the_oracle_cycles(api, oracle_lib):
    """KrylovSetRestart(m): GMRES(m) = full-GMRES cycles of the restatement of krylov.c:56-334, each from the current
    iterate (fixed work: atol = rtol = 0).  m >= max_iter is the reference's full GMRES: bitwise the same history as the
    default path."""
    m = kuhn_cube(6, jitter=0.2)
    wg, dwg = synthetic_fields(m)
    S = oracle_lib.System(m)
    F, vals = S.assemble_system(wg, dwg, True, True)
    L = api.lib()
    P = api.Problem(m, maxit=45, atol=0.0, rtol=0.0)
    try:
        wg_d, dwg_d = api.DeviceArray.from_numpy(wg), api.DeviceArray.from_numpy(dwg)
        F_d, x_d = api.DeviceArray(6 * P.N), api.DeviceArray(6 * P.N)
        P.assemble_system(wg_d, dwg_d, F_d, want_J=False)
        P.assemble_system(wg_d, dwg_d, None, want_J=True)
        it0, r00, h0, _ = P.solve(x_d, F_d)
        x0 = x_d.numpy().copy()
        L.KrylovSetRestart(P.ksp, 200)      # >= max_iter: same code path, same bits
        x_d.zero()
        it1, r01, h1, _ = P.solve(x_d, F_d)
        assert it1 == it0 == 45 and r01 == r00 and np.array_equal(h1, h0) and np.array_equal(x_d.numpy(), x0)
        L.KrylovSetRestart(P.ksp, 15)
        x_d.zero()
        it2, r02, h2, _ = P.solve(x_d, F_d)
        xo, ho, r0o, ito = S.gmres_restarted(vals, F, 15, 45)
        assert it2 == ito == 45 and abs(r02 - r0o) <= 1e-12 * r0o
        k = np.arange(1, 46)
        assert np.all(np.abs(h2 - ho) <= 1e-10 * r0o * np.maximum(1.0, k / 10.0))
        assert np.array_equal(h2[:15], h0[:15])            # the first cycle IS the full-GMRES start
        ok, err = close(x_d.numpy(), xo, 1e-8)
        assert ok, err
        # the restarted recurrence restarts from the TRUE residual: ||b - A x|| after the solve equals the last entry
        y_d = api.DeviceArray(6 * P.N)
        P.matvec(x_d, y_d)
        true = np.linalg.norm(F[:4 * P.N] - y_d.numpy()[:4 * P.N])
        assert abs(true - h2[-1]) <= 1e-9 * r0o
    finally:
        P.close()


def test_krylov_zero_rhs_returns_immediately(api):
    """b = 0, x0 = 0: the initial residual is exactly zero; the solve returns converged with 0 iterations and x stays 0
    (the reference would divide by the zero norm at krylov.c:130 -- deliberate guard)."""
    m = kuhn_cube(4, jitter=0.2)
    wg, dwg = synthetic_fields(m)
    P = api.Problem(m, maxit=30)
    try:
        wg_d, dwg_d = api.DeviceArray.from_numpy(wg), api.DeviceArray.from_numpy(dwg)
        P.assemble_system(wg_d, dwg_d, None, want_J=True)
        x_d, b_d = api.DeviceArray(6 * P.N), api.DeviceArray(6 * P.N)
        it, r0, hist, conv = P.solve(x_d, b_d)
        api.sync()
        assert it == 0 and r0 == 0.0 and conv and not x_d.numpy().any()
    finally:
        P.close()


def test_reference_utility_launchers(api):
    """Small reference launchers kept for launcher-level completeness (indexing.h:9-13, csr_impl.h:6-9, matrix_impl.h:17-26)."""
    import ctypes as C
    L = api.lib()
    i32, vp, f64 = C.c_int32, C.c_void_p, C.c_double
    rng = np.random.default_rng(5)
    data = rng.integers(0, 7, 5000).astype(np.int32)
    d = api.DeviceArray.from_numpy(data)
    for fn in (L.CountValueI, L.CountValueColorLegacy):
        fn.restype, fn.argtypes = i32, [vp, i32, i32]
        assert fn(d.ptr, data.size, 3) == int((data == 3).sum())
    L.CountValueColor.restype, L.CountValueColor.argtypes = i32, [vp, i32, i32, vp]
    assert L.CountValueColor(d.ptr, data.size, 6, None) == int((data == 6).sum())
    out = api.DeviceArray(int((data == 3).sum()), np.int32)
    L.FindValueI.argtypes = [vp, i32, i32, vp]
    L.FindValueI(d.ptr, data.size, 3, out.ptr)
    api.sync()
    assert np.array_equal(out.numpy(), np.nonzero(data == 3)[0])
    # vertex -> element maps for 6- and 8-vertex elements (same kernels as the tet map)
    for nshl, name in ((6, "Prism"), (8, "Hex")):
        E, Nn = 300, 120
        conn = np.stack([rng.permutation(Nn)[:nshl] for _ in range(E)]).astype(np.int32)
        c_d = api.DeviceArray.from_numpy(conn.reshape(-1))
        rp_d = api.DeviceArray(Nn + 1, np.int32)
        getattr(L, "GenerateV2EMapRow%sGPU" % name).argtypes = [vp, i32, i32, vp]
        getattr(L, "GenerateV2EMapRow%sGPU" % name)(c_d.ptr, E, Nn, rp_d.ptr)
        rpv = rp_d.numpy()
        assert np.array_equal(np.diff(rpv), np.bincount(conn.reshape(-1), minlength=Nn)) and rpv[-1] == E * nshl
        col_d = api.DeviceArray(E * nshl, np.int32)
        getattr(L, "GenerateV2EMapCol%sGPU" % name).argtypes = [vp, i32, i32, vp, vp]
        getattr(L, "GenerateV2EMapCol%sGPU" % name)(c_d.ptr, E, Nn, rp_d.ptr, col_d.ptr)
        colv = col_d.numpy()
        for n in range(0, Nn, 7):
            assert sorted(colv[rpv[n]:rpv[n + 1]]) == sorted(np.nonzero((conn == n).any(axis=1))[0])
    # pattern lookups and scalar value setters on a small CSR pattern
    m = kuhn_cube(3, jitter=0.0)
    P = api.Problem(m, color=False)
    try:
        rp, ci = P.pattern()
        nnz = ci.size
        k = rng.integers(0, nnz, 400)
        rows = (np.searchsorted(rp, k, side="right") - 1).astype(np.int32)
        cols = ci[k].astype(np.int32)
        cols[::50] = (cols[::50] + 1000) % P.N                      # some pairs that are (almost surely) not in the pattern
        present = np.array([c in ci[rp[r]:rp[r + 1]] for r, c in zip(rows, cols)])
        want = np.array([rp[r] + np.searchsorted(ci[rp[r]:rp[r + 1]], c) if ok else -1 for r, c, ok in zip(rows, cols, present)], np.int32)
        r_d, c_d, ind_d = api.DeviceArray.from_numpy(rows), api.DeviceArray.from_numpy(cols), api.DeviceArray(rows.size, np.int32)
        L.CSRAttrGetNZIndBatchedGPU.argtypes = [vp, i32, vp, vp, vp]
        L.CSRAttrGetNZIndBatchedGPU(C.cast(P.spy1x1, vp), rows.size, r_d.ptr, c_d.ptr, ind_d.ptr)
        api.sync()
        assert np.array_equal(ind_d.numpy(), want)
        vals0 = rng.normal(size=nnz)
        upd = rng.normal(size=rows.size)
        uniq = np.unique(want[want >= 0], return_index=True)[1]     # one update per nonzero: order-independent check
        sel = np.nonzero(want >= 0)[0][uniq]
        mv = api.DeviceArray.from_numpy(vals0)
        rp_d, ci_d = api.DeviceArray.from_numpy(rp.astype(np.int32)), api.DeviceArray.from_numpy(ci.astype(np.int32))
        rs, cs, us = (api.DeviceArray.from_numpy(np.ascontiguousarray(a[sel])) for a in (rows, cols, upd))
        L.MatrixCSRSetValuesCOOGPU.argtypes = [vp, f64, i32, i32, vp, vp, i32, vp, vp, vp, f64]
        L.MatrixCSRSetValuesCOOGPU(mv.ptr, 0.5, P.N, P.N, rp_d.ptr, ci_d.ptr, sel.size, rs.ptr, cs.ptr, us.ptr, 2.0)
        api.sync()
        exp = vals0.copy()
        exp[want[sel]] = 0.5 * exp[want[sel]] + 2.0 * upd[sel]
        assert np.array_equal(mv.numpy(), exp)
        # point Jacobi on the scalar pattern (pc_impl.h:6-7)
        dvals = rng.uniform(1.0, 2.0, nnz)
        dv, xv, yv = api.DeviceArray.from_numpy(dvals), api.DeviceArray.from_numpy(rng.normal(size=P.N)), api.DeviceArray(P.N)
        diag = np.array([dvals[rp[i] + np.searchsorted(ci[rp[i]:rp[i + 1]], i)] for i in range(P.N)])
        L.PCJacobiDevice.argtypes = [i32, i32, vp, vp, vp, vp, vp]
        L.PCJacobiDevice(P.N, nnz, dv.ptr, rp_d.ptr, ci_d.ptr, xv.ptr, yv.ptr)
        api.sync()
        x_host = xv.numpy()
        assert np.array_equal(yv.numpy(), x_host / diag)
        L.PCJacobiInplaceDevice.argtypes = [i32, i32, vp, vp, vp, vp]
        L.PCJacobiInplaceDevice(P.N, nnz, dv.ptr, rp_d.ptr, ci_d.ptr, xv.ptr)
        api.sync()
        assert np.array_equal(xv.numpy(), x_host / diag)
        idx = api.DeviceArray.from_numpy(want[sel])
        L.MatrixCSRSetValuesIndGPU.argtypes = [vp, f64, i32, vp, vp, f64]
        L.MatrixCSRSetValuesIndGPU(mv.ptr, 0.0, sel.size, idx.ptr, us.ptr, 1.0)
        api.sync()
        exp[want[sel]] = upd[sel]
        assert np.array_equal(mv.numpy(), exp)
    finally:
        P.close()


def test_reference_block_scatter_launcher(api):
    """SetBlockValueToSubmatGPU (matrix_impl.h:59-64), the reference's colored scatter, on one color batch against a numpy
    restatement of the row-expanded layout (csr_impl.cu:24-59): scalar row node*br+ii, entry (k-start)*bc+jj."""
    import ctypes as C
    m = kuhn_cube(4, jitter=0.1)
    P = api.Problem(m)
    L = api.lib()
    try:
        rp, ci = P.pattern()
        N, nnz = P.N, ci.size
        color, off, ind = P.color(), P.batch_offset(), P.batch_ind()
        batch = ind[off[2]:off[3]].astype(np.int32)                 # one color: conflict-free
        rng = np.random.default_rng(11)
        nshl, bs = 4, 6
        val = rng.normal(size=(batch.size, nshl * nshl, bs * bs))   # stride 36, lda 6
        offset = np.array([0, 3, 4, 5, 6], np.int32)
        shapes = {(0, 0): (3, 3), (0, 1): (3, 1), (1, 0): (1, 3), (1, 1): (1, 1)}
        arrays = {k: rng.normal(size=nnz * br * bc) for k, (br, bc) in shapes.items()}
        dev = {k: api.DeviceArray.from_numpy(v) for k, v in arrays.items()}
        ptrs = np.zeros(16, np.uint64)
        for (i, j), d in dev.items():
            ptrs[i * 4 + j] = d.ptr
        ptr_d = api.DeviceArray.from_numpy(ptrs)
        off_d, b_d, val_d = api.DeviceArray.from_numpy(offset), api.DeviceArray.from_numpy(batch), api.DeviceArray.from_numpy(val.reshape(-1))
        ien_d = api.DeviceArray.from_numpy(m.ien)
        rp_d, ci_d = api.DeviceArray.from_numpy(rp.astype(np.int32)), api.DeviceArray.from_numpy(ci.astype(np.int32))
        vp, i32, f64 = C.c_void_p, C.c_int32, C.c_double
        L.SetBlockValueToSubmatGPU.argtypes = [vp, f64, i32, vp, i32, i32, vp, vp, i32, i32, vp, vp, vp, C.c_int, C.c_int, f64, vp]
        alpha, beta = 0.75, 1.5
        L.SetBlockValueToSubmatGPU(ptr_d.ptr, alpha, 4, off_d.ptr, nshl, batch.size, b_d.ptr, ien_d.ptr, N, N, rp_d.ptr, ci_d.ptr,
                                   val_d.ptr, bs, bs * bs, beta, None)
        api.sync()
        exp = {k: v.copy() for k, v in arrays.items()}
        ien = m.ien.reshape(-1, 4)
        for s_, e in enumerate(batch):
            for a in range(4):
                for b in range(4):
                    row, col = ien[e, a], ien[e, b]
                    start, ln = rp[row], rp[row + 1] - rp[row]
                    k = start + np.searchsorted(ci[start:start + ln], col)
                    blk = val[s_, a * 4 + b].reshape(bs, bs)
                    for (i, j), (br, bc) in shapes.items():
                        base = start * br * bc + (k - start) * bc
                        for ii in range(br):
                            for jj in range(bc):
                                p_ = base + ii * ln * bc + jj
                                exp[(i, j)][p_] = alpha * exp[(i, j)][p_] + beta * blk[offset[i] + ii, offset[j] + jj]
        for k_, d in dev.items():
            assert np.allclose(d.numpy(), exp[k_], rtol=0, atol=1e-14), k_
    finally:
        P.close()


@pytest.mark.parametrize("group", [1, 4])
def test_weak_bc_faces_on_a_chosen_group(api, oracle_lib, group):
    """AssembleSystemTetFace on boundary group 4 (the reference's hard-coded group) and on another group through
    DflSetWeakBCGroup: residual and Jacobian face terms against the oracle's face assembly of the same group."""
    m = kuhn_cube(5, jitter=0.2)
    S = oracle_lib.System(m)
    wg, dwg = synthetic_fields(m)
    L = api.lib()
    L.DflSetWeakBCGroup.argtypes = [C.c_int32]
    L.DflSetWeakBCGroup(group)
    P = api.Problem(m, bcs=[])
    try:
        wg_d, dwg_d = api.DeviceArray.from_numpy(wg), api.DeviceArray.from_numpy(dwg)
        F_d = api.DeviceArray(6 * S.N)
        L.MatrixZero(P.J)
        P.assemble_face(wg_d, dwg_d, F_d, want_J=True)
        api.sync()
        F = np.zeros(6 * S.N)
        vals = S.new_values()
        S.assemble_face(wg, dwg, F, vals, group=group)
        assert np.abs(F).max() > 0
        ok, err = close(F_d.numpy(), F)
        assert ok, err
        for g, o in zip(P.export_values(), vals):
            ok, err = close(g, o)
            assert ok, err
    finally:
        P.close()
        L.DflSetWeakBCGroup(4)


def test_single_tet_all_faces(api, oracle_lib):
    """DBG_TET-like case (src/main.c:357-361): one element, face assembly on a chosen group."""
    m = single_tet()
    S = oracle_lib.System(m)
    P = api.Problem(m, bcs=[])
    try:
        rng = np.random.default_rng(5)
        wg, dwg = rng.normal(size=6 * S.N), rng.normal(size=6 * S.N)
        wg_d, dwg_d = api.DeviceArray.from_numpy(wg), api.DeviceArray.from_numpy(dwg)
        F_d = api.DeviceArray(6 * S.N)
        api.lib().MatrixZero(P.J)
        P.assemble_tet(wg_d, dwg_d, F_d, want_J=True)
        api.sync()
        F = np.zeros(6 * S.N)
        vals = S.new_values()
        S.assemble_tet(wg, dwg, F, vals)
        ok, err = close(F_d.numpy(), F)
        assert ok, err
        for g, o in zip(P.export_values(), vals):
            ok, err = close(g, o)
            assert ok, err
    finally:
        P.close()
