"""Round-3 GPU tests: exports that had no caller in the suite, the soft limits of the schedule builders
(item 8), the bounded default of the Krylov work-space placement (item 7), assembly into a reference-layout matrix (weak item
12) and BASELINE config 4 at its stated size (item 5)."""
import ctypes as C
import os

import numpy as np
import pytest

from dedflow_amd.meshgen import dem_particles, fan_mesh, kuhn_cube, synthetic_fields

pytestmark = pytest.mark.gpu
vp, i32, f64 = C.c_void_p, C.c_int32, C.c_double


@pytest.fixture(scope="module")
def api():
    from dedflow_amd import api as A
    A.lib()
    return A


def close(g, o, tol=1e-10):
    err = np.abs(g - o).max() / max(np.abs(o).max(), 1e-300)
    return err <= tol, err


# ---- exports of round 2 that only the symbol check covered -----------------------------------------------------------------
def test_particle_context_and_array_h5_round_trip(api, tmp_path):
    """ArraySave / ArrayLoad and ParticleContextSave / ParticleContextLoad (Array.c:242-261, Particle.c:66-103) through
    libdedflow_h5.so: <group>/{coord,vel,acc}, host arrays on disk, device arrays refreshed by the load."""
    from dedflow_amd import h5
    L, H = api.lib(), h5.lib()
    L.Init(0, None)
    for fn, res, args in (("ArrayCreateHost", C.POINTER(api.Array), [i32]), ("ArrayDestroy", None, [C.POINTER(api.Array)]),
                          ("ParticleContextCopy", None, [C.POINTER(api.ParticleContext)] * 2)):
        getattr(L, fn).restype, getattr(L, fn).argtypes = res, args
    H.ArraySave.argtypes = [C.POINTER(api.Array), vp, C.c_char_p]
    H.ArrayLoad.argtypes = [C.POINTER(api.Array), vp, C.c_char_p]
    H.ParticleContextSave.argtypes = [C.POINTER(api.ParticleContext), vp, C.c_char_p]
    H.ParticleContextLoad.argtypes = [C.POINTER(api.ParticleContext), vp, C.c_char_p]
    H.H5FileIsReadable.restype, H.H5FileIsReadable.argtypes = i32, [vp]
    H.H5FileIsWritable.restype, H.H5FileIsWritable.argtypes = i32, [vp]
    p = str(tmp_path / "particles.h5").encode()
    n = 1000
    a = L.ArrayCreateHost(n)
    src = np.random.default_rng(1).normal(size=n)
    C.memmove(a.contents.data, src.ctypes.data, 8 * n)
    x, v, R = dem_particles(500, 0.01)
    pc = api.Particles(x, v, R)
    pc.compute_forces()
    api.sync()
    L.ParticleContextUpdateHost(pc.ctx)      # device -> host arrays (coord, vel, acc)
    acc = pc.arrays()[2]
    f = H.H5OpenFile(p, b"w")
    assert H.H5FileIsWritable(f)          # ("w" creates the file read-write: it is readable too, as in h5util.c:43-57)
    H.ArraySave(a, f, b"aux/vec")
    H.ParticleContextSave(pc.ctx, f, b"ptc/group1")
    H.H5CloseFile(f)
    assert np.array_equal(h5.read_dataset(p.decode(), "aux/vec", np.float64), src)
    assert np.array_equal(h5.read_dataset(p.decode(), "ptc/group1/coord", np.float64), x)
    assert np.array_equal(h5.read_dataset(p.decode(), "ptc/group1/acc", np.float64), acc)
    b = L.ArrayCreateHost(n)
    pc2 = api.Particles(np.zeros_like(x), np.zeros_like(v), R)
    f = H.H5OpenFile(p, b"r")
    assert H.H5FileIsReadable(f) and not H.H5FileIsWritable(f)
    H.ArrayLoad(b, f, b"aux/vec")
    H.ParticleContextLoad(pc2.ctx, f, b"ptc/group1")
    H.H5CloseFile(f)
    got = np.ctypeslib.as_array(C.cast(b.contents.data, C.POINTER(f64)), shape=(n,))
    assert np.array_equal(got, src)
    x2, v2, a2 = pc2.arrays()                # the DEVICE arrays were refreshed by the load
    assert np.array_equal(x2, x) and np.array_equal(v2, v) and np.array_equal(a2, acc)
    L.ArrayDestroy(a); L.ArrayDestroy(b)
    pc.close(); pc2.close()


def test_matrix_value_setters_and_element_scatter_vs_numpy(api):
    """Object-level MatrixSetValuesCOO / MatrixSetValuesInd (val = alpha * val + beta * new) and MatrixAddElemValueBatched
    (one scalar per (element, a, b), color by color) on a scalar CSR matrix over the nodal pattern, against numpy; entries
    outside the pattern are skipped."""
    import scipy.sparse as sp
    L = api.lib()
    m = kuhn_cube(4, jitter=0.2)
    P = api.Problem(m)
    try:
        L.MatrixSetValuesCOO.argtypes = [C.POINTER(api.Matrix), f64, i32, vp, vp, vp, f64]
        L.MatrixSetValuesInd.argtypes = [C.POINTER(api.Matrix), f64, i32, vp, vp, f64]
        L.MatrixAddElemValueBatched.argtypes = [C.POINTER(api.Matrix), i32, i32, vp, vp, vp, vp]
        rp, ci = P.pattern()
        nnz, N = ci.size, P.N
        A = L.MatrixCreateTypeCSR(P.spy1x1, None)
        L.MatrixSetup(A)
        csr = C.cast(A.contents.data, C.POINTER(api.MatrixCSR)).contents
        rng = np.random.default_rng(2)
        base = rng.normal(size=nnz)
        keep = []                             # device operands must outlive the (asynchronous) calls that read them

        def dev(a):
            keep.append(api.DeviceArray.from_numpy(a))
            return keep[-1].ptr
        L.MatrixSetValuesInd(A, 0.0, nnz, None, dev(base), 1.0)        # ind = NULL: identity
        api.sync()
        assert np.array_equal(api.d2h(csr.val, nnz, np.float64), base)
        pick = rng.choice(nnz, 300, replace=False).astype(np.int32)
        new = rng.normal(size=300)
        L.MatrixSetValuesInd(A, 0.5, 300, dev(pick), dev(new), 2.0)
        ref = base.copy()
        ref[pick] = 0.5 * ref[pick] + 2.0 * new
        api.sync()
        assert np.allclose(api.d2h(csr.val, nnz, np.float64), ref, rtol=0, atol=1e-15)
        # COO: distinct (row, col) pairs inside the pattern, plus pairs outside it (skipped)
        rows = np.repeat(np.arange(N), np.diff(rp)).astype(np.int32)
        k = rng.choice(nnz, 200, replace=False)
        r_in, c_in = rows[k], ci[k]
        r_out = np.array([0, 1, 2], np.int32)
        c_out = np.array([N - 1, N - 1, N - 2], np.int32)          # far corner: not neighbours of nodes 0..2
        rr, cc = np.concatenate([r_in, r_out]).astype(np.int32), np.concatenate([c_in, c_out]).astype(np.int32)
        vv = rng.normal(size=rr.size)
        L.MatrixSetValuesCOO(A, 1.0, rr.size, dev(rr), dev(cc), dev(vv), -1.0)
        ref[k] = ref[k] - vv[:200]
        api.sync()
        assert np.allclose(api.d2h(csr.val, nnz, np.float64), ref, rtol=0, atol=1e-15)
        # element scatter, one conflict-free batch (color) at a time == dense accumulation
        L.MatrixZero(A)
        ien = m.ien.reshape(-1, 4)
        ev = rng.normal(size=(P.T, 4, 4))
        boff = P.batch_offset()
        bind = P.batch_ind()
        for c in range(P.num_color):
            lo, hi = int(boff[c]), int(boff[c + 1])
            # values are indexed by BATCH SLOT: hand every color its own slice of per-element values
            L.MatrixAddElemValueBatched(A, 4, hi - lo, P.mesh.contents.batch_ind + 4 * lo, P.mesh.contents.device.contents.ien,
                                        dev(ev[bind[lo:hi]].reshape(-1)), None)
        api.sync()
        dense = sp.coo_matrix((ev.reshape(-1), (np.repeat(ien, 4, axis=1).reshape(-1), np.tile(ien, (1, 4)).reshape(-1))),
                              shape=(N, N)).tocsr()
        dense.sort_indices()
        assert np.array_equal(dense.indices, ci) and np.array_equal(dense.indptr, rp)
        ok, err = close(api.d2h(csr.val, nnz, np.float64), dense.data, 1e-13)
        assert ok, err
        del keep
        L.MatrixDestroy(A)
    finally:
        P.close()


def test_csrattr_queries_against_the_oracle_pattern(api, oracle_lib):
    """CSRAttrLength / CSRAttrRow / CSRAttrGetNonzeroIndBatched (csr.h:32-36) on the nodal and an expanded pattern."""
    L = api.lib()
    m = kuhn_cube(5, jitter=0.2)
    S = oracle_lib.System(m)
    P = api.Problem(m)
    try:
        L.CSRAttrLength.restype, L.CSRAttrLength.argtypes = i32, [C.POINTER(api.CSRAttr), i32]
        L.CSRAttrRow.restype, L.CSRAttrRow.argtypes = vp, [C.POINTER(api.CSRAttr), i32]
        L.CSRAttrGetNonzeroIndBatched.argtypes = [C.POINTER(api.CSRAttr), i32, vp, vp, vp]
        for attr, rp, ci in ((P.spy1x1, S.rp11, S.ci11), (P.spy3x3, S.rp33, S.ci33)):
            for row in (0, 7, attr.contents.num_row - 1):
                n = L.CSRAttrLength(attr, row)
                assert n == rp[row + 1] - rp[row]
                assert np.array_equal(api.d2h(L.CSRAttrRow(attr, row), n, np.int32), ci[rp[row]:rp[row + 1]])
            rng = np.random.default_rng(4)
            k = rng.choice(ci.size, 500, replace=False)
            rows = (np.searchsorted(rp, k, side="right") - 1).astype(np.int32)
            cols = ci[k].astype(np.int32)
            rows = np.concatenate([rows, [0]]).astype(np.int32)                      # one pair outside the pattern -> -1
            cols = np.concatenate([cols, [attr.contents.num_col - 1]]).astype(np.int32)
            ind = api.DeviceArray(rows.size, np.int32)
            rows_d, cols_d = api.DeviceArray.from_numpy(rows), api.DeviceArray.from_numpy(cols)
            L.CSRAttrGetNonzeroIndBatched(attr, rows.size, rows_d.ptr, cols_d.ptr, ind.ptr)
            api.sync()
            got = ind.numpy()
            assert np.array_equal(got[:-1], k) and got[-1] == -1
    finally:
        P.close()


def test_matrix_object_life_cycles_leave_the_pool_flat(api):
    """MatrixCSRCreate / MatrixCSRDestroy and MatrixFSCreate / MatrixFSDestroy (matrix.h:141-147) directly, and twenty
    create -> setup -> fill -> destroy cycles of the driver's FS matrix: the device pool returns to where it was."""
    L = api.lib()
    m = kuhn_cube(6, jitter=0.2)
    P = api.Problem(m)
    try:
        L.MatrixCSRCreate.restype, L.MatrixCSRCreate.argtypes = C.POINTER(api.MatrixCSR), [C.POINTER(api.CSRAttr), vp]
        L.MatrixFSCreate.restype, L.MatrixFSCreate.argtypes = C.POINTER(api.MatrixFS), [i32, vp, vp]
        L.MatrixCSRDestroy.argtypes = [C.POINTER(api.Matrix)]
        L.MatrixFSDestroy.argtypes = [C.POINTER(api.Matrix)]

        def pool():
            r, u = C.c_int64(0), C.c_int64(0)
            L.DflDevicePoolStats(C.byref(r), C.byref(u))
            return r.value, u.value
        api.sync()
        before = pool()
        c = L.MatrixCSRCreate(P.spy3x3, None)
        assert c.contents.attr.contents.nnz == 9 * P.nnz1 and not c.contents.val and not c.contents.external_attr
        C.CDLL(None).free.argtypes = [vp]
        C.CDLL(None).free(C.cast(c, vp))                       # a bare MatrixCSR is host memory only (values come on first use)
        offset = (i32 * 5)(0, 3, 4, 5, 6)
        fs = L.MatrixFSCreate(4, offset, None)
        assert fs.contents.n_offset == 4 and not fs.contents.block_val
        libc = C.CDLL(None)
        libc.calloc.restype, libc.calloc.argtypes = vp, [C.c_size_t, C.c_size_t]
        shell = C.cast(libc.calloc(1, C.sizeof(api.Matrix)), C.POINTER(api.Matrix))   # MatrixFSDestroy frees the shell too
        shell.contents.type, shell.contents.data = 4, C.cast(fs, vp)
        L.MatrixFSDestroy(shell)
        for _ in range(20):
            J = L.MatrixCreateTypeFS(4, offset, None)
            f2 = C.cast(J.contents.data, C.POINTER(api.MatrixFS)).contents
            f2.spy1x1 = P.spy1x1
            f2.mat[0] = L.MatrixCreateTypeCSR(P.spy3x3, None)
            f2.mat[1] = L.MatrixCreateTypeCSR(P.spy3x1, None)
            f2.mat[4] = L.MatrixCreateTypeCSR(P.spy1x3, None)
            f2.mat[5] = L.MatrixCreateTypeCSR(P.spy1x1, None)
            L.MatrixSetup(J)
            L.MatrixZero(J)
            L.MatrixFSExportSubmatrices(J)                     # materialises the four reference-layout arrays
            api.sync()
            L.MatrixDestroy(J)
        A = L.MatrixCreateTypeCSR(P.spy1x1, None)
        L.MatrixSetup(A)
        L.MatrixZero(A)
        L.MatrixCSRDestroy(A)
        api.sync()
        assert pool() == before
    finally:
        P.close()


def test_mesh3d_data_create_h5(api, tmp_path):
    """Mesh3DDataCreateH5 (MeshData.c:57-109): host-side coordinates + connectivity of a group."""
    from dedflow_amd import h5
    H = h5.lib()
    m = kuhn_cube(3, jitter=0.1)
    p = str(tmp_path / "box.h5")
    h5.write_mesh(p, m)
    H.Mesh3DDataCreateH5.restype, H.Mesh3DDataCreateH5.argtypes = C.POINTER(api.Mesh3DData), [vp, C.c_char_p]
    L = api.lib()
    L.Mesh3DDataDestroy.argtypes = [C.POINTER(api.Mesh3DData)]
    f = H.H5OpenFile(p.encode(), b"r")
    d = H.Mesh3DDataCreateH5(f, b"mesh")
    H.H5CloseFile(f)
    dd = d.contents
    assert dd.is_host and (dd.num_node, dd.num_tet, dd.num_prism, dd.num_hex) == (m.num_node, m.num_tet, 0, 0)
    assert np.array_equal(np.ctypeslib.as_array(C.cast(dd.xg, C.POINTER(f64)), shape=(3 * m.num_node,)), m.xg)
    assert np.array_equal(np.ctypeslib.as_array(C.cast(dd.ien, C.POINTER(i32)), shape=(4 * m.num_tet,)), m.ien)
    L.Mesh3DDataDestroy(d)


# ---- schedule builders fail soft --------------------------------------------------------------------------------------------
def test_slot_schedule_refusal_falls_back_to_the_colored_schedule(api, oracle_lib, capfd):
    """A mesh the slot-owner kernel cannot hold (forced here by lowering its tet limit below the valence of a cube node) is
    refused with a printed reason and assembled by schedule 1 instead -- no trap, the overwrite contract of AssembleSystem
    kept (the fall-back schedule adds, so the library zeroes first), values equal to the oracle."""
    L = api.lib()
    L.DflSlotPatchSetTestLimits.argtypes = [C.c_int, C.c_int]
    m = kuhn_cube(5, jitter=0.2)
    wg, dwg = synthetic_fields(m)
    S = oracle_lib.System(m)
    F, vals = S.assemble_system(wg, dwg, True, True)
    L.DflSlotPatchSetTestLimits(0, 10)
    try:
        P = api.Problem(m, schedule=4)
        try:
            wg_d, dwg_d = api.DeviceArray.from_numpy(wg), api.DeviceArray.from_numpy(dwg)
            F_d = api.DeviceArray(6 * P.N)
            garbage = api.DeviceArray.from_numpy(np.full(P.nnz1 * 16, 7.0))
            api.hip().hipMemcpy(L.MatrixFSBlockValues(P.J), garbage.ptr, garbage.nbytes, 3)   # J must be OVERWRITTEN
            P.assemble_system(wg_d, dwg_d, F_d, want_J=True)
            api.sync()
            err = capfd.readouterr().err
            assert "tets touch one node patch" in err and "falling back to schedule 1" in err, err
            for g, o in zip(P.export_values(), vals):
                ok, e = close(g, o)
                assert ok, e
            ok, e = close(F_d.numpy(), F)
            assert ok, e
            P.assemble_system(wg_d, dwg_d, None, want_J=True)      # second call: already on schedule 1, same values
            api.sync()
            for g, o in zip(P.export_values(), vals):
                assert close(g, o)[0]
        finally:
            P.close()
    finally:
        L.DflSlotPatchSetTestLimits(0, 0)


@pytest.mark.parametrize("schedule", [4, 1])
def test_vertex_shared_by_76_tets(api, oracle_lib, schedule, capfd):
    """A fan of 76 tets around one vertex (inside the reference's limits: 41 nonzeros in its row, 76 colors) needs more than
    the 64 conflict-free classes of the compact execution schedule: the colored kernels run the reference's JPL batches
    instead (a message, no trap); the slot-owner schedule holds the mesh as it is.  F and J against the oracle."""
    m = fan_mesh(40)
    assert np.bincount(m.ien).max() == 76
    wg, dwg = synthetic_fields(m)
    bcs = [(0, (1, 1, 1))]
    S = oracle_lib.System(m)
    assert S.num_color == 76
    F, vals = S.assemble_system(wg, dwg, True, True, bcs=bcs)
    P = api.Problem(m, schedule=schedule, bcs=bcs)
    try:
        assert "more than 64 conflict-free classes" in capfd.readouterr().err
        assert np.array_equal(P.color(), S.color)
        wg_d, dwg_d = api.DeviceArray.from_numpy(wg), api.DeviceArray.from_numpy(dwg)
        F_d = api.DeviceArray(6 * P.N)
        P.assemble_system(wg_d, dwg_d, F_d, want_J=True)
        api.sync()
        for g, o in zip(P.export_values(), vals):
            ok, e = close(g, o)
            assert ok, e
        ok, e = close(F_d.numpy(), F)
        assert ok, e
    finally:
        P.close()


# ---- reference-layout matrix through the library's own assembly -----------------------------------------------------------
def test_assemble_system_into_a_reference_layout_matrix(api, oracle_lib):
    """AssembleSystem (tet + weak-BC faces + Dirichlet rows) with J kept in the reference's four row-expanded arrays
    (MatrixFSUseReferenceLayout): the kernels work on a scratch block array and the result lands in the sub-matrices."""
    m = kuhn_cube(5, jitter=0.2)
    wg, dwg = synthetic_fields(m)
    S = oracle_lib.System(m)
    F, vals = S.assemble_system(wg, dwg, True, True)
    L = api.lib()
    P = api.Problem(m, reference_layout=True)
    try:
        assert not L.MatrixFSBlockValues(P.J)
        wg_d, dwg_d = api.DeviceArray.from_numpy(wg), api.DeviceArray.from_numpy(dwg)
        F_d = api.DeviceArray(6 * P.N)
        P.assemble_system(wg_d, dwg_d, F_d, want_J=True)
        api.sync()
        for g, o in zip(P.export_values(), vals):
            ok, e = close(g, o)
            assert ok, e
        # additive contract of AssembleSystemTet on this layout: a second tet pass doubles the tet part
        vt = S.new_values()
        S.assemble_tet(wg, dwg, None, vt)
        before = P.export_values()
        P.assemble_tet(wg_d, dwg_d, None, want_J=True)
        api.sync()
        for g, b, t in zip(P.export_values(), before, vt):
            ok, e = close(g, b + t)
            assert ok, e
    finally:
        P.close()


# ---- Krylov work-space placement: bounded default, explicit heavy form ---------------------------------------------------
def test_default_placement_is_bounded_and_moves_nothing(api):
    """First solve of a system large enough for the calibration (4N >= 2^20): at most four candidates, no value-array copy,
    MatrixFSBlockValues unchanged, under a second; after the solver is destroyed free VRAM is back where it was."""
    import time
    L = api.lib()
    free_b, total_b = C.c_size_t(0), C.c_size_t(0)
    H = api.hip()
    H.hipMemGetInfo.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]

    def free_vram():
        api.sync()
        assert H.hipMemGetInfo(C.byref(free_b), C.byref(total_b)) == 0
        return free_b.value
    m = kuhn_cube(64, jitter=0.2)          # 274625 nodes: 4N = 1.1M
    wg, dwg = synthetic_fields(m)
    P = api.Problem(m, maxit=20, atol=0.0, rtol=0.0)
    try:
        wg_d, dwg_d = api.DeviceArray.from_numpy(wg), api.DeviceArray.from_numpy(dwg)
        F_d, x_d = api.DeviceArray(6 * P.N), api.DeviceArray(6 * P.N)
        P.assemble_system(wg_d, dwg_d, F_d, want_J=False)
        P.assemble_system(wg_d, dwg_d, None, want_J=True)
        val_before = L.MatrixFSBlockValues(P.J)
        L.DflWaitDeviceMemoryQuiet(10.0)
        f0 = free_vram()
        t0 = time.perf_counter()
        it, r0, hist, _ = P.solve(x_d, F_d)
        api.sync()
        dt = time.perf_counter() - t0
        log = (L.DflKrylovCalibrationLog() or b"").decode()
        assert "default, bounded" in log and "value array untouched" in log, log
        ncand = int(log.split("into ")[1].split(" candidates")[0])
        assert 2 <= ncand <= 4
        assert L.MatrixFSBlockValues(P.J) == val_before
        assert dt < 2.0, dt
        f1 = free_vram()
        ws = f0 - f1                        # what the solver keeps: basis + vectors (the losing candidates are gone)
        basis = 8 * 6 * P.N * 21
        assert 0 <= ws <= basis + (256 << 20), (ws, basis)
        it2, r02, hist2, _ = P.solve(x_d.__class__(6 * P.N), F_d)    # no second calibration, same history
        assert np.array_equal(hist2, hist)
        L.KrylovDestroy(P.ksp)
        P.ksp = L.KrylovCreateGMRES(20, 0.0, 0.0, None)
        assert abs(free_vram() - f0) <= (64 << 20)
    finally:
        P.close()


def test_explicit_calibration_respects_its_byte_cap(api):
    """DflKrylovCalibratePlacement(ksp, A, cap): never more than `cap` bytes of transient device memory (sampled by a
    watcher thread through hipMemGetInfo), the solve afterwards does not calibrate again and gives the same history as a
    solver that never calibrated."""
    import threading
    L = api.lib()
    L.DflKrylovCalibratePlacement.argtypes = [vp, C.POINTER(api.Matrix), C.c_int64]
    H = api.hip()
    H.hipMemGetInfo.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    m = kuhn_cube(64, jitter=0.2)
    wg, dwg = synthetic_fields(m)
    os.environ["DFL_WS_SETTLE_S"] = "1"
    P = api.Problem(m, maxit=20, atol=0.0, rtol=0.0)
    try:
        wg_d, dwg_d = api.DeviceArray.from_numpy(wg), api.DeviceArray.from_numpy(dwg)
        F_d, x_d = api.DeviceArray(6 * P.N), api.DeviceArray(6 * P.N)
        P.assemble_system(wg_d, dwg_d, F_d, want_J=False)
        P.assemble_system(wg_d, dwg_d, None, want_J=True)
        api.sync()
        a, b = C.c_size_t(0), C.c_size_t(0)
        H.hipMemGetInfo(C.byref(a), C.byref(b))
        f0, low, stop = a.value, [a.value], [False]

        def watch():
            fa, fb = C.c_size_t(0), C.c_size_t(0)
            while not stop[0]:
                H.hipMemGetInfo(C.byref(fa), C.byref(fb))
                low[0] = min(low[0], fa.value)
        th = threading.Thread(target=watch)
        th.start()
        cap = 3 << 30
        L.DflKrylovCalibratePlacement(P.ksp, P.J, cap)
        api.sync()
        stop[0] = True
        th.join()
        basis = 8 * 6 * P.N * 21
        assert f0 - low[0] <= cap + basis + (512 << 20), (f0 - low[0], cap, basis)
        log = (L.DflKrylovCalibrationLog() or b"").decode()
        assert "explicit calibration" in log, log
        it, r0, hist, _ = P.solve(x_d, F_d)
        assert "explicit calibration" in (L.DflKrylovCalibrationLog() or b"").decode()     # the solve did not calibrate again
        os.environ["DFL_WS_CANDIDATES"] = "1"
        try:
            L.KrylovDestroy(P.ksp)
            P.ksp = L.KrylovCreateGMRES(20, 0.0, 0.0, None)
            L.KrylovSetVerbose(P.ksp, 0)
            x2 = api.DeviceArray(6 * P.N)
            it2, r02, hist2, _ = P.solve(x2, F_d)
        finally:
            del os.environ["DFL_WS_CANDIDATES"]
        assert it2 == it and np.array_equal(hist2, hist)
    finally:
        os.environ.pop("DFL_WS_SETTLE_S", None)
        P.close()


# ---- BASELINE config 4 at its stated size ---------------------------------------------------------------------------------
def test_config4_coupled_step_at_size(api):
    """1M-tet fluid mesh (M=55) + 100k DEM particles, DflTimeStep with 10 contact sub-steps, two steps: Newton residuals drop,
    every linear solve converged, the particles end bitwise where ten stand-alone ParticleContextUpdate calls per step take
    them (fluid and particles do not interact -- nor in the reference, main.c:547-569), the device pool stays flat."""
    L = api.lib()
    m = kuhn_cube(55, jitter=0.2)
    assert m.num_tet == 998250
    wg, dwg = synthetic_fields(m)
    N = m.num_node
    wg[3 * N:4 * N] = 0.0
    x, v, R = dem_particles(100000, 0.004)
    P = api.Problem(m, maxit=120, atol=1e-12, rtol=1e-4)
    # (the reference's Jacobi tree does not reach rtol 1e-4 within GMRES(120) at 1M tets -- the reference would simply go on
    # with an unconverged increment; "every solve converged" needs the build's two-level preconditioner)
    L.KrylovSetPCType(P.ksp, api.PC_TWOLEVEL)
    pc = api.Particles(x, v, R, dt=1e-4)
    alone = api.Particles(x, v, R, dt=1e-4)
    try:
        st = [api.DeviceArray.from_numpy(a) for a in (wg, 0.1 * dwg, 0.1 * dwg)]
        F_d, dx_d = api.DeviceArray(6 * N), api.DeviceArray(6 * N)

        def pool():
            r, u = C.c_int64(0), C.c_int64(0)
            L.DflDevicePoolStats(C.byref(r), C.byref(u))
            return r.value, u.value
        it, rn, ri = P.time_step(st[0], st[1], st[2], F_d, dx_d, newton_maxit=2, particles=pc, dem_substeps=10)
        api.sync()
        p1 = pool()
        stats = L.KrylovGetStats(P.ksp).contents
        assert stats.total_solves == it and stats.total_converged == it, (it, stats.total_solves, stats.total_converged)
        assert np.all(np.isfinite(rn)) and rn[0] < 0.2 * ri[0] and rn[1] < ri[1], (rn, ri)
        it2, rn2, ri2 = P.time_step(st[0], st[1], st[2], F_d, dx_d, newton_maxit=2, particles=pc, dem_substeps=10)
        api.sync()
        stats = L.KrylovGetStats(P.ksp).contents
        assert stats.total_solves == it + it2 == stats.total_converged
        assert np.all(np.isfinite(rn2)) and rn2[0] < ri2[0]
        assert pool() == p1
        for _ in range(20):
            alone.update()
        api.sync()
        for a, b in zip(pc.arrays(), alone.arrays()):
            assert np.array_equal(a, b)
        assert np.abs(pc.arrays()[0] - x).max() > 0.0
    finally:
        alone.close()
        pc.close()
        P.close()


def test_pipelined_gmres_single_gpu_matches_the_oracle_history(api, oracle_lib):
    """KrylovSetPipelined on one GPU (no communicator: same arithmetic, nothing to overlap): 40 steps of the Jacobi-tree GMRES
    follow the oracle's residual history to 1e-6 r0 (the z-recurrence and the Pythagorean norm cost accuracy: the
    unpipelined solver holds 1e-10), the solution agrees to 1e-6, and switching the option off restores the exact path."""
    m = kuhn_cube(8, jitter=0.2)
    wg, dwg = synthetic_fields(m)
    S = oracle_lib.System(m)
    F, vals = S.assemble_system(wg, dwg, True, True)
    xo, ho, r0o, ito = S.gmres(vals, F, maxit=40, atol=0.0, rtol=0.0)
    L = api.lib()
    P = api.Problem(m, maxit=40, atol=0.0, rtol=0.0)
    try:
        wg_d, dwg_d = api.DeviceArray.from_numpy(wg), api.DeviceArray.from_numpy(dwg)
        F_d, x_d = api.DeviceArray(6 * P.N), api.DeviceArray(6 * P.N)
        P.assemble_system(wg_d, dwg_d, F_d, want_J=False)
        P.assemble_system(wg_d, dwg_d, None, want_J=True)
        it0, r00, h0, _ = P.solve(x_d, F_d)
        assert np.abs(h0 - ho).max() <= 1e-9 * r0o
        L.KrylovSetPipelined(P.ksp, 1)
        x_d.zero()
        it, r0, hist, _ = P.solve(x_d, F_d)
        api.sync()
        assert it == 40 and abs(r0 - r0o) <= 1e-12 * r0o
        dev = np.abs(hist - ho).max() / r0o
        assert dev <= 1e-6, dev
        assert np.abs(x_d.numpy() - xo).max() <= 1e-6 * np.abs(xo).max()
        assert not L.KrylovGetStats(P.ksp).contents.fused_norm_cancelled
        L.KrylovSetPipelined(P.ksp, 0)
        x_d.zero()
        it2, r02, h2, _ = P.solve(x_d, F_d)
        assert np.array_equal(h2, h0)
    finally:
        P.close()


# ---- Array BLAS-1 wrappers and Field (VERDICT r2 missing item 5) ------------------------------------------------------------
def test_array_blas_wrappers_and_field_vs_numpy(api, tmp_path):
    """Array.h:24-36 (Array.c:83-238) on host AND device arrays against numpy: Set / Zero / SetAt (repeated index keeps the last
    value) / GetAt / Scale / Dot / Norm2 / AXPY / AXPBY; Field.h:22-33 (Field.c:15-77): create on a mesh, Init through a
    callback, Copy, UpdateHost / UpdateDevice, Save / Load through libdedflow_h5.so.  Tolerance 1e-14 relative for the two
    reductions on the device (a fixed two-stage tree, numpy pairwise), 1e-12 on the host (index order, as the reference); AXPY /
    AXPBY to one rounding; everything else bit-exact."""
    from dedflow_amd import h5
    L, H = api.lib(), h5.lib()
    L.Init(0, None)
    AP = C.POINTER(api.Array)

    class Field(C.Structure):
        _fields_ = [("shape", i32 * 2), ("host", AP), ("device", AP)]
    FP = C.POINTER(Field)
    for fn, res, args in (("ArrayCreateHost", AP, [i32]), ("ArrayCreateDevice", AP, [i32]), ("ArrayDestroy", None, [AP]),
                          ("ArrayCopy", None, [AP, AP, C.c_int]), ("ArraySet", None, [AP, f64]), ("ArrayZero", None, [AP]),
                          ("ArraySetAt", None, [AP, i32, vp, vp]), ("ArrayGetAt", None, [AP, i32, vp, vp]),
                          ("ArrayAt", None, [AP, i32, vp, vp]), ("ArrayScale", None, [AP, f64]),
                          ("ArrayDot", None, [C.POINTER(f64), AP, AP]), ("ArrayNorm2", None, [C.POINTER(f64), AP]),
                          ("ArrayAXPY", None, [AP, f64, AP]), ("ArrayAXPBY", None, [AP, f64, AP, f64]),
                          ("FieldCreate3D", FP, [C.POINTER(api.Mesh3D), i32]), ("FieldDestroy", None, [FP]),
                          ("FieldInit", None, [FP, vp, vp]), ("FieldCopy", None, [FP, FP]), ("FieldUpdateHost", None, [FP]),
                          ("FieldUpdateDevice", None, [FP])):
        getattr(L, fn).restype, getattr(L, fn).argtypes = res, args
    H.FieldSave.argtypes = [FP, vp, C.c_char_p]
    H.FieldLoad.argtypes = [FP, vp, C.c_char_p]
    H2D, D2H = 1, 2
    n = 100003
    rng = np.random.default_rng(5)
    x0, y0 = rng.normal(size=n), rng.normal(size=n)

    def host_view(a):
        return np.ctypeslib.as_array(C.cast(a.contents.data, C.POINTER(f64)), shape=(a.contents.len,))

    def make(side, src):
        h = L.ArrayCreateHost(src.size)
        host_view(h)[:] = src
        if side == "host":
            return h, None
        d = L.ArrayCreateDevice(src.size)
        L.ArrayCopy(d, h, H2D)
        return d, h

    def read(a, stage):
        if stage is None:
            return host_view(a).copy()
        L.ArrayCopy(stage, a, D2H)
        return host_view(stage).copy()

    for side in ("host", "device"):
        x, xs = make(side, x0)
        y, ys = make(side, y0)
        r = f64()
        tol = 1e-12 if side == "host" else 1e-14     # (the host side sums in index order like the reference: ~sqrt(n) eps)
        L.ArrayDot(C.byref(r), x, y)
        assert abs(r.value - np.dot(x0, y0)) <= tol * np.abs(x0 * y0).sum(), side
        L.ArrayNorm2(C.byref(r), x)
        assert abs(r.value - np.linalg.norm(x0)) <= tol * np.linalg.norm(x0), side
        L.ArrayAXPY(y, 0.75, x)
        ref = y0 + 0.75 * x0
        got = read(y, ys)
        assert np.abs(got - ref).max() <= 2.3e-16 * np.abs(ref).max() + 1e-300, side   # (an fma may round once, numpy twice)
        L.ArrayAXPBY(y, -1.5, x, 0.25)
        ref2 = -1.5 * x0 + 0.25 * got
        got2 = read(y, ys)
        assert np.abs(got2 - ref2).max() <= 4.5e-16 * max(np.abs(ref2).max(), 1.0), side
        L.ArrayScale(x, 3.0)
        assert np.array_equal(read(x, xs), 3.0 * x0), side
        idx = np.array([5, 17, n - 1, 17, 0], np.int32)          # 17 twice: the later value stays
        val = np.array([1.0, 2.0, 3.0, 4.0, 5.0])
        L.ArraySetAt(x, idx.size, idx.ctypes.data, val.ctypes.data)
        ref3 = 3.0 * x0
        for i, v in zip(idx, val):
            ref3[i] = v
        assert np.array_equal(read(x, xs), ref3), side
        out = np.zeros(idx.size)
        L.ArrayGetAt(x, idx.size, idx.ctypes.data, out.ctypes.data)
        assert np.array_equal(out, ref3[idx]), side
        out2 = np.zeros(idx.size)
        L.ArrayAt(x, idx.size, idx.ctypes.data, out2.ctypes.data)
        assert np.array_equal(out2, out), side
        L.ArraySet(x, -2.5)
        assert np.array_equal(read(x, xs), np.full(n, -2.5)), side
        L.ArrayZero(x)
        assert not read(x, xs).any(), side
        for a in (x, xs, y, ys):
            if a is not None:
                L.ArrayDestroy(a)

    # Field on a mesh: 6 nodal values per node
    mesh = kuhn_cube(4, jitter=0.1)
    P = api.Problem(mesh, schedule=4)
    f = L.FieldCreate3D(P.mesh, 6)
    assert tuple(f.contents.shape) == (P.N, 6)
    m = f.contents.host.contents.len
    assert m == P.N * 6 and f.contents.device.contents.len == m and not f.contents.device.contents.is_host
    src = rng.normal(size=m)
    INIT = C.CFUNCTYPE(None, C.POINTER(f64), vp)

    def fill(p, ctx):
        np.ctypeslib.as_array(p, shape=(m,))[:] = src
    cb = INIT(fill)
    L.FieldInit(f, C.cast(cb, vp), None)
    assert np.array_equal(api.d2h(f.contents.device.contents.data, m, np.float64), src)    # Init uploads
    g = L.FieldCreate3D(P.mesh, 6)
    L.FieldCopy(g, f)
    assert np.array_equal(host_view(g.contents.host), src)
    assert np.array_equal(api.d2h(g.contents.device.contents.data, m, np.float64), src)
    L.ArrayScale(g.contents.device, 2.0)
    L.FieldUpdateHost(g)
    assert np.array_equal(host_view(g.contents.host), 2.0 * src)
    host_view(g.contents.host)[:] = 7.0
    L.FieldUpdateDevice(g)
    assert np.array_equal(api.d2h(g.contents.device.contents.data, m, np.float64), np.full(m, 7.0))
    path = str(tmp_path / "field.h5").encode()
    fh = H.H5OpenFile(path, b"w")
    H.FieldSave(f, fh, b"sol/step0")
    H.H5CloseFile(fh)
    assert np.array_equal(h5.read_dataset(path.decode(), "sol/step0", np.float64), src)
    fh = H.H5OpenFile(path, b"r")
    H.FieldLoad(g, fh, b"sol/step0")
    H.H5CloseFile(fh)
    assert np.array_equal(host_view(g.contents.host), src)
    assert np.array_equal(api.d2h(g.contents.device.contents.data, m, np.float64), src)   # Load refreshes the device copy
    L.FieldDestroy(f); L.FieldDestroy(g)
    P.close()


def test_pick_concurrent_stream_overlaps_the_library_stream(api):
    """DflPickConcurrentStream (host/comm_rccl.c): the stream the halo exchange / boundary rows run on must really run BESIDE
    the library stream -- a 2-ms resident wave on the library stream (dfl_spin_us, bounded) must not delay an empty launch on the
    picked stream.  Also: the spin kernel ends by itself (its bound), in about the time asked for."""
    import time
    L, H = api.lib(), api.hip()
    L.Init(0, None)
    L.DflPickConcurrentStream.restype, L.DflPickConcurrentStream.argtypes = vp, [vp]
    L.dfl_spin_us.argtypes = [C.c_int, vp]
    H.hipStreamSynchronize.argtypes = [vp]
    H.hipStreamDestroy.argtypes = [vp]
    main = L.DflStream()
    side = L.DflPickConcurrentStream(main)
    assert side and side != main
    api.sync()
    t0 = time.perf_counter()
    L.dfl_spin_us(2000, main)
    H.hipStreamSynchronize(C.c_void_p(main))
    dt = time.perf_counter() - t0
    assert 1.5e-3 < dt < 50e-3, dt                      # ~2 ms, and it ended
    # 20-ms wave on the library stream, then an empty launch on the side stream: the latter returns long before the former
    L.dfl_spin_us(20000, main)
    t0 = time.perf_counter()
    L.dfl_spin_us(1, C.c_void_p(side))
    H.hipStreamSynchronize(C.c_void_p(side))
    t_side = time.perf_counter() - t0
    H.hipStreamSynchronize(C.c_void_p(main))
    t_main = time.perf_counter() - t0
    assert t_side < 0.5 * t_main and t_main > 10e-3, (t_side, t_main)
    assert H.hipStreamDestroy(C.c_void_p(side)) == 0


def test_interleaved_matvec_is_bitwise_the_reference_layout_matvec(api, oracle_lib):
    """dfl_interleave4 + dfl_bcsr_spmv_x4 / DflMatrixFSMatVecX4Range (the matvec gathering x from x4[node][4] with one 16-byte
    load per lane) against the reference-layout kernel: the same products in the same order, so y is BITWISE equal; row ranges;
    MatrixMatVec takes the interleaved path by itself from 4096 nodes on and agrees with the oracle's matvec (1e-10).  The
    single-precision-values matvec of PC_TWOLEVEL's opt-in mixed-precision mode agrees with the double one to 1e-6."""
    m = kuhn_cube(17, jitter=0.2)                      # 5832 nodes: above the threshold of the automatic path
    wg, dwg = synthetic_fields(m)
    P = api.Problem(m, schedule=4)
    L = api.lib()
    try:
        N = P.N
        assert N >= 4096
        wg_d, dwg_d = api.DeviceArray.from_numpy(wg), api.DeviceArray.from_numpy(dwg)
        P.assemble_system(wg_d, dwg_d, None, want_J=True)
        x = np.random.default_rng(3).normal(size=6 * N)
        x_d, y_d, y2_d = api.DeviceArray.from_numpy(x), api.DeviceArray(6 * N), api.DeviceArray(6 * N)
        x4_d = api.DeviceArray(4 * N)
        L.dfl_interleave4.argtypes = [i32, i32, i32, vp, vp, vp]
        L.dfl_bcsr_spmv_rows.argtypes = [i32, i32, vp, vp, vp, f64, vp, f64, vp, vp]
        L.DflMatrixFSMatVecX4Range.argtypes = [C.POINTER(api.Matrix), vp, vp, i32, i32]
        fs = C.cast(P.J.contents.data, C.POINTER(api.MatrixFS)).contents
        spy = fs.spy1x1.contents
        L.dfl_interleave4(0, N, N, x_d.ptr, x4_d.ptr, L.DflStream())
        x4 = x4_d.numpy().reshape(N, 4)
        assert np.array_equal(x4[:, :3], x[:3 * N].reshape(N, 3)) and np.array_equal(x4[:, 3], x[3 * N:4 * N])
        L.dfl_bcsr_spmv_rows(N, N, spy.row_ptr, spy.col_ind, fs.block_val, 1.0, x_d.ptr, 0.0, y_d.ptr, L.DflStream())  # old gathers
        L.DflMatrixFSMatVecX4Range(P.J, x4_d.ptr, y2_d.ptr, 0, N)
        api.sync()
        y_ref, y_x4 = y_d.numpy()[:4 * N], y2_d.numpy()[:4 * N]
        assert np.array_equal(y_ref, y_x4)
        y2_d.zero()
        L.DflMatrixFSMatVecX4Range(P.J, x4_d.ptr, y2_d.ptr, 1000, 3000)       # a row range writes its rows only
        api.sync()
        yr = y2_d.numpy()
        rows = np.zeros(N, bool); rows[1000:3000] = True
        mask = np.concatenate([np.repeat(rows, 3), rows])
        assert np.array_equal(yr[:4 * N][mask], y_ref[mask]) and not yr[:4 * N][~mask].any()
        P.matvec(x_d, y2_d)                                                    # automatic path
        api.sync()
        assert np.array_equal(y2_d.numpy()[:4 * N], y_ref)
        S = oracle_lib.System(m)
        _, vals = S.assemble_system(wg, dwg, False, True)
        yo = S.matvec(vals, x)
        ok, e = close(y_ref, yo[:4 * N])
        assert ok, e
        # single-precision copy of the values
        L.dfl_bcsr_values_to_f32.argtypes = [C.c_int64, vp, vp, vp]
        L.dfl_bcsr_spmv_f32.argtypes = [i32, i32, vp, vp, vp, vp, vp, vp]
        valf = api.DeviceArray(spy.nnz * 16, np.float32)
        L.dfl_bcsr_values_to_f32(spy.nnz * 16, fs.block_val, valf.ptr, L.DflStream())
        y2_d.zero()
        L.dfl_bcsr_spmv_f32(N, N, spy.row_ptr, spy.col_ind, valf.ptr, x_d.ptr, y2_d.ptr, L.DflStream())
        api.sync()
        assert np.abs(y2_d.numpy()[:4 * N] - y_ref).max() <= 1e-6 * np.abs(y_ref).max()
    finally:
        P.close()
