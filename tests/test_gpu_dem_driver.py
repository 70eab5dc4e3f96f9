"""GPU tests of the two "caller-side" pieces: the DEM contact sweep (build-defined, parity
unpinned: the reference's Particle.c has no physics) and the generalized-alpha
Newton driver (SolveFlowSystem / time step, src/main.c:77-283,535-565) against the oracle-backed
restatement in tests/ref_driver.py."""
import numpy as np
import pytest

from dedflow_amd.meshgen import dem_particles, kuhn_cube, synthetic_fields

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api():
    from dedflow_amd import api as A
    A.lib()
    return A


@pytest.mark.parametrize("P,R", [(3000, 0.03), (100000, 0.004), (20000, 0.002), (300, 0.2)])
def test_dem_forces_match_oracle(api, oracle_lib, P, R):
    """(20000, 0.002): the grid is sized by the particle count (34^3 cells of 15 R: cells far wider than the interaction
    range); (300, 0.2): 4R > 1/2, a single cell = all pairs."""
    x, v, _ = dem_particles(P, R)
    pc = api.Particles(x, v, R, mass=1.0, kn=1.0e4, gamma_n=1.0)
    try:
        pc.compute_forces()
        api.sync()
        _, _, acc = pc.arrays()
        ref, tested = oracle_lib.dem_forces(x, v, R)
        assert np.count_nonzero(ref) > 0 and tested > 0
        assert np.abs(acc - ref).max() <= 1e-10 * np.abs(ref).max()
        if P <= 3000:  # cell list == all-pairs
            brute, _ = oracle_lib.dem_forces(x, v, R, brute=True)
            assert np.abs(acc - brute).max() <= 1e-10 * np.abs(brute).max()
        # bitwise reproducible (stable sort => fixed summation order)
        pc.compute_forces()
        api.sync()
        assert np.array_equal(pc.arrays()[2], acc)
    finally:
        pc.close()


def test_dem_properties_and_update(api, oracle_lib):
    # two overlapping spheres in the interior: equal and opposite forces along the line of centres
    R = 0.05
    x = np.array([0.5, 0.5, 0.5, 0.5 + 1.6 * R, 0.5, 0.5])
    v = np.zeros(6)
    pc = api.Particles(x, v, R, mass=2.0, kn=100.0, gamma_n=0.0, dt=1e-3)
    try:
        pc.compute_forces()
        api.sync()
        acc = pc.arrays()[2]
        f = 100.0 * (2 * R - 1.6 * R) / 2.0
        assert np.allclose(acc, [-f, 0, 0, f, 0, 0], atol=1e-12)
        pc.update()
        api.sync()
        xn, vn, an = pc.arrays()
        assert np.allclose(vn, 1e-3 * an) and np.allclose(xn, x + 1e-3 * vn)
    finally:
        pc.close()


def test_solve_flow_system_matches_oracle_driver(api, oracle_lib):
    from ref_driver import solve_flow_system
    m = kuhn_cube(6, jitter=0.2)
    S = oracle_lib.System(m)
    N = S.N
    wg0, dw0 = synthetic_fields(m)
    wgold = wg0.copy()
    wgold[3 * N:4 * N] = 0.0
    dwgold = 0.1 * dw0
    dwg = 0.05 * dw0
    it_o, rn_o, ri_o, dwg_o, F_o, gm = solve_flow_system(S, wgold, dwgold, dwg, maxit=2)
    P = api.Problem(m)
    try:
        d = [api.DeviceArray.from_numpy(a) for a in (wgold, dwgold, dwg)]
        F_d, dx_d = api.DeviceArray(6 * N), api.DeviceArray(6 * N)
        it, rn, ri = P.solve_flow_system(d[0], d[1], d[2], F_d, dx_d, maxit=2)
        api.sync()
        assert it == it_o
        assert np.allclose(ri, ri_o, rtol=1e-10, atol=1e-10 * ri_o.max())
        # two Newton steps each with a GMRES solve stopped at rtol 1e-4: compare at solver accuracy
        assert np.allclose(rn, rn_o, rtol=1e-6, atol=1e-8 * ri_o.max())
        assert np.abs(d[2].numpy() - dwg_o).max() <= 1e-7 * np.abs(dwg_o).max()
    finally:
        P.close()


def test_coupled_time_step(api, oracle_lib):
    """config 4 shape: fluid step + DEM sub-steps in one call; fluid state against the oracle driver."""
    from ref_driver import time_step
    m = kuhn_cube(5, jitter=0.2)
    S = oracle_lib.System(m)
    N = S.N
    wg0, dw0 = synthetic_fields(m)
    wgold = wg0.copy()
    wgold[3 * N:4 * N] = 0.0
    dwgold = 0.1 * dw0
    dwg = dwgold.copy()
    it_o, rn_o, ri_o, wgold_o, dwgold_o, dwg_o = time_step(S, wgold, dwgold, dwg, maxit=2)
    x, v, R = dem_particles(2000, 0.03)
    P = api.Problem(m)
    pc = api.Particles(x, v, R, dt=1e-4)
    try:
        d = [api.DeviceArray.from_numpy(a) for a in (wgold, dwgold, dwg)]
        F_d, dx_d = api.DeviceArray(6 * N), api.DeviceArray(6 * N)
        it, rn, ri = P.time_step(d[0], d[1], d[2], F_d, dx_d, newton_maxit=2, particles=pc, dem_substeps=3)
        api.sync()
        assert it == it_o
        assert np.abs(d[0].numpy() - wgold_o).max() <= 1e-7 * np.abs(wgold_o).max()
        assert np.abs(d[1].numpy() - dwgold_o).max() <= 1e-7 * np.abs(dwgold_o).max()
        xn, vn, an = pc.arrays()
        assert np.all(np.isfinite(xn)) and np.abs(xn - x).max() > 0.0
    finally:
        pc.close()
        P.close()


def test_device_pool_first_fit_and_coalescing(api):
    """Large DEVICE allocations of the default allocator come out of the pool reserved at Init (host/runtime.c):
    zero-filled, first fit, neighbours coalesce on free, small requests bypass the pool."""
    import ctypes as C
    L = api.lib()
    L.ArrayCreateDevice.restype = C.POINTER(api.Array)
    L.ArrayCreateDevice.argtypes = [C.c_int32]
    L.ArrayDestroy.argtypes = [C.POINTER(api.Array)]
    L.DflDevicePoolStats.argtypes = [C.POINTER(C.c_int64), C.POINTER(C.c_int64)]

    def stats():
        r, u = C.c_int64(0), C.c_int64(0)
        L.DflDevicePoolStats(C.byref(r), C.byref(u))
        return r.value, u.value

    reserved, base = stats()
    if reserved == 0:
        pytest.skip("device pool disabled (DFL_DEVICE_POOL_GB=0)")
    n = 4 << 20                                   # 32 MiB of f64 each
    a, b, c = (L.ArrayCreateDevice(n) for _ in range(3))
    pa, pb, pc = (C.cast(v.contents.data, C.c_void_p).value for v in (a, b, c))
    assert stats()[1] == base + 3 * 8 * n and pb == pa + 8 * n and pc == pb + 8 * n
    assert not api.DeviceArray(n, np.float64, ptr=pb, owner=False).numpy().any()      # zero-filled
    small = L.ArrayCreateDevice(1000)             # below the pool threshold: plain hipMalloc
    assert stats()[1] == base + 3 * 8 * n
    L.ArrayDestroy(a); L.ArrayDestroy(b)          # two neighbours -> one free block of 64 MiB
    assert stats()[1] == base + 8 * n
    d = L.ArrayCreateDevice(2 * n)
    assert C.cast(d.contents.data, C.c_void_p).value == pa
    L.ArrayDestroy(d); L.ArrayDestroy(c); L.ArrayDestroy(small)
    assert stats() == (reserved, base)
