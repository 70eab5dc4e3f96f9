"""The drop-in claim in executable form: tests/c_abi/dedflow_main.c is a plain C99 program against include/dedflow.h that
follows the reference's src/main.c (HDF5 mesh in, field-split matrix, coloring, Dirichlet groups, generalized-alpha time
loop, sol.<k>.h5 out).  It is compiled with gcc here, run on the GPU, and its solution file is compared with the
oracle-backed restatement of the time step (tests/ref_driver.py) on the same inputs."""
import os
import subprocess
import sys

import numpy as np
import pytest

from dedflow_amd.meshgen import kuhn_cube, synthetic_fields

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_c_host_program_time_step_matches_oracle_driver(tmp_path, oracle_lib):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from ref_driver import time_step
    from dedflow_amd import api, h5
    if not os.path.exists(os.path.join(ROOT, "dedflow_amd", "libdedflow_h5.so")):
        pytest.skip("libdedflow_h5.so not built (no HDF5 headers)")
    exe = str(tmp_path / "dedflow_main")
    lib_dir = os.path.join(ROOT, "dedflow_amd")
    subprocess.check_call(["gcc", "-std=gnu99", "-O1", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(ROOT, "include"),
                           "-I/opt/rocm/include", os.path.join(ROOT, "tests", "c_abi", "dedflow_main.c"), "-o", exe,
                           "-L" + lib_dir, "-ldedflow", "-ldedflow_h5", "-Wl,-rpath," + lib_dir])
    m = kuhn_cube(5, jitter=0.2)
    S = oracle_lib.System(m)
    N = S.N
    wg0, dw0 = synthetic_fields(m)
    wgold = wg0.copy()
    wgold[3 * N:4 * N] = 0.0
    dwgold = 0.1 * dw0
    it_o, rn_o, ri_o, wgold_o, dwgold_o, dwg_o = time_step(S, wgold, dwgold, dwgold.copy(), maxit=2)
    # inputs in the reference's file formats
    mesh_file, sol0 = str(tmp_path / "box.h5"), str(tmp_path / "sol.0.h5")
    h5.write_mesh(mesh_file, m)
    a, b = api.DeviceArray.from_numpy(wgold), api.DeviceArray.from_numpy(dwgold)
    h5.lib().DflSolutionWriteH5(sol0.encode(), N, a.ptr, b.ptr)
    api.sync()
    out = subprocess.run([exe, mesh_file, sol0, str(tmp_path / "sol"), "1", "2"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "step 1: %d newton iterations" % it_o in out.stdout, out.stdout
    sol1 = str(tmp_path / "sol.1.h5")
    u = h5.read_dataset(sol1, "u", np.float64)
    assert np.abs(u - wgold_o[:3 * N]).max() <= 1e-7 * np.abs(wgold_o[:3 * N]).max()
    for name, ref in (("phi", wgold_o[4 * N:5 * N]), ("T", wgold_o[5 * N:]), ("du", dwgold_o[:3 * N]), ("p", dwgold_o[3 * N:4 * N]),
                      ("dphi", dwgold_o[4 * N:5 * N]), ("dT", dwgold_o[5 * N:])):
        got = h5.read_dataset(sol1, name, np.float64)
        assert np.abs(got - ref).max() <= 1e-7 * max(np.abs(ref).max(), 1e-300), name
    # the same program with the two build-defined preconditioners (one KrylovSetPCType call each): every solve reaches the
    # same tolerance, so the solution files agree to the solver's accuracy
    for pc in ("ilu0", "twolevel"):
        pref = str(tmp_path / ("sol_" + pc))
        out = subprocess.run([exe, mesh_file, sol0, pref, "1", "2", pc], capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
        up = h5.read_dataset(pref + ".1.h5", "u", np.float64)
        assert np.abs(up - u).max() <= 2e-3 * np.abs(u).max(), (pc, np.abs(up - u).max() / np.abs(u).max())
