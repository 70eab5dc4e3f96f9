"""Multi-rank tests (torch.distributed.run, gloo).  CPU leg: the distributed algorithm with
the oracle as local compute (world_size 2 and 3).  GPU leg: the product path -- C GMRES +
HIP kernels + DflComm callbacks -- with 2 and 3 ranks sharing the one GPU of the box."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _launch(mode, world, M, its, timeout=600, extra_env=None):
    env = dict(os.environ)
    env.update(extra_env or {})
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["OMP_NUM_THREADS"] = "2"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dist_worker.py"), mode, str(M), str(its)]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return r.stdout


@pytest.mark.parametrize("world", [2, 3, 8])
def test_distributed_algorithm_cpu_gloo(world, oracle_lib):
    """World 8 = the rank count of the driver's scaling run: partition, ownership, halo plan (every rank has up to 7
    neighbours), the build-once-and-scatter setup of bench.py and the distributed GMRES on the oracle's local compute."""
    subprocess.check_call(["make", "-s", "-j8", "-C", ROOT])
    out = _launch("cpu", world, 10 if world == 8 else 6, 25)
    assert "DIST_CPU_OK" in out


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_distributed_product_path_gpu_gloo(world, oracle_lib):
    """(A GPU box allows 6 processes on its card, and the test runner and the launcher count: world 5 was rehearsed by hand
    -- `torch.distributed.run --nproc-per-node 5 tests/dist_worker.py gpu 10 30` passes -- but does not fit in the suite.)"""
    out = _launch("gpu", world, 8, 30)
    assert "DIST_GPU_OK" in out


@pytest.mark.gpu
def test_distributed_time_step_gpu_gloo(oracle_lib):
    """SolveFlowSystem / DflTimeStep on a 2-way partition: ghost residual zeroing, all-reduced Newton norms,
    halo exchange of the Newton increment."""
    out = _launch("gpu_step", 2, 6, 0)
    assert "DIST_STEP_OK" in out


@pytest.mark.gpu
def test_rccl_communicator_single_rank_gpu(oracle_lib):
    """C-level RCCL DflComm (host/comm_rccl.c) on the nccl backend, world_size 1 (one GPU per box; RCCL refuses two
    ranks on one device): bootstrap, in-stream all-reduce, verification against torch.distributed, solve parity."""
    out = _launch("gpu_rccl", 1, 8, 30)
    assert "DIST_RCCL_OK" in out


@pytest.mark.gpu
@pytest.mark.parametrize("fused_kernel", [True, False])
def test_distributed_fused_norm_option_gpu_gloo(oracle_lib, fused_kernel):
    """KrylovSetFusedNorm: h and w.w in ONE all-reduce per Arnoldi step, ||w - Qh|| from the Pythagorean identity; same
    residual history as the single-domain oracle within the partitioned-run tolerance, no cancellation flag.  Both forms of
    the step: update + Givens + next preconditioner application in one launch (ranks up to 500k owned nodes) and as three
    launches (larger ranks; forced here with DFL_NO_FUSED_UPDATE_PC)."""
    env = {"DFL_FUSED_NORM": "1"}
    if not fused_kernel:
        env["DFL_NO_FUSED_UPDATE_PC"] = "1"
    out = _launch("gpu", 2, 8, 30, extra_env=env)
    assert "DIST_GPU_OK" in out


@pytest.mark.gpu
@pytest.mark.parametrize("mode,world", [("gpu", 2), ("gpu_rccl", 1), ("gpu_twolevel", 2)])
def test_distributed_interleaved_matvec_gpu(mode, world, oracle_lib):
    """The matvec gathering from the interleaved copy of its input (dfl_bcsr_spmv_x4; by default only for >= 4096 nodes) forced
    onto the small test meshes: partitioned GMRES with split rows (owned part of the copy from the producer kernel, ghost part
    behind the unpack), the C-level RCCL communicator with its side stream, and FGMRES + PC_TWOLEVEL -- same checks as the
    default path (the result is bitwise the same matvec)."""
    out = _launch(mode, world, 14 if mode == "gpu_twolevel" else 8, 0 if mode == "gpu_twolevel" else 30, extra_env={"DFL_SPMV_X4_MIN": "1"})
    assert {"gpu": "DIST_GPU_OK", "gpu_rccl": "DIST_RCCL_OK", "gpu_twolevel": "DIST_TWOLEVEL_OK"}[mode] in out


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_distributed_twolevel_gpu_gloo(world, oracle_lib):
    """PC_TWOLEVEL on element-partitioned matrices: per-rank aggregates, replicated Galerkin coarse
    problem, rank-local DILU smoother; iteration count within 20 % of the one-process solve, same solution, coarse matrix
    = P^T A P of the global matrix."""
    out = _launch("gpu_twolevel", world, 14, 0)
    assert "DIST_TWOLEVEL_OK" in out


@pytest.mark.gpu
def test_distributed_transient_twolevel_gpu_gloo(oracle_lib):
    """BASELINE config 5 in small on 2 ranks: 3 time steps of 2 Newton iterations on the partitioned mesh
    with PC_TWOLEVEL, every solve converged, Newton counts / residuals / final states equal to the one-process transient."""
    out = _launch("gpu_transient_twolevel", 2, 14, 3)
    assert "DIST_TRANSIENT_TWOLEVEL_OK" in out


@pytest.mark.gpu
def test_distributed_pipelined_gmres_gpu_gloo(oracle_lib):
    """KrylovSetPipelined (p(1)-GMRES: one reduction per step, overlapped with the next matvec through the auxiliary basis
    z = A M^-1 v) on a 2-way partition: residual history of the single-domain oracle at the loosened tolerance 1e-6 r0, same
    solution, one all-reduce per step, no cancellation flag.  (The reduction's own stream is exercised by the RCCL
    communicator test; the gloo callbacks are host-synchronous.)"""
    out = _launch("gpu", 2, 8, 30, extra_env={"DFL_PIPELINED": "1"})
    assert "DIST_GPU_OK" in out
