"""ctypes view of libdedflow.so (the C host layer + HIP kernels) for the Python
harness (tests, bench.py, __graft_entry__).  Python never computes anything on
the path: it generates synthetic inputs, calls the C object API in the order of
the reference driver (``src/main.c:372-477``) and copies results back.

The library is the product; if it is missing, importing this module's
``lib()`` raises -- there is no CPU or PyTorch fallback.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_HIP = None

c_i32p = C.POINTER(C.c_int32)
c_f64p = C.POINTER(C.c_double)
vp = C.c_void_p


class MissingExtension(RuntimeError):
    pass


def lib_path() -> str:
    # DFL_LIB: an alternative build of the same library (`make asan`: host layer under AddressSanitizer / UBSan, CPU runs only)
    return os.environ.get("DFL_LIB") or os.path.join(_HERE, "libdedflow.so")


def lib():
    """Load libdedflow.so; fail loudly (no fallback) when it has not been built."""
    global _LIB
    if _LIB is None:
        p = lib_path()
        if not os.path.exists(p):
            raise MissingExtension(
                f"{p} not found: build it with `make` (or __graft_entry__.build()); the hot path has no CPU fallback")
        L = C.CDLL(p, mode=C.RTLD_GLOBAL)
        _declare(L)
        _LIB = L
    return _LIB


def hip():
    global _HIP
    if _HIP is None:
        H = C.CDLL("libamdhip64.so", mode=C.RTLD_GLOBAL)
        H.hipMalloc.argtypes = [C.POINTER(vp), C.c_size_t]
        H.hipFree.argtypes = [vp]
        H.hipMemcpy.argtypes = [vp, vp, C.c_size_t, C.c_int]
        H.hipMemset.argtypes = [vp, C.c_int, C.c_size_t]
        H.hipDeviceSynchronize.argtypes = []
        H.hipEventCreate.argtypes = [C.POINTER(vp)]
        H.hipEventRecord.argtypes = [vp, vp]
        H.hipEventSynchronize.argtypes = [vp]
        H.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), vp, vp]
        H.hipEventDestroy.argtypes = [vp]
        H.hipGetErrorString.restype = C.c_char_p
        H.hipSetDevice.argtypes = [C.c_int]
        _HIP = H
    return _HIP


def _chk(e):
    if e != 0:
        raise RuntimeError("HIP error %d: %s" % (e, hip().hipGetErrorString(e).decode()))


H2D, D2H, D2D = 1, 2, 3


class DeviceArray:
    """Raw device buffer (hipMalloc, zero-filled) with numpy transfer helpers."""

    def __init__(self, n, dtype=np.float64, ptr=None, owner=True):
        self.dtype = np.dtype(dtype)
        self.n = int(n)
        self.nbytes = self.n * self.dtype.itemsize
        self.owner = owner and ptr is None
        if ptr is None:
            p = vp()
            _chk(hip().hipMalloc(C.byref(p), max(self.nbytes, 16)))
            _chk(hip().hipMemset(p, 0, max(self.nbytes, 16)))
            self.ptr = p.value
        else:
            self.ptr = int(ptr)

    @classmethod
    def from_numpy(cls, a):
        a = np.ascontiguousarray(a)
        d = cls(a.size, a.dtype)
        d.upload(a)
        return d

    def upload(self, a):
        a = np.ascontiguousarray(a, dtype=self.dtype)
        assert a.size == self.n
        _chk(hip().hipMemcpy(self.ptr, a.ctypes.data, self.nbytes, H2D))

    def numpy(self):
        out = np.empty(self.n, self.dtype)
        if self.nbytes:
            _chk(hip().hipMemcpy(out.ctypes.data, self.ptr, self.nbytes, D2H))
        return out

    def zero(self):
        _chk(hip().hipMemset(self.ptr, 0, max(self.nbytes, 1)))

    def view(self, offset, n):
        return DeviceArray(n, self.dtype, ptr=self.ptr + offset * self.dtype.itemsize)

    def free(self):
        if self.owner and self.ptr:
            hip().hipFree(self.ptr)
            self.ptr = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def d2h(ptr, n, dtype):
    return DeviceArray(n, dtype, ptr=ptr).numpy()


def sync():
    _chk(hip().hipDeviceSynchronize())


class Timer:
    """hipEvent pair on the library stream (the null stream unless DflSetStream was called)."""

    def __init__(self):
        self.a, self.b = vp(), vp()
        _chk(hip().hipEventCreate(C.byref(self.a)))
        _chk(hip().hipEventCreate(C.byref(self.b)))

    def start(self):
        _chk(hip().hipEventRecord(self.a, lib().DflStream()))

    def stop(self):
        _chk(hip().hipEventRecord(self.b, lib().DflStream()))

    def ms(self):
        _chk(hip().hipEventSynchronize(self.b))
        t = C.c_float(0)
        _chk(hip().hipEventElapsedTime(C.byref(t), self.a, self.b))
        return float(t.value)


# ---- struct mirrors of include/dedflow.h ---------------------------------------------
class Mesh3DData(C.Structure):
    _fields_ = [("is_host", C.c_int32), ("num_node", C.c_int32), ("num_tet", C.c_int32), ("num_prism", C.c_int32),
                ("num_hex", C.c_int32), ("xg", vp), ("ien", vp)]


class Mesh3D(C.Structure):
    _fields_ = [("num_node", C.c_int32), ("num_tet", C.c_int32), ("num_prism", C.c_int32), ("num_hex", C.c_int32),
                ("host", C.POINTER(Mesh3DData)), ("device", C.POINTER(Mesh3DData)),
                ("num_bound", C.c_int32), ("bound_fid", vp), ("bound_node_offset", vp), ("bound_node", vp),
                ("bound_elem_offset", vp), ("bound_ien", vp), ("bound_f2e", vp), ("bound_forn", vp),
                ("num_batch", C.c_int32), ("batch_offset", vp), ("batch_ind", vp),
                ("num_color", C.c_int32), ("color", vp), ("ext", vp)]


class CSRAttr(C.Structure):
    pass


CSRAttr._fields_ = [("num_row", C.c_int32), ("num_col", C.c_int32), ("nnz", C.c_int32), ("row_ptr", vp), ("col_ind", vp),
                    ("parent", C.POINTER(CSRAttr))]


class Matrix(C.Structure):
    _fields_ = [("size", C.c_int32 * 2), ("type", C.c_int), ("data", vp), ("stream_ref", vp), ("op", vp * 15)]


class MatrixCSR(C.Structure):
    _fields_ = [("external_attr", C.c_int32), ("attr", C.POINTER(CSRAttr)), ("val", vp), ("descr", vp),
                ("buffer_size", C.c_int32), ("buffer", vp), ("owner", vp), ("owner_slot", C.c_int32)]


class MatrixFS(C.Structure):
    _fields_ = [("n_offset", C.c_int32), ("offset", vp), ("d_offset", vp), ("stream", vp), ("spy1x1", C.POINTER(CSRAttr)),
                ("d_matval", vp), ("mat", C.POINTER(C.POINTER(Matrix))), ("block_mode", C.c_int32), ("block_val", vp)]


class KrylovStats(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("rnrm_init", C.c_double), ("res_hist", C.c_double * 512), ("converged", C.c_int32),
                ("fused_norm_cancelled", C.c_int32), ("total_solves", C.c_int32), ("total_converged", C.c_int32),
                ("total_iterations", C.c_int64)]


class Dirichlet(C.Structure):
    _fields_ = [("mesh", vp), ("face_ind", C.c_int32), ("shape", C.c_int32), ("buffer_size", C.c_size_t), ("buffer", vp)]
    # followed by BCType bctype[shape]


class Array(C.Structure):
    _fields_ = [("is_host", C.c_int32), ("len", C.c_int32), ("data", vp)]


class ParticleContext(C.Structure):
    _fields_ = [("num_particle", C.c_int32), ("num_pointwise_dof", C.c_int32), ("h_arr", C.POINTER(Array) * 3),
                ("d_arr", C.POINTER(Array) * 3), ("buff", C.c_double * 2), ("ext", vp)]


class Particles:
    """ParticleContext (src/Particle.h) + the build-defined contact sweep."""

    def __init__(self, coord, vel, radius, mass=1.0, kn=1.0e4, gamma_n=1.0, dt=1.0e-4):
        L = lib()
        L.Init(0, None)
        P = coord.size // 3
        self.P = P
        self.ctx = L.ParticleContextCreate(P)
        c = self.ctx.contents
        c.buff[0], c.buff[1] = mass, radius
        C.memmove(c.h_arr[0].contents.data, np.ascontiguousarray(coord).ctypes.data, 24 * P)
        C.memmove(c.h_arr[1].contents.data, np.ascontiguousarray(vel).ctypes.data, 24 * P)
        L.ParticleContextUpdateDevice(self.ctx)
        L.ParticleContextSetContactModel(self.ctx, kn, gamma_n, dt)

    def compute_forces(self):
        lib().ParticleContextComputeForces(self.ctx)

    def update(self):
        lib().ParticleContextUpdate(self.ctx)

    def arrays(self):
        """(coord, vel, acc) copied back from the device"""
        c = self.ctx.contents
        return tuple(d2h(c.d_arr[k].contents.data, 3 * self.P, np.float64) for k in range(3))

    def close(self):
        lib().ParticleContextDestroy(self.ctx)


PC_DECOMPOSITION, PC_ILU0, PC_TWOLEVEL = 0x2, 0x5, 0x6   # PCType values (include/dedflow.h)
ALLREDUCE_FN = C.CFUNCTYPE(None, vp, vp, C.c_int32)
HALO_FN = C.CFUNCTYPE(None, vp, vp)
STREAM_FN = C.CFUNCTYPE(vp, vp)


class DflComm(C.Structure):
    _fields_ = [("allreduce_sum", ALLREDUCE_FN), ("halo_exchange", HALO_FN), ("ctx", vp), ("num_owned_node", C.c_int32),
                ("halo_begin", HALO_FN), ("halo_end", HALO_FN), ("num_interior_node", C.c_int32),
                ("rank", C.c_int), ("world", C.c_int), ("halo_stream", STREAM_FN)]


def _declare(L):
    def f(name, res, args):
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    i32, f64 = C.c_int32, C.c_double
    f("Init", None, [C.c_int, vp]); f("Finalize", None, [])
    f("DflStream", vp, []); f("DflSetStream", None, [vp]); f("DflSetQuiet", None, [i32])
    f("DflSetAssemblySchedule", None, [C.c_int]); f("DflSetPatchParameters", None, [i32, i32])
    f("DflSetRowPatchParameters", None, [i32, i32]); f("dfl_tune_asm", None, [C.c_int])
    f("DflMeshSetAssemblySchedule", None, [C.POINTER(Mesh3D), C.c_int]); f("DflMeshSetWeakBCGroup", None, [C.POINTER(Mesh3D), i32])
    f("DflSetWeakBCGroup", None, [i32])
    f("DflSetSlotPatchParameters", None, [i32, i32, i32]); f("DflSetRhsWaveParameters", None, [i32, i32])
    f("DflSetRhsPatchParameters", None, [i32, i32]); f("DflMeshGeometryChanged", None, [C.POINTER(Mesh3D)])
    f("Mesh3DCreate", C.POINTER(Mesh3D), [i32, i32, i32, i32]); f("Mesh3DDestroy", None, [C.POINTER(Mesh3D)])
    f("Mesh3DUpdateDevice", None, [C.POINTER(Mesh3D)]); f("Mesh3DGenerateColorBatch", None, [C.POINTER(Mesh3D)])
    f("Mesh3DSetBound", None, [C.POINTER(Mesh3D), i32, vp, vp, vp, vp, vp])
    f("CSRAttrCreate", C.POINTER(CSRAttr), [C.POINTER(Mesh3D)]); f("CSRAttrDestroy", None, [C.POINTER(CSRAttr)])
    f("CSRAttrCreateBlock", C.POINTER(CSRAttr), [C.POINTER(CSRAttr), i32, i32])
    f("MatrixCreateTypeCSR", C.POINTER(Matrix), [C.POINTER(CSRAttr), vp])
    f("MatrixCreateTypeFS", C.POINTER(Matrix), [i32, vp, vp])
    f("MatrixDestroy", None, [C.POINTER(Matrix)]); f("MatrixSetup", None, [C.POINTER(Matrix)])
    f("MatrixZero", None, [C.POINTER(Matrix)])
    f("MatrixAMVPBY", None, [C.POINTER(Matrix), f64, vp, f64, vp]); f("MatrixMatVec", None, [C.POINTER(Matrix), vp, vp])
    f("MatrixGetDiag", None, [C.POINTER(Matrix), vp, i32])
    f("MatrixFSBlockValues", vp, [C.POINTER(Matrix)]); f("MatrixFSUseReferenceLayout", None, [C.POINTER(Matrix), i32])
    f("MatrixAddElemValueBlockedBatched", None, [C.POINTER(Matrix), i32, i32, vp, vp, i32, i32, vp, C.c_int, C.c_int, vp])
    f("MatrixMatVecWithMask", None, [C.POINTER(Matrix), vp, vp, vp, vp]); f("MatrixZeroRow", None, [C.POINTER(Matrix), i32, vp, i32, f64])
    f("VecAXPY", None, [f64, vp, vp, i32]); f("VecPointwiseMult", None, [vp, vp, vp, i32]); f("VecPointwiseDiv", None, [vp, vp, vp, i32])
    f("VecPointwiseInv", None, [vp, i32])
    f("MatrixFSExportSubmatrices", None, [C.POINTER(Matrix)]); f("MatrixFSImportSubmatrices", None, [C.POINTER(Matrix)])
    f("DirichletCreate", C.POINTER(Dirichlet), [C.POINTER(Mesh3D), i32, i32]); f("DirichletDestroy", None, [C.POINTER(Dirichlet)])
    f("DirichletApplyVec", None, [C.POINTER(Dirichlet), vp]); f("DirichletApplyMat", None, [C.POINTER(Dirichlet), C.POINTER(Matrix)])
    f("KrylovCreateGMRES", vp, [i32, f64, f64, vp]); f("KrylovCreateCG", vp, [i32, f64, f64, vp])
    f("KrylovDestroy", None, [vp]); f("KrylovSolve", None, [vp, C.POINTER(Matrix), vp, vp])
    f("KrylovGetStats", C.POINTER(KrylovStats), [vp]); f("KrylovSetCheckInterval", None, [vp, i32])
    f("KrylovSetVerbose", None, [vp, i32]); f("KrylovSetComm", None, [vp, C.POINTER(DflComm)])
    f("PCSetup", None, [vp]); f("PCApply", None, [vp, vp, vp]); f("PCDestroy", None, [vp])
    f("PCCreateJacobi", vp, [C.POINTER(Matrix), i32, vp]); f("PCCreateNone", vp, [C.POINTER(Matrix), i32])
    f("PCCreateDILU", vp, [C.POINTER(Matrix)]); f("PCDILUGetColors", i32, [vp, vp]); f("PCDILUGetInverseBlocks", vp, [vp])
    f("KrylovSetPCType", None, [vp, C.c_int]); f("KrylovGetPC", vp, [vp]); f("KrylovSetFusedNorm", None, [vp, C.c_int]); f("KrylovSetPipelined", None, [vp, C.c_int]); f("KrylovSetRestart", None, [vp, i32])
    f("KrylovSetFlexible", None, [vp, i32]); f("KrylovSetMesh", None, [vp, C.POINTER(Mesh3D)]); f("KrylovSetAggregateSize", None, [vp, i32])
    f("PCTwoLevelInfo", None, [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(C.c_int64)]); f("PCTwoLevelAggregates", vp, [vp])
    f("PCTwoLevelCoarseMatrix", C.POINTER(Matrix), [vp]); f("PCTwoLevelSetInner", None, [vp, i32, f64])
    f("AssembleSystemTet", None, [C.POINTER(Mesh3D), vp, vp, vp, C.POINTER(Matrix)])
    f("AssembleSystemTetFace", None, [C.POINTER(Mesh3D), vp, vp, vp, C.POINTER(Matrix)])
    f("AssembleSystem", None, [C.POINTER(Mesh3D), vp, vp, vp, C.POINTER(Matrix), vp, i32])
    # kernel-level C ABI used directly by tests / bench
    f("dfl_bcsr_spmv", None, [i32, vp, vp, vp, f64, vp, f64, vp, vp])
    f("dfl_cgs_work_size", C.c_int64, [i32, i32]); f("dfl_reduce_work_size", i32, [])
    f("dfl_cgs_dots", None, [i32, i32, vp, C.c_int64, vp, vp, vp, vp])
    f("dfl_cgs_update", None, [i32, i32, vp, C.c_int64, vp, vp, vp, C.c_int, vp, vp])
    f("dfl_ddot", None, [i32, vp, vp, vp, vp, vp]); f("dfl_dnrm2", None, [i32, vp, vp, vp, vp])
    f("dfl_daxpy", None, [i32, f64, vp, vp, vp]); f("dfl_dscal", None, [i32, f64, vp, vp])
    f("dfl_pc_jacobi_setup", None, [i32, vp, vp, vp, vp, vp, vp]); f("dfl_pc_jacobi_apply", None, [i32, i32, vp, vp, vp, vp, vp])
    f("dfl_assemble_tet_lhs", None, [i32, vp, vp, vp, vp, vp, vp])
    f("dfl_assemble_tet_rhs", None, [i32, vp, vp, vp, vp])
    f("dfl_pack_nodes", None, [i32, vp, vp, vp, vp, vp]); f("dfl_unpack_rhs", None, [i32, vp, vp, vp])
    f("ParticleContextCreate", C.POINTER(ParticleContext), [i32]); f("ParticleContextDestroy", None, [C.POINTER(ParticleContext)])
    f("ParticleContextUpdateDevice", None, [C.POINTER(ParticleContext)]); f("ParticleContextUpdateHost", None, [C.POINTER(ParticleContext)])
    f("ParticleContextSetContactModel", None, [C.POINTER(ParticleContext), f64, f64, f64])
    f("ParticleContextComputeForces", None, [C.POINTER(ParticleContext)]); f("ParticleContextUpdate", None, [C.POINTER(ParticleContext)])
    f("SolveFlowSystem", i32, [C.POINTER(Mesh3D), vp, vp, vp, C.POINTER(Matrix), vp, vp, vp, vp, i32, i32, vp, vp])
    f("DflTimeStep", i32, [C.POINTER(Mesh3D), vp, vp, vp, C.POINTER(Matrix), vp, vp, vp, vp, i32, i32, C.POINTER(ParticleContext),
                           i32, vp, vp])
    f("DflDevicePoolStats", None, [C.POINTER(C.c_int64), C.POINTER(C.c_int64)])
    f("DflKrylovCalibrationLog", C.c_char_p, []); f("DflDeviceMemoryInUse", C.c_int64, []); f("DflWaitDeviceMemoryQuiet", C.c_double, [C.c_double])
    f("DflProfileEnable", None, [C.c_int]); f("DflProfileCollect", C.c_int, [C.c_int, C.POINTER(f64), C.POINTER(f64)])
    f("GenerateRandomColor", None, [vp, i32, i32])
    f("dfl_abi_version", C.c_int, [])


BC_STRONG = 1
REFERENCE_BCS = [(0, (1, 1, 1)), (2, (0, 1, 0)), (3, (0, 0, 1)), (4, (0, 0, 0))]  # src/main.c:454-476


class Problem:
    """The reference driver's setup sequence (src/main.c:362-477) through the C API."""

    def __init__(self, mesh, maxit=120, atol=1e-12, rtol=1e-4, quiet=True, bcs=REFERENCE_BCS, color=True, schedule=4,
                 reference_layout=False):
        L = lib()
        L.Init(0, None)
        L.DflSetQuiet(1 if quiet else 0)
        L.DflSetAssemblySchedule(int(schedule))
        self.mesh_np = mesh
        self.N, self.T = mesh.num_node, mesh.num_tet
        self.mesh = L.Mesh3DCreate(self.N, self.T, 0, 0)
        m = self.mesh.contents
        C.memmove(m.host.contents.xg, mesh.xg.ctypes.data, mesh.xg.nbytes)
        C.memmove(m.host.contents.ien, mesh.ien.ctypes.data, mesh.ien.nbytes)
        L.Mesh3DUpdateDevice(self.mesh)
        L.Mesh3DSetBound(self.mesh, mesh.num_bound, mesh.bound_node_offset.ctypes.data, mesh.bound_node.ctypes.data,
                         mesh.bound_elem_offset.ctypes.data, mesh.bound_f2e.ctypes.data, mesh.bound_forn.ctypes.data)
        self.spy1x1 = L.CSRAttrCreate(self.mesh)
        self.spy1x3 = L.CSRAttrCreateBlock(self.spy1x1, 1, 3)
        self.spy3x1 = L.CSRAttrCreateBlock(self.spy1x1, 3, 1)
        self.spy3x3 = L.CSRAttrCreateBlock(self.spy1x1, 3, 3)
        offset = (C.c_int32 * 5)(0, 3, 4, 5, 6)
        self.J = L.MatrixCreateTypeFS(4, offset, None)
        fs = C.cast(self.J.contents.data, C.POINTER(MatrixFS)).contents
        fs.spy1x1 = self.spy1x1
        fs.mat[0] = L.MatrixCreateTypeCSR(self.spy3x3, None)
        fs.mat[1] = L.MatrixCreateTypeCSR(self.spy3x1, None)
        fs.mat[4] = L.MatrixCreateTypeCSR(self.spy1x3, None)
        fs.mat[5] = L.MatrixCreateTypeCSR(self.spy1x1, None)
        if reference_layout:
            L.MatrixFSUseReferenceLayout(self.J, 1)
        L.MatrixSetup(self.J)
        self.fs = fs
        self.nnz1 = int(self.spy1x1.contents.nnz)
        self.ksp = L.KrylovCreateGMRES(maxit, atol, rtol, None)
        L.KrylovSetVerbose(self.ksp, 0 if quiet else 1)
        L.KrylovSetMesh(self.ksp, self.mesh)
        if color:
            L.Mesh3DGenerateColorBatch(self.mesh)
        self.bcs = []
        for group, bctype in bcs:
            bc = L.DirichletCreate(self.mesh, group, 3)
            bt = C.cast(C.addressof(bc.contents) + C.sizeof(Dirichlet), C.POINTER(C.c_int))
            for i, t in enumerate(bctype):
                bt[i] = t
            self.bcs.append(bc)
        self.bc_arr = (C.POINTER(Dirichlet) * len(self.bcs))(*self.bcs)
        sync()

    # ---- integer structure back to numpy -------------------------------------------
    @property
    def num_color(self):
        return int(self.mesh.contents.num_color)

    def color(self):
        return d2h(self.mesh.contents.color, self.T, np.int32)

    def batch_offset(self):
        nc = self.num_color
        return np.ctypeslib.as_array(C.cast(self.mesh.contents.batch_offset, c_i32p), shape=(nc + 1,)).copy()

    def batch_ind(self):
        return d2h(self.mesh.contents.batch_ind, self.T, np.int32)

    def pattern(self, attr=None):
        a = (attr or self.spy1x1).contents
        return d2h(a.row_ptr, a.num_row + 1, np.int32), d2h(a.col_ind, a.nnz, np.int32)

    # ---- values ------------------------------------------------------------------------
    def block_values(self):
        return DeviceArray(self.nnz1 * 16, np.float64, ptr=lib().MatrixFSBlockValues(self.J))

    def export_values(self):
        """The four sub-matrix value arrays in the reference layout (A00, A01, A10, A11)."""
        lib().MatrixFSExportSubmatrices(self.J)   # no-op unless block mode
        sync()
        out = []
        for slot, mult in ((0, 9), (1, 3), (4, 3), (5, 1)):
            csr = C.cast(self.fs.mat[slot].contents.data, C.POINTER(MatrixCSR)).contents
            out.append(d2h(csr.val, self.nnz1 * mult, np.float64))
        return out

    # ---- the hot path --------------------------------------------------------------------
    def assemble_system(self, wg, dwg, F=None, want_J=False):
        lib().AssembleSystem(self.mesh, wg.ptr, dwg.ptr, F.ptr if F is not None else None, self.J if want_J else None,
                             C.cast(self.bc_arr, vp), len(self.bcs))

    def assemble_tet(self, wg, dwg, F=None, want_J=False):
        lib().AssembleSystemTet(self.mesh, wg.ptr, dwg.ptr, F.ptr if F is not None else None, self.J if want_J else None)

    def assemble_face(self, wg, dwg, F=None, want_J=False):
        lib().AssembleSystemTetFace(self.mesh, wg.ptr, dwg.ptr, F.ptr if F is not None else None, self.J if want_J else None)

    def matvec(self, x, y):
        lib().MatrixMatVec(self.J, x.ptr, y.ptr)

    def solve(self, x, b):
        lib().KrylovSolve(self.ksp, self.J, x.ptr, b.ptr)
        st = lib().KrylovGetStats(self.ksp).contents
        it = int(st.iterations)
        return it, float(st.rnrm_init), np.array(st.res_hist[:min(it, 512)]), bool(st.converged)

    def solve_flow_system(self, wgold, dwgold, dwg, F, dx, maxit=4):
        """SolveFlowSystem (src/main.c:77-283); returns (newton_its, rnorm[4], rnorm_init[4])."""
        rn, r0 = (C.c_double * 4)(), (C.c_double * 4)()
        it = lib().SolveFlowSystem(self.mesh, wgold.ptr, dwgold.ptr, dwg.ptr, self.J, F.ptr, dx.ptr, self.ksp,
                                   C.cast(self.bc_arr, vp), len(self.bcs), maxit, C.cast(rn, vp), C.cast(r0, vp))
        return int(it), np.array(rn[:]), np.array(r0[:])

    def time_step(self, wgold, dwgold, dwg, F, dx, newton_maxit=4, particles=None, dem_substeps=0):
        rn, r0 = (C.c_double * 4)(), (C.c_double * 4)()
        it = lib().DflTimeStep(self.mesh, wgold.ptr, dwgold.ptr, dwg.ptr, self.J, F.ptr, dx.ptr, self.ksp, C.cast(self.bc_arr, vp),
                               len(self.bcs), newton_maxit, particles.ctx if particles is not None else None, dem_substeps,
                               C.cast(rn, vp), C.cast(r0, vp))
        return int(it), np.array(rn[:]), np.array(r0[:])

    def close(self):
        L = lib()
        sync()
        for bc in self.bcs:
            L.DirichletDestroy(bc)
        self.bcs = []
        L.KrylovDestroy(self.ksp)
        L.MatrixDestroy(self.J)
        for a in (self.spy1x1, self.spy1x3, self.spy3x1, self.spy3x3):
            L.CSRAttrDestroy(a)
        L.Mesh3DDestroy(self.mesh)
