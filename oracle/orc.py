"""ctypes loader for the CPU oracle (oracle/liboracle.so).

ORACLE -- TEST INFRASTRUCTURE ONLY.  Importable from tests/, from
``__graft_entry__.smoke()`` and from ``bench.py``'s cpu_baseline leg; never from
``dedflow_amd`` (the product path fails loudly without its HIP library instead
of falling back to this).  See oracle/oracle.cpp for the parity status
("parity unpinned": the reference has no runnable CPU path and no fixtures).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")


def build() -> str:
    subprocess.check_call(["make", "-s", "-C", _HERE])
    return os.path.join(_HERE, "liboracle.so")


def lib():
    global _LIB
    if _LIB is None:
        path = os.environ.get("DFL_ORACLE_LIB") or os.path.join(_HERE, "liboracle.so")  # DFL_ORACLE_LIB: the sanitizer build
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
        _LIB.orc_set_threads(1)  # serial restatement by default; bench.py's cpu_baseline also times all cores
    return _LIB


def set_threads(n):
    lib().orc_set_threads(int(n))


def num_procs():
    return int(lib().orc_num_procs())


def _p(a):
    """pointer-or-NULL for optional arrays"""
    if a is None:
        return None
    return a.ctypes.data_as(C.c_void_p)


BC_STRONG = 1

# BC table of the reference driver (src/main.c:454-476): (boundary group, bctype[3])
REFERENCE_BCS = [(0, (1, 1, 1)), (2, (0, 1, 0)), (3, (0, 0, 1)), (4, (0, 0, 0))]
FACE_GROUP = 4  # src/assemble.cu:1826-1828


def xorwow_legacy(n, seed=1234):
    out = np.empty(n, np.uint32)
    lib().orc_xorwow_legacy(C.c_ulonglong(seed), C.c_int(n), _p(out))
    return out


def xorwow_sequential(n, seed=1234):
    out = np.empty(n, np.uint32)
    lib().orc_xorwow_sequential(C.c_ulonglong(seed), C.c_int(n), _p(out))
    return out


def xorwow_substates(nsub=4096, seed=1234):
    out = np.empty(nsub * 6, np.uint32)
    lib().orc_xorwow_substates(C.c_ulonglong(seed), C.c_int(nsub), _p(out))
    return out.reshape(nsub, 6)


def priorities(T, seed=1234):
    raw = xorwow_legacy(T, seed)
    prio = np.empty(T, np.int32)
    lib().orc_priorities_from_u32(_p(raw), C.c_int(T), _p(prio))
    return prio


def v2e(ien, T, N):
    row = np.empty(N + 1, np.int32)
    col = np.empty(4 * T, np.int32)
    lib().orc_v2e(_p(ien), C.c_int(T), C.c_int(N), _p(row), _p(col))
    return row, col


def color_jpl(ien, T, N, prio=None, tie_break=False, max_color=256):
    if prio is None:
        prio = priorities(T)
    row, col = v2e(ien, T, N)
    color = np.ascontiguousarray(prio, dtype=np.int32).copy()
    ties = C.c_int(0)
    f = lib().orc_color_jpl
    f.restype = C.c_int
    nc = f(_p(ien), C.c_int(T), C.c_int(N), _p(row), _p(col), _p(color), C.c_int(max_color),
           C.c_int(1 if tie_break else 0), C.byref(ties))
    return color, int(nc), int(ties.value)


def color_jpl_sorted(ien, T, N, prio=None, tie_break=False):
    """The same coloring by one pass in descending priority order (orc_color_jpl_sorted): set-up of bench.py's 10M-tet
    cpu_baseline leg.  Returns (color, num_color)."""
    if prio is None:
        prio = priorities(T)
    prio = np.ascontiguousarray(prio, dtype=np.int32)
    row, col = v2e(ien, T, N)
    idx = np.arange(T, dtype=np.int32)
    order = np.ascontiguousarray(np.lexsort((idx, prio))[::-1].astype(np.int32))   # descending (priority, index)
    color = np.zeros(T, np.int32)
    f = lib().orc_color_jpl_sorted
    f.restype = C.c_int
    nc = f(_p(ien), C.c_int(T), _p(row), _p(col), _p(prio), _p(order), _p(color), C.c_int(1 if tie_break else 0))
    return color, int(nc)


def batches(color, num_color):
    T = color.size
    off = np.empty(num_color + 1, np.int32)
    ind = np.empty(T, np.int32)
    lib().orc_batches(_p(color), C.c_int(T), C.c_int(num_color), _p(off), _p(ind))
    return off, ind


def csr_pattern(ien, T, N):
    row = np.empty(N + 1, np.int32)
    col = np.empty(64 * N, np.int32)
    f = lib().orc_csr_pattern
    f.restype = C.c_int
    nnz = f(_p(ien), C.c_int(T), C.c_int(N), _p(row), _p(col))
    if nnz < 0:
        raise RuntimeError("CSRHashMapPush: row overflow (PREALLOC_SIZE 64)")
    return row, np.ascontiguousarray(col[:nnz])


def csr_expand(row, col, br, bc):
    N = row.size - 1
    nr = np.empty(N * br + 1, np.int32)
    nc = np.empty(col.size * br * bc, np.int32)
    lib().orc_csr_expand(_p(row), _p(col), C.c_int(N), C.c_int(br), C.c_int(bc), _p(nr), _p(nc))
    return nr, nc


class System:
    """The reference's setup sequence (src/main.c:372-413) on the CPU: nodal
    pattern, three expanded patterns, JPL coloring, color batches."""

    def __init__(self, mesh, tie_break=False, prio=None, sorted_coloring=False):
        self.mesh = mesh
        self.N = mesh.num_node
        self.T = mesh.num_tet
        self.rp11, self.ci11 = csr_pattern(mesh.ien, self.T, self.N)
        self.nnz1 = int(self.ci11.size)
        self.rp33, self.ci33 = csr_expand(self.rp11, self.ci11, 3, 3)
        self.rp31, self.ci31 = csr_expand(self.rp11, self.ci11, 3, 1)
        self.rp13, self.ci13 = csr_expand(self.rp11, self.ci11, 1, 3)
        if sorted_coloring:   # one pass instead of num_color rounds, same colors (bench.py's 10M-tet CPU leg)
            self.color, self.num_color = color_jpl_sorted(mesh.ien, self.T, self.N, prio=prio, tie_break=tie_break)
            self.num_ties = None
        else:
            self.color, self.num_color, self.num_ties = color_jpl(mesh.ien, self.T, self.N, prio=prio, tie_break=tie_break)
        self.batch_offset, self.batch_ind = batches(self.color, self.num_color)

    def new_values(self):
        n = self.nnz1
        return [np.zeros(9 * n), np.zeros(3 * n), np.zeros(3 * n), np.zeros(n)]

    def assemble_tet(self, wg, dwg, F=None, vals=None):
        m = self.mesh
        v = vals if vals is not None else [None] * 4
        lib().orc_assemble_tet(_p(m.xg), _p(m.ien), C.c_int(self.N), C.c_int(self.num_color), _p(self.batch_offset),
                               _p(self.batch_ind), _p(wg), _p(dwg), _p(F), _p(self.rp11), _p(self.ci11),
                               _p(v[0]), _p(v[1]), _p(v[2]), _p(v[3]))

    def assemble_face(self, wg, dwg, F=None, vals=None, group=FACE_GROUP):
        m = self.mesh
        lo, hi = int(m.bound_elem_offset[group]), int(m.bound_elem_offset[group + 1])
        f2e = np.ascontiguousarray(m.bound_f2e[lo:hi])
        forn = np.ascontiguousarray(m.bound_forn[lo:hi])
        v = vals if vals is not None else [None] * 4
        lib().orc_assemble_face(_p(m.xg), _p(m.ien), C.c_int(self.N), C.c_int(hi - lo), _p(f2e), _p(forn), _p(self.color),
                                C.c_int(self.num_color), _p(wg), _p(dwg), _p(F), _p(self.rp11), _p(self.ci11),
                                _p(v[0]), _p(v[1]), _p(v[2]), _p(v[3]))

    def bnodes(self, group):
        m = self.mesh
        return np.ascontiguousarray(m.bound_node[m.bound_node_offset[group]: m.bound_node_offset[group + 1]])

    def assemble_system(self, wg, dwg, want_F=False, want_J=False, bcs=REFERENCE_BCS):
        """AssembleSystem (src/main.c:31-75): zero, tet, face, zero F[4N:6N], BCs."""
        F = np.zeros(6 * self.N) if want_F else None
        vals = self.new_values() if want_J else None
        self.assemble_tet(wg, dwg, F, vals)
        self.assemble_face(wg, dwg, F, vals)
        if F is not None:
            F[4 * self.N:] = 0.0
        for group, bctype in bcs:
            bt = np.asarray(bctype, np.int32)
            bn = self.bnodes(group)
            if F is not None:
                lib().orc_dirichlet_vec(_p(F), C.c_int(bn.size), _p(bn), C.c_int(3), _p(bt))
            if vals is not None:
                lib().orc_dirichlet_mat(C.c_int(bn.size), _p(bn), C.c_int(3), _p(bt), C.c_int(self.N), _p(self.rp33),
                                        _p(self.ci33), _p(vals[0]), _p(self.rp31), _p(self.ci31), _p(vals[1]))
        return F, vals

    def _pat(self):
        return (_p(self.rp33), _p(self.ci33), _p(self.rp31), _p(self.ci31), _p(self.rp13), _p(self.ci13), _p(self.rp11),
                _p(self.ci11))

    def amvpby(self, vals, alpha, x, beta, y):
        lib().orc_fs_amvpby(C.c_int(self.N), *self._pat(), _p(vals[0]), _p(vals[1]), _p(vals[2]), _p(vals[3]),
                            C.c_double(alpha), _p(x), C.c_double(beta), _p(y))

    def matvec(self, vals, x):
        y = np.zeros(6 * self.N)
        self.amvpby(vals, 1.0, x, 0.0, y)
        return y

    def pc_setup(self, vals):
        d33 = np.empty(9 * self.N)
        d1 = np.empty(self.N)
        lib().orc_pc_setup(C.c_int(self.N), _p(self.rp11), _p(self.ci11), _p(vals[0]), _p(vals[3]), _p(d33), _p(d1))
        return d33, d1

    def pc_apply(self, d33, d1, x):
        y = np.empty(6 * self.N)
        lib().orc_pc_apply(C.c_int(self.N), _p(d33), _p(d1), _p(x), _p(y))
        return y

    def gmres(self, vals, b, x0=None, maxit=120, atol=1e-12, rtol=1e-4, pc=True):
        x = np.zeros(6 * self.N) if x0 is None else x0.copy()
        hist = np.zeros(maxit)
        r0 = C.c_double(0.0)
        f = lib().orc_gmres
        f.restype = C.c_int
        it = f(C.c_int(self.N), *self._pat(), _p(vals[0]), _p(vals[1]), _p(vals[2]), _p(vals[3]), C.c_int(1 if pc else 0),
               _p(x), _p(b), C.c_int(maxit), C.c_double(atol), C.c_double(rtol), _p(hist), C.byref(r0))
        return x, hist[:it].copy(), float(r0.value), int(it)

    def gmres_restarted(self, vals, b, restart, maxit, pc=True):
        """GMRES(restart) as a sequence of full-GMRES cycles of the restatement above, each started from the current
        iterate (build-defined feature: KrylovSetRestart); fixed work (atol = rtol = 0), `maxit` iterations in total."""
        x = np.zeros(6 * self.N)
        hist, total, r0 = [], 0, None
        while total < maxit:
            x, h, r, it = self.gmres(vals, b, x0=x, maxit=min(restart, maxit - total), atol=0.0, rtol=0.0, pc=pc)
            r0 = r if r0 is None else r0
            hist.extend(h.tolist())
            total += it
        return x, np.array(hist), r0, total

    def to_scipy(self, vals):
        """4N x 4N scipy CSR of the assembled system in the global [u|p] ordering."""
        import scipy.sparse as sp
        N = self.N
        A00 = sp.csr_matrix((vals[0], self.ci33, self.rp33), shape=(3 * N, 3 * N))
        A01 = sp.csr_matrix((vals[1], self.ci31, self.rp31), shape=(3 * N, N))
        A10 = sp.csr_matrix((vals[2], self.ci13, self.rp13), shape=(N, 3 * N))
        A11 = sp.csr_matrix((vals[3], self.ci11, self.rp11), shape=(N, N))
        return sp.bmat([[A00, A01], [A10, A11]], format="csr")


def elem_geometry(xg, nodes):
    invJ = np.empty(9); detJ = C.c_double(0.0); shg = np.empty(12); G = np.empty(9)
    lib().orc_elem_geometry(_p(xg), _p(np.asarray(nodes, np.int32)), _p(invJ), C.byref(detJ), _p(shg), _p(G))
    return invJ, float(detJ.value), shg, G


def elem_tensors(xg, nodes, N, wg, dwg):
    eF = np.empty(24); eJ = np.empty(576); qw = np.empty(24); qd = np.empty(24); qg = np.empty(18)
    lib().orc_elem_tensors(_p(xg), _p(np.asarray(nodes, np.int32)), C.c_int(N), _p(wg), _p(dwg), _p(eF), _p(eJ), _p(qw),
                           _p(qd), _p(qg))
    return eF, eJ, qw, qd, qg


def elem_heat(xg, nodes):
    J = np.empty(16)
    lib().orc_elem_heat(_p(xg), _p(np.asarray(nodes, np.int32)), _p(J))
    return J.reshape(4, 4)


def dem_forces(x, v, R, mass=1.0, kn=1.0e4, gn=1.0, brute=False):
    """Build-defined DEM contact sweep (oracle_ext.cpp); returns (acc[3P], tested pair count)."""
    P = x.size // 3
    acc = np.empty(3 * P)
    tested = C.c_longlong(0)
    lib().orc_dem_forces(C.c_int(P), _p(np.ascontiguousarray(x)), _p(np.ascontiguousarray(v)), C.c_double(R), C.c_double(mass),
                         C.c_double(kn), C.c_double(gn), C.c_int(1 if brute else 0), _p(acc), C.byref(tested))
    return acc, int(tested.value)
