"""Synthetic Kuhn-cube tet meshes and fields.

The reference ships no meshes (``box.h5`` / ``tet.h5`` are referenced by
``src/main.c:358-360`` but absent), so every input here is synthetic.  The
arrays follow the on-disk schema the reference reads (``src/MeshData.c:57-109``,
``src/Mesh.c:12-59``; written by ``tools/mesh_convert.py:116-126``):

  xg      f64[3N]   node coordinates, AoS
  ien     i32[4T]   tet connectivity, positive orientation
  bound   6 groups in the order x-,x+,y-,y+,z-,z+ with
          node_offset/node (sorted unique), elem_offset/ien(3 per face)/
          f2e (parent tet)/forn (local index of the tet vertex opposite the face,
          ``tools/mesh_convert.py:60-66``)
"""
from __future__ import annotations

import itertools
from dataclasses import dataclass

import numpy as np

_PERMS = list(itertools.permutations(range(3)))


def _perm_is_odd(p) -> bool:
    inv = sum(1 for a in range(3) for b in range(a + 1, 3) if p[a] > p[b])
    return inv % 2 == 1


@dataclass
class TetMesh:
    M: int
    xg: np.ndarray  # f64 [3N]
    ien: np.ndarray  # i32 [4T]
    bound_node_offset: np.ndarray  # i32 [nb+1]
    bound_node: np.ndarray  # i32
    bound_elem_offset: np.ndarray  # i32 [nb+1]
    bound_ien: np.ndarray  # i32 [3*nf]
    bound_f2e: np.ndarray  # i32 [nf]
    bound_forn: np.ndarray  # i32 [nf]

    @property
    def num_node(self) -> int:
        return self.xg.size // 3

    @property
    def num_tet(self) -> int:
        return self.ien.size // 4

    @property
    def num_bound(self) -> int:
        return self.bound_node_offset.size - 1


def kuhn_cube(M: int, jitter: float = 0.0, seed: int = 20241016) -> TetMesh:
    """M^3 cells x 6 tets on [0,1]^3; node id = i + (M+1)(j + (M+1)k)."""
    n1 = M + 1
    N = n1 ** 3
    ids = np.arange(N, dtype=np.int64)
    ii = ids % n1
    jj = (ids // n1) % n1
    kk = ids // (n1 * n1)
    xg = np.stack([ii, jj, kk], axis=1).astype(np.float64) / M
    if jitter > 0.0:
        rng = np.random.default_rng(seed)
        d = rng.uniform(-jitter / M, jitter / M, size=(N, 3))
        interior = (ii > 0) & (ii < M) & (jj > 0) & (jj < M) & (kk > 0) & (kk < M)
        xg[interior] += d[interior]

    c = np.arange(M ** 3, dtype=np.int64)
    ci = c % M
    cj = (c // M) % M
    ck = c // (M * M)
    base = ci + n1 * (cj + n1 * ck)
    strides = (1, n1, n1 * n1)
    ien = np.empty((M ** 3, 6, 4), dtype=np.int32)
    for t, p in enumerate(_PERMS):
        v0 = base
        v1 = v0 + strides[p[0]]
        v2 = v1 + strides[p[1]]
        v3 = v2 + strides[p[2]]
        if _perm_is_odd(p):
            v1, v2 = v2, v1  # restore positive orientation
        ien[:, t, 0] = v0
        ien[:, t, 1] = v1
        ien[:, t, 2] = v2
        ien[:, t, 3] = v3
    ien = ien.reshape(-1, 4)

    node_off = [0]
    elem_off = [0]
    nodes, bien, f2e, forn = [], [], [], []
    ccoord = (ci, cj, ck)
    for ax in range(3):
        for side in (0, M):
            # only the cell layer touching the plane can hold boundary faces
            layer = np.nonzero(ccoord[ax] == (0 if side == 0 else M - 1))[0]
            cand = (layer[:, None] * 6 + np.arange(6)[None, :]).reshape(-1)
            sub = ien[cand].astype(np.int64)
            on = ((sub // strides[ax]) % n1) == side
            is_face = on.sum(axis=1) == 3
            tets = cand[is_face]
            on = on[is_face]
            opp = np.argmin(on, axis=1)  # the single False entry
            face_nodes = ien[tets][on].reshape(-1, 3)
            nodes.append(np.unique(face_nodes))
            bien.append(face_nodes.reshape(-1))
            f2e.append(tets.astype(np.int32))
            forn.append(opp.astype(np.int32))
            node_off.append(node_off[-1] + nodes[-1].size)
            elem_off.append(elem_off[-1] + tets.size)
    return TetMesh(
        M=M,
        xg=np.ascontiguousarray(xg.reshape(-1)),
        ien=np.ascontiguousarray(ien.reshape(-1)),
        bound_node_offset=np.asarray(node_off, dtype=np.int32),
        bound_node=np.concatenate(nodes).astype(np.int32),
        bound_elem_offset=np.asarray(elem_off, dtype=np.int32),
        bound_ien=np.concatenate(bien).astype(np.int32),
        bound_f2e=np.concatenate(f2e).astype(np.int32),
        bound_forn=np.concatenate(forn).astype(np.int32),
    )


def single_tet() -> TetMesh:
    """Reference-tet mesh with all four faces as boundary groups (cf. the
    DBG_TET configuration, ``src/main.c:357-361``)."""
    xg = np.array([0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1], dtype=np.float64)
    ien = np.array([0, 1, 2, 3], dtype=np.int32)
    node_off, elem_off, nodes, bien, f2e, forn = [0], [0], [], [], [], []
    for opp in range(4):
        fn = np.array([v for v in range(4) if v != opp], dtype=np.int32)
        nodes.append(fn)
        bien.append(fn)
        f2e.append(np.array([0], dtype=np.int32))
        forn.append(np.array([opp], dtype=np.int32))
        node_off.append(node_off[-1] + 3)
        elem_off.append(elem_off[-1] + 1)
    return TetMesh(1, xg, ien, np.asarray(node_off, np.int32), np.concatenate(nodes),
                   np.asarray(elem_off, np.int32), np.concatenate(bien),
                   np.concatenate(f2e), np.concatenate(forn))


def synthetic_fields(mesh: TetMesh, seed_dw: int = 7):
    """wgalpha / dwgalpha of length 6N in the layout [u: Nx3 AoS | p | phi | T]
    (``src/main.c:108-118``).  u = e_x + 0.1 sin(2 pi x_perm), p-slot of wgalpha
    zero (``src/main.c:118``), phi = x, T = -x (cf. ``src/main.c:301-319``);
    dwgalpha = 0.1 * uniform[-1,1] including the pressure slot."""
    N = mesh.num_node
    x = mesh.xg.reshape(-1, 3)
    wg = np.zeros(6 * N)
    u = np.zeros((N, 3))
    u[:, 0] = 1.0 + 0.1 * np.sin(2 * np.pi * x[:, 1])
    u[:, 1] = 0.1 * np.sin(2 * np.pi * x[:, 2])
    u[:, 2] = 0.1 * np.sin(2 * np.pi * x[:, 0])
    wg[: 3 * N] = u.reshape(-1)
    wg[4 * N: 5 * N] = x[:, 0]
    wg[5 * N: 6 * N] = -x[:, 0]
    rng = np.random.default_rng(seed_dw)
    dwg = 0.1 * rng.uniform(-1.0, 1.0, size=6 * N)
    return wg, dwg


def dem_particles(P: int, radius: float, seed_x: int = 11, seed_v: int = 12):
    """DEM inputs of SURVEY.md section 8(d): centres uniform in the unit cube,
    velocities N(0, 0.1^2)."""
    x = np.random.default_rng(seed_x).uniform(0.0, 1.0, size=(P, 3))
    v = np.random.default_rng(seed_v).normal(0.0, 0.1, size=(P, 3))
    return np.ascontiguousarray(x.reshape(-1)), np.ascontiguousarray(v.reshape(-1)), radius


def fan_mesh(num_surface: int = 40, seed: int = 5) -> TetMesh:
    """A ball of tets around ONE central vertex: `num_surface` points on the unit sphere, their convex hull triangulated
    (2 V - 4 triangles), every triangle joined to the centre.  The central vertex is shared by 2 V - 4 tets -- V = 40 gives
    76, more than the 64 conflict-free classes of the compact assembly schedule, with a nodal row of V + 1 = 41 nonzeros
    (inside the reference's 64 per row, csr.c:10) and 76 JPL colors (inside its 256).  Boundary group 0 = the whole surface,
    groups 1-5 empty."""
    from scipy.spatial import ConvexHull
    rng = np.random.default_rng(seed)
    p = rng.normal(size=(num_surface, 3))
    p /= np.linalg.norm(p, axis=1)[:, None]
    p *= rng.uniform(0.8, 1.0, size=(num_surface, 1))   # not all on one sphere: breaks symmetry, still star-shaped
    hull = ConvexHull(p / np.linalg.norm(p, axis=1)[:, None])
    tri = hull.simplices.astype(np.int64)
    xg = np.vstack([np.zeros((1, 3)), p])               # node 0 = the centre
    tets = np.column_stack([np.zeros(len(tri), np.int64), tri + 1])
    x = xg[tets]
    vol = np.einsum("ij,ij->i", np.cross(x[:, 1] - x[:, 0], x[:, 2] - x[:, 0]), x[:, 3] - x[:, 0])
    flip = vol < 0
    tets[flip, 2], tets[flip, 3] = tets[flip, 3].copy(), tets[flip, 2].copy()
    nf = len(tets)
    surf = np.arange(1, num_surface + 1, dtype=np.int32)
    return TetMesh(M=0, xg=np.ascontiguousarray(xg.reshape(-1)), ien=np.ascontiguousarray(tets.reshape(-1).astype(np.int32)),
                   bound_node_offset=np.array([0] + [num_surface] * 6, np.int32), bound_node=surf,
                   bound_elem_offset=np.array([0] + [nf] * 6, np.int32),
                   bound_ien=np.ascontiguousarray(tets[:, 1:].reshape(-1).astype(np.int32)),
                   bound_f2e=np.arange(nf, dtype=np.int32), bound_forn=np.zeros(nf, np.int32))
