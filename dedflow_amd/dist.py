"""Element-partitioned multi-GPU layer: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on MI355X; "gloo" for CPU tests).

The reference is single-GPU, so this is new design, not a match:

* partition   RCB on tet centroids (C, DflPartitionRCB).  node owner = lowest part
              touching the node.  A rank's local mesh = every tet touching a node
              it owns (own tets + one halo layer), so owned matrix rows and owned
              RHS entries are complete after a purely local colored assembly --
              ZERO assembly communication.
* numbering   local nodes: owned first (ascending global id), then ghosts.  SpMV
              and the preconditioner run on the owned rows only; ghost entries of
              every Krylov vector stay zero, so local dot products are exact
              partial sums.
* collectives (1) halo exchange of the SpMV input (4 dof per interface node,
              point-to-point to the few neighbours that need it), (2) all-reduce
              of the k+1 CGS coefficients and of ||w||^2 per GMRES iteration,
              (3) one all-reduce for the initial residual.  All latency-bound.

The numpy/torch pieces here are plumbing (index sets, buffers, collectives); all
arithmetic on the path is in libdedflow.so.
"""
from __future__ import annotations

import ctypes as C
import os
import sys
from dataclasses import dataclass

import numpy as np

from .meshgen import TetMesh


@dataclass
class LocalMesh:
    mesh: TetMesh            # local connectivity / coordinates / boundary groups (local numbering)
    n_owned: int             # local nodes [0, n_owned) are owned
    l2g_node: np.ndarray     # local node -> global node
    l2g_elem: np.ndarray     # local tet  -> global tet
    ghost_owner: np.ndarray  # owner rank of each ghost node (local ids n_owned..)
    rank: int
    world: int
    n_interior: int = 0      # local nodes [0, n_interior) are owned AND have no ghost neighbour


def partition_rcb(mesh: TetMesh, num_part: int) -> np.ndarray:
    """epart[T] through the C partitioner in libdedflow.so."""
    from . import api
    L = api.lib()
    epart = np.empty(mesh.num_tet, np.int32)
    L.DflPartitionRCB.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    L.DflPartitionRCB.restype = None
    L.DflPartitionRCB(mesh.num_tet, mesh.ien.ctypes.data, mesh.xg.ctypes.data, num_part, epart.ctypes.data)
    return epart


def node_owner(mesh: TetMesh, epart: np.ndarray, num_part: int) -> np.ndarray:
    """owner(node) = min part over the tets touching it."""
    ien = mesh.ien.reshape(-1, 4)
    owner = np.full(mesh.num_node, num_part, np.int32)
    for p in range(num_part - 1, -1, -1):  # smaller parts overwrite larger ones
        owner[ien[epart == p].reshape(-1)] = p
    return owner


def build_local(mesh: TetMesh, epart: np.ndarray, owner: np.ndarray, rank: int, world: int) -> LocalMesh:
    ien = mesh.ien.reshape(-1, 4)
    touches = (owner[ien] == rank).any(axis=1)
    l2g_elem = np.nonzero(touches)[0].astype(np.int64)
    lien_g = ien[l2g_elem]
    nodes = np.unique(lien_g)
    own_mask = owner[nodes] == rank
    # owned nodes whose every neighbour is owned come first: their matrix rows read no ghost entry, so the
    # matvec of those rows can run while the halo exchange is in flight
    ghost_vertex = owner[lien_g] != rank
    near_ghost = np.zeros(mesh.num_node, bool)
    near_ghost[np.unique(lien_g[ghost_vertex.any(axis=1)])] = True
    interior = own_mask & ~near_ghost[nodes]
    boundary = own_mask & near_ghost[nodes]
    l2g_node = np.concatenate([nodes[interior], nodes[boundary], nodes[~own_mask]]).astype(np.int64)
    n_owned = int(own_mask.sum())
    n_interior = int(interior.sum())
    g2l = np.full(mesh.num_node, -1, np.int64)
    g2l[l2g_node] = np.arange(l2g_node.size)
    e_g2l = np.full(mesh.num_tet, -1, np.int64)
    e_g2l[l2g_elem] = np.arange(l2g_elem.size)
    lien = g2l[lien_g].astype(np.int32)
    xg = mesh.xg.reshape(-1, 3)[l2g_node]

    node_off, elem_off = [0], [0]
    bnodes, bf2e, bforn, bien = [], [], [], []
    for g in range(mesh.num_bound):
        gn = mesh.bound_node[mesh.bound_node_offset[g]:mesh.bound_node_offset[g + 1]]
        ln = g2l[gn]
        ln = np.sort(ln[ln >= 0]).astype(np.int32)  # every local node of the group (owned or ghost)
        lo, hi = mesh.bound_elem_offset[g], mesh.bound_elem_offset[g + 1]
        fe = e_g2l[mesh.bound_f2e[lo:hi]]
        keep = fe >= 0                               # faces whose parent tet is local
        bnodes.append(ln)
        bf2e.append(fe[keep].astype(np.int32))
        bforn.append(mesh.bound_forn[lo:hi][keep].astype(np.int32))
        bien.append(g2l[mesh.bound_ien[3 * lo:3 * hi].reshape(-1, 3)[keep]].reshape(-1).astype(np.int32))
        node_off.append(node_off[-1] + ln.size)
        elem_off.append(elem_off[-1] + int(keep.sum()))
    lm = TetMesh(M=mesh.M, xg=np.ascontiguousarray(xg.reshape(-1)), ien=np.ascontiguousarray(lien.reshape(-1)),
                 bound_node_offset=np.asarray(node_off, np.int32), bound_node=np.concatenate(bnodes).astype(np.int32),
                 bound_elem_offset=np.asarray(elem_off, np.int32), bound_ien=np.concatenate(bien).astype(np.int32),
                 bound_f2e=np.concatenate(bf2e).astype(np.int32), bound_forn=np.concatenate(bforn).astype(np.int32))
    return LocalMesh(lm, n_owned, l2g_node, l2g_elem, owner[l2g_node[n_owned:]].astype(np.int32), rank, world, n_interior)


def localize_vector(v: np.ndarray, lm: LocalMesh, N_global: int) -> np.ndarray:
    """global [u|p|phi|T] vector -> local layout"""
    g = lm.l2g_node
    n = g.size
    out = np.empty(6 * n)
    out[:3 * n] = v[:3 * N_global].reshape(-1, 3)[g].reshape(-1)
    for s in range(3):
        out[(3 + s) * n:(4 + s) * n] = v[(3 + s) * N_global:(4 + s) * N_global][g]
    return out


class HaloPlan:
    """Who sends which owned nodes to whom.  Built with one all_gather_object of the ghost
    lists (setup time); exchange = one batch of isend/irecv per neighbour."""

    def __init__(self, lm: LocalMesh, dist, device, use_host_staging: bool):
        import torch
        self.torch, self.dist, self.device, self.staged = torch, dist, device, use_host_staging
        self.rank, self.world = lm.rank, lm.world
        n = lm.l2g_node.size
        self.n_local, self.n_owned = n, lm.n_owned
        self.n_interior = lm.n_interior
        ghosts_g = lm.l2g_node[lm.n_owned:]
        need = {int(q): ghosts_g[lm.ghost_owner == q] for q in np.unique(lm.ghost_owner)}
        gathered = [None] * self.world
        dist.all_gather_object(gathered, need)
        g2l_owned = {}
        own_g = lm.l2g_node[:lm.n_owned]
        order = np.argsort(own_g)
        self.recv_idx, self.send_idx = {}, {}
        for q, arr in need.items():  # my ghosts owned by q, in the order I asked for them
            loc = np.nonzero(lm.ghost_owner == q)[0] + lm.n_owned
            self.recv_idx[q] = self._dof_index(loc, n)
        for q in range(self.world):
            if q == self.rank or gathered[q] is None or self.rank not in gathered[q]:
                continue
            want = gathered[q][self.rank]
            pos = order[np.searchsorted(own_g, want, sorter=order)]
            assert np.array_equal(own_g[pos], want)
            self.send_idx[q] = self._dof_index(pos, n)
        self.neighbours = sorted(set(self.recv_idx) | set(self.send_idx))
        self.bytes_per_exchange = 8 * sum(int(v.numel()) for v in self.send_idx.values())
        # one-collective form (NCCL/RCCL all_to_all_single): concatenated index lists in rank order
        self.send_splits = [int(self.send_idx[q].numel()) if q in self.send_idx else 0 for q in range(self.world)]
        self.recv_splits = [int(self.recv_idx[q].numel()) if q in self.recv_idx else 0 for q in range(self.world)]
        empty = torch.zeros(0, dtype=torch.int64, device=device)
        self.send_all = torch.cat([self.send_idx.get(q, empty) for q in range(self.world)]) if self.world else empty
        self.recv_all = torch.cat([self.recv_idx.get(q, empty) for q in range(self.world)]) if self.world else empty
        self.use_all_to_all = (not use_host_staging) and dist.get_backend() == "nccl"

    def _dof_index(self, nodes, n):
        """flat indices of the 4 (u0,u1,u2,p) dofs of each node in the local [u|p|...] layout"""
        nodes = np.asarray(nodes, np.int64)
        idx = np.concatenate([3 * nodes, 3 * nodes + 1, 3 * nodes + 2, 3 * n + nodes])
        return self.torch.as_tensor(idx, device=self.device)

    def exchange(self, x):
        """x: torch f64 tensor (local vector, on self.device); fills the ghost dofs in place."""
        torch, dist = self.torch, self.dist
        if self.use_all_to_all:
            # xGMI is a full mesh: one grouped send/recv launch to all neighbours at once
            send = x.index_select(0, self.send_all)
            recv = torch.empty(self.recv_all.numel(), dtype=x.dtype, device=self.device)
            dist.all_to_all_single(recv, send, self.recv_splits, self.send_splits)
            x.index_copy_(0, self.recv_all, recv)
            return
        ops, recv_bufs = [], {}
        for q in self.neighbours:
            if q in self.send_idx:
                buf = x.index_select(0, self.send_idx[q])
                if self.staged:
                    buf = buf.cpu()
                ops.append(dist.P2POp(dist.isend, buf, q))
            if q in self.recv_idx:
                rb = torch.empty(self.recv_idx[q].numel(), dtype=x.dtype, device="cpu" if self.staged else self.device)
                recv_bufs[q] = rb
                ops.append(dist.P2POp(dist.irecv, rb, q))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        for q, rb in recv_bufs.items():
            x.index_copy_(0, self.recv_idx[q], rb.to(self.device) if self.staged else rb)


class TorchDeviceAllocator:
    """Installs a torch-backed DEVICE allocator through the reference's own Allocator vtable
    (alloc.h:19-24) so that device buffers the C layer creates (Krylov work space, matrices)
    can be handed to torch.distributed as tensors.  Zero-filled like src/alloc.c:23-30."""

    def __init__(self, device):
        import torch
        from . import api
        self.torch, self.device = torch, device
        self.blocks = {}   # base ptr -> uint8 tensor
        self.bases = []
        self._views = {}   # (ptr, n) -> (base, f64 view): the GMRES loop asks for the same few buffers
        MALLOC = C.CFUNCTYPE(C.c_void_p, C.c_ssize_t, C.c_void_p)
        FREE = C.CFUNCTYPE(None, C.c_void_p, C.c_ssize_t, C.c_void_p)

        class Allocator(C.Structure):
            _fields_ = [("malloc", MALLOC), ("free", FREE), ("ctx", C.c_void_p)]

        def _malloc(size, ctx):
            if size <= 0:
                return None
            t = torch.zeros(int(size) + 16, dtype=torch.uint8, device=device)
            p = t.data_ptr()
            self.blocks[p] = t
            self.bases = sorted(self.blocks)
            return p

        def _free(ptr, size, ctx):
            if ptr and ptr in self.blocks:
                del self.blocks[ptr]
                self.bases = sorted(self.blocks)
                self._views.clear()

        self._m, self._f = MALLOC(_malloc), FREE(_free)
        L = api.lib()
        L.GetDefaultAllocator.restype = C.POINTER(Allocator)
        L.GetDefaultAllocator.argtypes = [C.c_int]
        a = L.GetDefaultAllocator(1).contents
        a.malloc, a.free = self._m, self._f

    def tensor(self, ptr, n, dtype=None):
        """f64 view of n elements at raw device pointer `ptr` (must lie in one of our blocks)."""
        import bisect
        torch = self.torch
        key = (ptr, n)
        hit = self._views.get(key)
        if hit is not None and hit[0] in self.blocks:
            return hit[1]
        i = bisect.bisect_right(self.bases, ptr) - 1
        base = self.bases[i]
        blk = self.blocks[base]
        off = ptr - base
        assert 0 <= off and off + 8 * n <= blk.numel(), "pointer outside torch-allocated device memory"
        v = blk[off:off + 8 * n].view(torch.float64)
        if len(self._views) > 4096:
            self._views.clear()
        self._views[key] = (base, v)
        return v


class RawPointerViews:
    """torch f64 views of raw device pointers (zero copy, through __cuda_array_interface__): lets the
    torch.distributed callbacks work on buffers the C layer allocated from its own device pool, so no torch-backed
    allocator has to be installed.  Same `.tensor()` / `.blocks` surface as TorchDeviceAllocator."""

    class _Raw:
        def __init__(self, ptr, n):
            self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (int(ptr), False), "version": 2}

    def __init__(self, device):
        import torch
        self.torch, self.device = torch, device
        self.blocks = {}   # keeps tensors created by dist_bench.device_vector alive
        self.bases = []
        self._views = {}

    def tensor(self, ptr, n, dtype=None):
        key = (ptr, n)
        v = self._views.get(key)
        if v is None:
            v = self.torch.as_tensor(self._Raw(ptr, n), device=self.device)
            assert v.data_ptr() == ptr and v.numel() == n
            if len(self._views) > 4096:
                self._views.clear()
            self._views[key] = v
        return v


class DistSolverComm:
    """DflComm callbacks (include/dedflow.h) implemented with torch.distributed."""

    def __init__(self, plan: HaloPlan, alloc: TorchDeviceAllocator, dist):
        from . import api
        self.plan, self.alloc, self.dist = plan, alloc, dist
        self.n_allreduce = 0
        self.n_halo = 0
        self.staged = plan.staged

        def _allreduce(ctx, ptr, n):
            t = self.alloc.tensor(ptr, n)
            if self.staged:
                h = t.cpu()
                dist.all_reduce(h)
                t.copy_(h)
            else:
                dist.all_reduce(t)
            self.n_allreduce += 1

        def _halo(ctx, ptr):
            t = self.alloc.tensor(ptr, 4 * plan.n_local)
            plan.exchange(t)
            self.n_halo += 1

        self._a, self._h = api.ALLREDUCE_FN(_allreduce), api.HALO_FN(_halo)
        # no split exchange on this path (synchronous callbacks); the interior / boundary row split is still used
        self.comm = api.DflComm(self._a, self._h, None, plan.n_owned, api.HALO_FN(), api.HALO_FN(), plan.n_interior,
                                plan.rank, plan.world, api.STREAM_FN())

    def install(self, ksp):
        from . import api
        api.lib().KrylovSetComm(ksp, C.byref(self.comm))


class RcclSolverComm:
    """The C-level RCCL communicator (host/comm_rccl.c) behind the same DflComm struct: collectives are enqueued
    from the C GMRES loop on the library stream, no Python in the iteration.  torch.distributed is only the
    bootstrap channel for the 128-byte unique id.  `verify()` checks it against the torch.distributed path."""

    def __init__(self, plan: HaloPlan, dist, device):
        import torch
        from . import api
        self.plan, self.dist, self.device, self.torch = plan, dist, device, torch
        L = self.L = api.lib()
        vp, i32 = C.c_void_p, C.c_int32
        L.DflRcclLoad.restype, L.DflRcclLoad.argtypes = C.c_int, [C.c_char_p]
        L.DflRcclUniqueIdBytes.restype, L.DflRcclUniqueIdBytes.argtypes = C.c_int, []
        L.DflRcclGetUniqueId.restype, L.DflRcclGetUniqueId.argtypes = C.c_int, [C.c_char_p]
        L.DflRcclCommCreate.restype, L.DflRcclCommCreate.argtypes = vp, [C.c_char_p, C.c_int, C.c_int]
        L.DflRcclCommSetHalo.restype, L.DflRcclCommSetHalo.argtypes = None, [vp, i32, i32, vp, vp, vp, vp]
        L.DflRcclCommVtable.restype, L.DflRcclCommVtable.argtypes = vp, [vp]
        L.DflRcclCommCounters.restype, L.DflRcclCommCounters.argtypes = None, [vp, vp, vp]
        L.DflRcclCommDestroy.restype, L.DflRcclCommDestroy.argtypes = None, [vp]
        # ncclCommInitRank is itself a rendezvous: a rank that skipped it would leave the others waiting inside it.  So every
        # local precondition (library loaded, ids made and received) is agreed on with a MIN all-reduce over torch.distributed
        # BEFORE anybody enters it, the outcome of each ncclCommInitRank is agreed on right after it, and all ranks take the
        # same branch: either every rank holds both communicators, or every rank raises here.
        def agree(ok_local):
            flag = torch.tensor([1.0 if ok_local else 0.0], dtype=torch.float64, device=device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            return bool(flag.item() > 0.5)

        bundled = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        path = bundled if os.path.exists(bundled) else ""     # the RCCL build torch already runs on
        ok, why = True, ""
        if L.DflRcclLoad(path.encode()) != 0:
            ok, why = False, "RCCL could not be loaded from %r" % path
        nbytes = L.DflRcclUniqueIdBytes()
        box = [b"", b""]     # two ids: all-reduce communicator, halo communicator (own stream)
        if plan.rank == 0 and ok:
            for k in range(2):
                buf = C.create_string_buffer(nbytes)
                if L.DflRcclGetUniqueId(buf) != 0:
                    ok, why = False, "ncclGetUniqueId failed"
                    box = [b"", b""]
                    break
                box[k] = bytes(buf.raw)
        dist.broadcast_object_list(box, src=0)
        torch.cuda.synchronize()
        self.c = None
        if ok and not (len(box[0]) == nbytes and len(box[1]) == nbytes):
            ok, why = False, "rank 0 could not create the RCCL unique ids"
        if not agree(ok):                                       # nobody has entered ncclCommInitRank yet
            raise RuntimeError(why or "RCCL is unavailable on another rank")
        self.c = L.DflRcclCommCreate(box[0], plan.rank, plan.world)
        if not agree(bool(self.c)):
            if self.c:
                L.DflRcclCommDestroy(self.c)
                self.c = None
            raise RuntimeError("ncclCommInitRank failed on this or another rank")   # raised on EVERY rank
        L.DflRcclCommCreateHaloComm.restype, L.DflRcclCommCreateHaloComm.argtypes = C.c_int, [vp, C.c_char_p]
        L.DflRcclCommDropHaloComm.restype, L.DflRcclCommDropHaloComm.argtypes = None, [vp]
        halo_ok = L.DflRcclCommCreateHaloComm(self.c, box[1]) == 0
        if not agree(halo_ok):
            # one rank sharing the main communicator for its halo traffic while its peers use a second one would pair
            # sends and receives of different communicators: every rank drops the second communicator
            L.DflRcclCommDropHaloComm(self.c)
        sc = np.asarray(plan.send_splits, np.int32)
        rc = np.asarray(plan.recv_splits, np.int32)
        si = np.ascontiguousarray(plan.send_all.cpu().numpy().astype(np.int32))
        ri = np.ascontiguousarray(plan.recv_all.cpu().numpy().astype(np.int32))
        self._keep = (sc, rc, si, ri)
        L.DflRcclCommSetHalo(self.c, plan.n_local, plan.n_owned, sc.ctypes.data, si.ctypes.data, rc.ctypes.data, ri.ctypes.data)
        L.DflRcclCommSetInterior.restype, L.DflRcclCommSetInterior.argtypes = None, [vp, i32]
        L.DflRcclCommSetInterior(self.c, plan.n_interior)
        self.vt = L.DflRcclCommVtable(self.c)
        self.staged = False

    def _counters(self):
        a, h = C.c_int64(0), C.c_int64(0)
        self.L.DflRcclCommCounters(self.c, C.byref(a), C.byref(h))
        return a.value, h.value

    n_allreduce = property(lambda self: self._counters()[0])
    n_halo = property(lambda self: self._counters()[1])

    def verify(self):
        """Halo exchange and all-reduce through the C path against torch.distributed on the same inputs.
        Returns True on every rank or False on every rank."""
        from . import api
        torch, dist, plan = self.torch, self.dist, self.plan
        vt = api.DflComm.from_address(self.vt)
        ok = True
        try:
            g = torch.Generator(device="cpu").manual_seed(1234 + plan.rank)
            x = torch.rand(6 * plan.n_local, dtype=torch.float64, generator=g).to(self.device)
            x1, x2 = x.clone(), x.clone()
            if plan.world > 1:
                plan.exchange(x1[:4 * plan.n_local])
            torch.cuda.synchronize()
            vt.halo_exchange(vt.ctx, x2.data_ptr())
            torch.cuda.synchronize()
            ok = ok and bool(torch.equal(x1, x2))
            x3 = x.clone()
            torch.cuda.synchronize()
            vt.halo_begin(vt.ctx, x3.data_ptr())      # split form: side stream + events
            vt.halo_end(vt.ctx, x3.data_ptr())
            torch.cuda.synchronize()
            ok = ok and bool(torch.equal(x1, x3))
            r1 = x[:7].clone()
            r2 = x[:7].clone()
            dist.all_reduce(r1)
            torch.cuda.synchronize()
            vt.allreduce_sum(vt.ctx, r2.data_ptr(), 7)
            torch.cuda.synchronize()
            ok = ok and bool(torch.allclose(r1, r2, rtol=1e-14, atol=0.0))
        except Exception as exc:  # noqa: BLE001 - any failure means "do not use this path"
            print("RcclSolverComm.verify: %r" % (exc,), file=sys.stderr)
            ok = False
        flag = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device=self.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return bool(flag.item() > 0.5)

    def install(self, ksp):
        from . import api
        self.L.KrylovSetComm(ksp, C.cast(self.vt, C.POINTER(api.DflComm)))
