"""One-process-per-GPU partitioned assembly and solve (torch.distributed: RCCL on GPUs, gloo in
the CPU tests).  Nothing like this exists in the reference (it is single-process numpy); the
partition follows BASELINE.json north_star: contiguous ELEMENT blocks per rank.

Layout
  * rank r owns elements [r*E/W, (r+1)*E/W).  Its local mesh is those elements over the nodes
    they touch, renumbered in ascending global order (so local rows/columns keep the global
    relative order).  Assembly of the local K needs no communication; rows of interface nodes
    hold partial sums.
  * vectors (x, r, z, p, s, b) are CONSISTENT: an interface dof carries the full value on
    every rank that has it.  Products w = K z are completed by summing interface rows over ranks.
  * all interface dofs of the job form one packed "interface vector" (G entries).  Each CG
    iteration does ONE all-reduce of G+3 doubles: interface partial sums of w plus the three
    scalars gamma = r.z, delta = z.w, r.r (Chronopoulos-Gear single-reduction PCG), so the
    iteration has a single synchronisation point and no host round trip.  delta uses
    z.(K z) = sum_ranks z_loc.(K_loc z_loc), so it is taken from the un-summed local product;
    gamma and r.r count every dof once through the ownership mask (lowest rank owns).
  * xGMI is point-to-point: for strip partitions (structured meshes) G is tiny (one node row per
    cut) and the all-reduce is latency-bound -- see DESIGN.md for the numbers and the next step
    (neighbour P2P + overlap with the interior SpMV).

`LocalOps` is what the driver needs from a rank-local backend.  The product backend is
HipLocalOps (C ABI -> HIP kernels).  tests/ provides a numpy stand-in so the partition logic and
the iteration can be exercised under gloo with world_size 2 on a CPU-only host.
"""
import ctypes as C

import numpy as np


# ------------------------------------------------------------------------------------------
# partition
# ------------------------------------------------------------------------------------------
def centroid_element_order(X, conn):
    """Permutation that sorts elements along a Morton (Z-order) curve of their centroids.  Element-index blocks
    are compact strips on ProblemCreator meshes; on unstructured or renumbered meshes (arm.inp, C4) contiguous
    index ranges can be scattered all over the domain, and sorting by centroid first restores compact
    blocks with short interfaces.  Stable, so a mesh that is already ordered keeps its order inside a cell."""
    X = np.asarray(X, dtype=float)
    c = X[np.asarray(conn)].mean(axis=1)
    lo, hi = c.min(axis=0), c.max(axis=0)
    q = ((c - lo) / np.where(hi > lo, hi - lo, 1.0) * 1023.0).astype(np.uint64)  # 10 bits per axis
    code = np.zeros(c.shape[0], dtype=np.uint64)
    nd = c.shape[1]
    for bit in range(10):
        for ax in range(nd):
            code |= ((q[:, ax] >> np.uint64(bit)) & np.uint64(1)) << np.uint64(bit * nd + ax)
    return np.argsort(code, kind="stable")


class ElementBlockPartition:
    """Element-block partition of (conn, nnodes) for `world` ranks, seen from `rank`.
    elem_perm: optional element permutation applied before cutting the blocks (centroid_element_order)."""

    def __init__(self, conn, nnodes, world, rank, ndof_per_node, elem_perm=None):
        conn = np.asarray(conn)
        if elem_perm is not None:
            conn = conn[np.asarray(elem_perm)]
        self.elem_perm = elem_perm
        E = conn.shape[0]
        if E < world:
            raise ValueError("fewer elements than ranks")
        self.world, self.rank, self.d = world, rank, ndof_per_node
        bounds = [(r * E) // world for r in range(world + 1)]
        self.elem_range = (bounds[rank], bounds[rank + 1])
        count = np.zeros(nnodes, dtype=np.int32)
        owner = np.full(nnodes, world, dtype=np.int32)
        mine = None
        for r in range(world):
            nodes_r = np.unique(conn[bounds[r]:bounds[r + 1]])
            count[nodes_r] += 1
            owner[nodes_r] = np.minimum(owner[nodes_r], r)
            if r == rank:
                mine = nodes_r
        self.nodes = mine  # ascending global node ids of this rank
        self.nnodes_local = mine.size
        self.conn_local = np.searchsorted(mine, conn[bounds[rank]:bounds[rank + 1]]).astype(np.int32)
        iface_nodes = np.flatnonzero(count > 1)  # global interface nodes, ascending
        self.n_iface_dofs = iface_nodes.size * ndof_per_node  # G
        slot_of = np.full(nnodes, -1, dtype=np.int64)
        slot_of[iface_nodes] = np.arange(iface_nodes.size)
        shared_local = np.flatnonzero(count[mine] > 1)  # local node ids on an interface
        d = ndof_per_node
        k = np.arange(d)
        self.iface_local_dofs = (shared_local[:, None] * d + k).ravel().astype(np.int32)
        self.iface_slots = (slot_of[mine[shared_local]][:, None] * d + k).ravel().astype(np.int32)
        self.owned_mask_nodes = owner[mine] == rank
        self.own_weight = np.repeat(self.owned_mask_nodes.astype(np.float64), d)  # per local dof
        self.multiplicity = np.repeat(count[mine].astype(np.float64), d)

    def local_dofs_of(self, global_dofs):
        """Map global dof ids to local ones; returns (local_ids, mask_present)."""
        g = np.asarray(global_dofs, dtype=np.int64)
        node, k = g // self.d, g % self.d
        pos = np.searchsorted(self.nodes, node)
        pos = np.minimum(pos, self.nodes.size - 1)
        present = self.nodes[pos] == node
        return (pos[present] * self.d + k[present]).astype(np.int64), present

    def global_dofs(self):
        return (self.nodes[:, None] * self.d + np.arange(self.d)).ravel()


# ------------------------------------------------------------------------------------------
# rank-local backend on the HIP library
# ------------------------------------------------------------------------------------------
class HipLocalOps:
    """Rank-local kernels through the C ABI.  Vectors are torch.float64 tensors on the rank's GPU."""

    def __init__(self, hctx):
        from . import _lib

        self.h = hctx
        self.lib = hctx.lib
        self._lib = _lib
        self.torch = hctx.torch
        self.device = hctx.device
        self.n = hctx.nrows

    def _p(self, t):
        return C.c_void_p(0) if t is None else C.c_void_p(t.data_ptr())

    def begin(self):
        """Bind the context to torch's current stream once per solve (the per-call rebinding of the
        generic HipContext methods costs more host time than the kernels take at 8 GPUs)."""
        self.h._sync_stream()

    def _call(self, name, *a):
        self._lib.check(self.h._ctx, getattr(self.lib, name)(self.h._ctx, *a), name)

    def zeros(self, n, dtype=None):
        return self.torch.zeros(int(n), dtype=dtype or self.torch.float64, device=self.device)

    def from_numpy(self, a, dtype=None):
        return self.h.to_device(a, dtype)

    def diag(self, K):
        return self.h.diag(K)

    def spmv_dot(self, K, x, y):
        self._call("pf_spmv_dot", self._p(K), self._p(x), self._p(y))

    def copy_indexed(self, n, src_idx, dst_idx, src, dst):
        self._call("pf_copy_indexed", C.c_int64(n), self._p(src_idx), self._p(dst_idx), self._p(src), self._p(dst))

    def dot(self, a, b, w, out):
        self._call("pf_dot", C.c_int64(a.numel()), self._p(a), self._p(b), self._p(w), self._p(out))

    def cgcg_update_xrz(self, state, p, s, dinv, w, x, r, z):
        self._call("pf_cgcg_update_xrz", self._p(state), self._p(p), self._p(s), self._p(dinv), self._p(w),
                   self._p(x), self._p(r), self._p(z))

    def cgcg_reduce(self, red, zero, nzero):
        self._call("pf_cgcg_reduce", self._p(red), self._p(zero), C.c_int64(nzero))

    def spmv(self, K, x, y):
        self._call("pf_spmv", self._p(K), self._p(x), self._p(y))

    def amg_prepare(self, K, reuse=False):
        self._call("pf_amg_prepare", self._p(K), int(bool(reuse)))

    def add_to_diag(self, K, add):
        self._call("pf_add_to_diag", self._p(K), self._p(add))

    def amg_vcycle(self, r, z):
        self._call("pf_amg_vcycle", self._p(r), self._p(z))

    def cgcg_update_ps(self, state_in, state_out, red, first, z, w, p, s):
        self._call("pf_cgcg_update_ps", self._p(state_in), self._p(state_out), self._p(red), int(first),
                   self._p(z), self._p(w), self._p(p), self._p(s))


# ------------------------------------------------------------------------------------------
# distributed pieces
# ------------------------------------------------------------------------------------------
def _all_reduce(t, group):
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)


class Interface:
    """Packed interface vector of the job + this rank's (local dof, slot) lists on the device."""

    def __init__(self, ops, part, group=None):
        torch = ops.torch
        self.ops, self.group = ops, group
        self.G = part.n_iface_dofs
        self.n = part.iface_local_dofs.size
        self.local = ops.from_numpy(part.iface_local_dofs, torch.int32)
        self.slots = ops.from_numpy(part.iface_slots, torch.int32)
        self.buf = ops.zeros(self.G + 3)  # [interface partial sums | gamma, delta, rr]

    def sum(self, v):
        """v[interface dofs] <- sum over ranks (in place); other entries untouched."""
        if hasattr(self.ops, "begin"):
            self.ops.begin()
        self.buf.zero_()
        self.ops.copy_indexed(self.n, self.local, self.slots, v, self.buf)
        _all_reduce(self.buf, self.group)
        self.ops.copy_indexed(self.n, self.slots, self.local, self.buf, v)
        return v


def dist_cg(ops, K, b, dinv, own_w, iface, rtol=1e-8, atol=0.0, maxiter=None, check_every=50):
    """Chronopoulos-Gear single-reduction Jacobi-PCG over ranks, x0 = 0.
    Stops when ||r||_2 <= max(rtol ||b||_2, atol)  (the reference's convention, pyfem.py:2418).
    Returns (x_local, iters, relres); raises RuntimeError("cg failed with code <iters>") at maxiter
    (pyfem.py:2421-2422)."""
    n = ops.n
    G = iface.G
    buf = iface.buf
    red = buf[G:]  # gamma, delta, rr
    if hasattr(ops, "begin"):
        ops.begin()
    x, r, z, w, p, s = ops.zeros(n), b.clone(), ops.zeros(n), ops.zeros(n), ops.zeros(n), ops.zeros(n)
    state, state_next = ops.zeros(3), ops.zeros(3)  # [alpha, beta, gamma], ping-pong
    bb = ops.zeros(1)
    ops.dot(b, b, own_w, bb)
    _all_reduce(bb, iface.group)
    bnorm = float(bb.item()) ** 0.5
    tol = max(rtol * bnorm, atol)
    if maxiter is None:
        maxiter = 10 * n
    it, rnorm = 0, float("inf")
    while it < maxiter:
        ops.cgcg_update_xrz(state, p, s, dinv, own_w, x, r, z)
        ops.spmv_dot(K, z, w)
        ops.cgcg_reduce(red, buf, G)  # clear the interface part; local [gamma, delta, rr] into the tail
        ops.copy_indexed(iface.n, iface.local, iface.slots, w, buf)
        _all_reduce(buf, iface.group)
        ops.copy_indexed(iface.n, iface.slots, iface.local, buf, w)
        # r.r in `red` belongs to the residual after `it` updates
        if it % check_every == 0 or it + 1 == maxiter:
            rnorm = float(red[2].item()) ** 0.5
            if not rnorm == rnorm or rnorm <= tol:
                break
        ops.cgcg_update_ps(state, state_next, red, it == 0, z, w, p, s)
        state, state_next = state_next, state
        it += 1
    relres = rnorm / bnorm if bnorm > 0 else rnorm
    if not rnorm <= tol:
        raise RuntimeError(f"cg failed with code {it}")
    return x, it, relres


class BinCoarseSpace:
    """Global coarse space for the Schwarz preconditioner: the rigid-body modes (two translations and the rotation
    about the bin centre) of the nodes in each cell of a uniform bx x by grid over the mesh's bounding box.

    The bin of a node follows from its coordinates alone, so every rank builds the same basis Z for the nodes it
    shares with others without any communication, and Z^T applied to the un-summed local product K_loc p adds up
    over ranks to Z^T (K p).  E = Z^T K Z (3 bx by square, dense) is found with 27 local products: bins of one of
    the 3 x 3 colours are three cells apart and K couples adjacent cells only (cells are at least two elements wide),
    so one product per colour and mode separates all their columns.  E is summed over ranks once per matrix and
    pseudo-inverted redundantly; applying Z E^+ Z^T then needs no communication of its own (Z^T r is updated from
    Z^T (K p), which rides along with the interface sums)."""

    def __init__(self, ops, X_local, free_mask, bbox, nelems_global, target_bins=512):
        torch = ops.torch
        self.ops = ops
        (x0, y0), (x1, y1) = bbox
        lx, ly = max(x1 - x0, 1e-300), max(y1 - y0, 1e-300)
        h = (lx * ly / max(nelems_global, 1)) ** 0.5  # typical element size
        bx = int(max(1, min(round((target_bins * lx / ly) ** 0.5), lx / (2.5 * h))))
        by = int(max(1, min(round(target_bins / max(bx, 1)), ly / (2.5 * h))))
        self.bx, self.by, self.nb = bx, by, bx * by
        Xl = np.asarray(X_local, dtype=float)
        ix = np.clip(((Xl[:, 0] - x0) / lx * bx).astype(np.int64), 0, bx - 1)
        iy = np.clip(((Xl[:, 1] - y0) / ly * by).astype(np.int64), 0, by - 1)
        self.bin = ops.from_numpy(ix + bx * iy, torch.int64)
        self.colour = ops.from_numpy((ix % 3) + 3 * (iy % 3), torch.int64)
        self.dx = ops.from_numpy(Xl[:, 0] - (x0 + (ix + 0.5) * lx / bx), torch.float64)
        self.dy = ops.from_numpy(Xl[:, 1] - (y0 + (iy + 0.5) * ly / by), torch.float64)
        fm = free_mask.to(torch.float64).view(-1, 2)
        self.fx, self.fy = fm[:, 0].contiguous(), fm[:, 1].contiguous()  # Z vanishes on fixed dofs
        self.ncoarse = 3 * self.nb
        # neighbour of bin b with colour c inside b's 3 x 3 neighbourhood (-1: outside the grid)
        b = np.arange(self.nb)
        bix, biy = b % bx, b // bx
        nbr = np.full((9, self.nb), -1, dtype=np.int64)
        for c in range(9):
            jx = bix + ((c % 3 - bix % 3 + 4) % 3 - 1)
            jy = biy + ((c // 3 - biy % 3 + 4) % 3 - 1)
            ok = (jx >= 0) & (jx < bx) & (jy >= 0) & (jy < by)
            nbr[c, ok] = (jx + bx * jy)[ok]
        self.nbr = ops.from_numpy(nbr, torch.int64)
        self.Einv = None

    def ZT(self, v):
        """Z^T v for a local vector (un-summed product, or a consistent vector times the ownership weights)."""
        torch = self.ops.torch
        vx, vy = v[0::2] * self.fx, v[1::2] * self.fy
        out = torch.zeros(self.nb, 3, dtype=torch.float64, device=v.device)
        out[:, 0].index_add_(0, self.bin, vx)
        out[:, 1].index_add_(0, self.bin, vy)
        out[:, 2].index_add_(0, self.bin, self.dx * vy - self.dy * vx)
        return out.view(-1)

    def Z(self, y, out):
        """out = Z y (consistent local vector)."""
        yb = y.view(self.nb, 3)[self.bin]
        out[0::2] = (yb[:, 0] - self.dy * yb[:, 2]) * self.fx
        out[1::2] = (yb[:, 1] + self.dx * yb[:, 2]) * self.fy
        return out

    def setup(self, K, group):
        """E = Z^T K Z summed over ranks, pseudo-inverted (empty bins and bins whose nodes do not span the three
        modes give zero eigenvalues, which are dropped)."""
        ops, torch = self.ops, self.ops.torch
        n = self.bin.numel() * 2
        E = torch.zeros(self.ncoarse, self.ncoarse, dtype=torch.float64, device=self.bin.device)
        v, w = ops.zeros(n), ops.zeros(n)
        rows = torch.arange(self.nb, device=self.bin.device)
        for c in range(9):
            sel = (self.colour == c).to(torch.float64)
            col = self.nbr[c]
            ok = col >= 0
            for k in range(3):
                v.zero_()
                if k == 0:
                    v[0::2] = sel * self.fx
                elif k == 1:
                    v[1::2] = sel * self.fy
                else:
                    v[0::2] = -self.dy * sel * self.fx
                    v[1::2] = self.dx * sel * self.fy
                ops.spmv(K, v, w)
                t = self.ZT(w).view(self.nb, 3)
                for kp in range(3):
                    E[3 * rows[ok] + kp, 3 * col[ok] + k] += t[ok, kp]
        _all_reduce(E, group)
        E = 0.5 * (E + E.T)
        lam, Q = torch.linalg.eigh(E)
        keep = lam > 1e-10 * lam.max()
        self.Einv = (Q[:, keep] / lam[keep]) @ Q[:, keep].T
        return self


def schwarz_prepare(ops, K, iface):
    """Numeric AMG setup of this rank's block for dist_pcg_schwarz: the local matrix with a tenth of the missing part of
    its interface diagonals added (assembled minus local value), which pins floating pieces and costs compact
    partitions nothing (tools/dist_schwarz_rehearsal.py: 0.01 .. 0.1 give the same counts, 1.0 twice as many, 0 fails
    on floating pieces).  Returns the shifted copy (it must stay alive while the hierarchy is in use)."""
    import os

    dloc = ops.diag(K)
    dsum = iface.sum(dloc.clone())
    Kp = K.clone()
    ops.add_to_diag(Kp, (dsum - dloc) * float(os.environ.get("PF_SCHWARZ_EPS", "0.1")))
    ops.amg_prepare(Kp)
    return Kp


def dist_pcg_schwarz(ops, K, b, inv_mult, own_w, iface, rtol=1e-8, atol=0.0, maxiter=None, check_every=1,
                     coarse=None, prepared=False):
    """PCG over ranks with a one-level additive Schwarz preconditioner whose subdomain solves are the LOCAL AMG
    V-cycles:  M^-1 = sum_i R_i^T D A_i~^-1 D R_i,  D = 1/multiplicity on interface dofs (symmetric, so CG applies),
    A_i~^-1 = one V-cycle on rank i's block with a fraction of the missing part of its interface DIAGONALS added
    (assembled minus local value): the raw block is a Neumann-type problem and is singular whenever a piece of the
    rank's mesh touches no Dirichlet node (always the case for scattered element blocks); the shifted diagonal makes
    every local problem definite without any communication in the cycle.  There is no global coarse level yet, so
    the iteration count grows with the number of ranks (strips of the 512x256 cantilever: 16 / 165 / ~300 iterations
    for 1 / 2 / 4 ranks against 3600 for Jacobi).  Jacobi-PCG needs ~10^4 iterations on the 2048x1024 plane-stress
    system; a V-cycle per rank brings that to tens, at TWO all-reduces per iteration:
      1. interface partial sums of w = K p, with delta = p.w riding along (p_loc.(K_loc p_loc) summed over ranks);
      2. interface sums of z = M^-1 r, with gamma = r.z (sum of (D r_loc).z_loc before the interface sum) and r.r.
    Scalars stay on the device; the residual is read back every `check_every` iterations.  x0 = 0.
    coarse: a BinCoarseSpace -- adds the global correction Z E^+ Z^T (two-level method); its Z^T (K p) rides along
    with all-reduce 1, so the number of collectives per iteration stays two.
    prepared: schwarz_prepare was already called for this matrix on ops' context (constant matrices: the filter).
    Returns (x_local, iters, relres); raises RuntimeError("cg failed with code <iters>") at maxiter."""
    torch = ops.torch
    n, G = ops.n, iface.G
    nc = coarse.ncoarse if coarse is not None else 0
    buf = ops.zeros(G + 3 + nc) if nc else iface.buf
    tail = buf[G:G + 3]  # three scalars ride along with the interface sums
    ctail = buf[G + 3:]  # and the coarse-space moments of the un-summed product
    if hasattr(ops, "begin"):
        ops.begin()
    Kp = None if prepared else schwarz_prepare(ops, K, iface)  # prepared: the caller did it for this (constant) matrix
    x, r = ops.zeros(n), b.clone()
    w, z, zl, rs = ops.zeros(n), ops.zeros(n), ops.zeros(n), ops.zeros(n)
    cr = None
    if coarse is not None:
        coarse.setup(K, iface.group)
        cr = coarse.ZT(b * own_w)  # Z^T r, kept up to date from Z^T (K p)
        _all_reduce(cr, iface.group)
        zc = ops.zeros(n)

    def precond():  # z = M^-1 r;  returns (gamma, rr) as 0-dim device tensors
        torch.mul(r, inv_mult, out=rs)
        ops.amg_vcycle(rs, zl)
        gamma_loc = torch.dot(rs, zl)
        torch.mul(zl, inv_mult, out=z)
        buf.zero_()
        ops.copy_indexed(iface.n, iface.local, iface.slots, z, buf)
        tail[0] = gamma_loc
        tail[1] = torch.dot(r * own_w, r)
        _all_reduce(buf, iface.group)
        ops.copy_indexed(iface.n, iface.slots, iface.local, buf, z)
        g = tail[0].clone()
        if coarse is not None:  # + Z E^+ Z^T r: redundant on every rank, no communication
            y = coarse.Einv @ cr
            z.add_(coarse.Z(y, zc))
            g = g + torch.dot(cr, y)
        return g, tail[1].clone()

    gamma, rr = precond()
    bnorm = float(rr.item()) ** 0.5
    tol = max(rtol * bnorm, atol)
    p = z.clone()
    if maxiter is None:
        maxiter = 10 * n
    it, rnorm = 0, bnorm
    while it < maxiter and rnorm > tol:
        ops.spmv(K, p, w)
        delta_loc = torch.dot(p, w)
        buf.zero_()
        ops.copy_indexed(iface.n, iface.local, iface.slots, w, buf)
        tail[0] = delta_loc
        if coarse is not None:
            ctail.copy_(coarse.ZT(w))
        _all_reduce(buf, iface.group)
        ops.copy_indexed(iface.n, iface.slots, iface.local, buf, w)
        alpha = gamma / tail[0]
        x.add_(p * alpha)
        r.sub_(w * alpha)
        if coarse is not None:
            cr.sub_(ctail * alpha)
        gamma_new, rr = precond()
        p.mul_(gamma_new / gamma).add_(z)
        gamma = gamma_new
        it += 1
        if it % check_every == 0 or it == maxiter:
            rnorm = float(rr.item()) ** 0.5
            if rnorm != rnorm:
                break
    relres = rnorm / bnorm if bnorm > 0 else rnorm
    if not rnorm <= tol:
        raise RuntimeError(f"cg failed with code {it}")
    return x, it, relres


# ------------------------------------------------------------------------------------------
# the model-level wrappers
# ------------------------------------------------------------------------------------------
class _NodeView:
    """The interface of an ElementBlockPartition seen per node instead of per dof."""

    def __init__(self, part):
        d = part.d
        self.n_iface_dofs = part.n_iface_dofs // d
        self.iface_local_dofs = (part.iface_local_dofs[::d] // d).astype(np.int32)
        self.iface_slots = (part.iface_slots[::d] // d).astype(np.int32)


class PartitionedHelmholtz:
    """The Helmholtz filter (pyfem.py:2071-2177) over the same element-block partition as the model it serves:
    K = r0^2 A + M and R = M assembled from the rank's elements, `apply` / `apply_gradient` as distributed PCG solves
    (additive Schwarz with the local AMG cycle).  Inputs and outputs are this rank's nodal values (device tensors,
    consistent on shared nodes)."""

    def __init__(self, r0, X, conn, quadrature, basis, group=None, rank=None, world=None, device=None, elem_order="index"):
        import torch.distributed as dist

        from . import pyfem

        if world is None:
            world = dist.get_world_size(group) if dist.is_initialized() else 1
            rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.group, self.rank, self.world = group, rank, world
        X = np.asarray(X, dtype=float)
        self.nnodes_global = X.shape[0]
        perm = centroid_element_order(X, conn) if elem_order == "centroid" else None
        self.part = ElementBlockPartition(conn, X.shape[0], world, rank, 1, elem_perm=perm)
        pt = self.part
        self.model = pyfem.Helmholtz(r0, X[pt.nodes], pt.conn_local, quadrature, basis, device=device)
        self.ops = HipLocalOps(self.model.hctx)
        self.iface = Interface(self.ops, pt, group)
        torch = self.ops.torch
        self.own_w = self.ops.from_numpy(pt.own_weight, torch.float64)
        self.inv_mult = self.ops.from_numpy(1.0 / pt.multiplicity, torch.float64)
        self.rtol = 1e-10
        self.last_iters = None
        self._Kp = schwarz_prepare(self.ops, self.model.K.vals, self.iface)  # the filter matrix never changes

    def _solve(self, b):
        x, it, rr = dist_pcg_schwarz(self.ops, self.model.K.vals, b, self.inv_mult, self.own_w, self.iface, rtol=self.rtol,
                                     prepared=True)
        self.last_iters = it
        return x

    def apply(self, x_local):
        """rho = K^-1 R x (pyfem.py:2102-2107)."""
        rhs = self.ops.zeros(self.ops.n)
        self.ops.spmv(self.model.R.vals, x_local, rhs)
        return self._solve(self.iface.sum(rhs))

    def apply_gradient(self, g_local):
        """R^T K^-1 g (pyfem.py:2110-2115); g consistent on shared nodes."""
        y = self._solve(g_local.clone())
        out = self.ops.zeros(self.ops.n)
        self.ops.spmv(self.model.R.vals, y, out)
        return self.iface.sum(out)

    def local_nodal(self, v_global):
        return self.ops.from_numpy(np.asarray(v_global, dtype=float)[self.part.nodes], self.ops.torch.float64)

    def gather_nodal(self, v_local):
        """Global nodal vector on every rank (one all-reduce of nnodes values)."""
        torch = self.ops.torch
        out = torch.zeros(self.nnodes_global, dtype=torch.float64, device=v_local.device)
        out[self.ops.from_numpy(self.part.nodes, torch.int64)] = v_local * self.own_w
        _all_reduce(out, self.group)
        return out



class PartitionedElasticity:
    """LinearElasticity (pyfem.py:1667-2068) over an element-block partition.

    Same constructor arguments as pyfem.LinearElasticity (global mesh, global dof ids), plus the
    process group.  compute_jacobian()/apply BCs/solve act on the rank-local block; `solve`
    returns the local solution and `gather_global` assembles the global vector on every rank."""

    def __init__(self, X, conn, dof_fixed, dof_fixed_vals, nodal_force, quadrature, basis, E=10.0, nu=0.3, p=0.0,
                 group=None, rank=None, world=None, device=None, elem_order="index"):
        import torch.distributed as dist

        from . import pyfem

        if world is None:
            world = dist.get_world_size(group) if dist.is_initialized() else 1
            rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.group, self.rank, self.world = group, rank, world
        X = np.asarray(X, dtype=float)
        self.nnodes_global = X.shape[0]
        self._bbox = (tuple(X.min(axis=0)), tuple(X.max(axis=0)))
        self._nelems_global = int(np.asarray(conn).shape[0])
        self._coarse = None
        if elem_order not in ("index", "centroid"):
            raise ValueError("elem_order must be 'index' (contiguous element-index blocks) or 'centroid'")
        perm = centroid_element_order(X, conn) if elem_order == "centroid" else None
        self.part = ElementBlockPartition(conn, X.shape[0], world, rank, 2, elem_perm=perm)
        pt = self.part
        fixed_local, present = pt.local_dofs_of(dof_fixed)
        vals_local = None if dof_fixed_vals is None else np.asarray(dof_fixed_vals, dtype=float)[present]
        force_local = {}
        for node, f in nodal_force.items():
            pos = np.searchsorted(pt.nodes, node)
            if pos < pt.nodes.size and pt.nodes[pos] == node:
                force_local[int(pos)] = f
        self.model = pyfem.LinearElasticity(X[pt.nodes], pt.conn_local, fixed_local, vals_local, force_local,
                                            quadrature, basis, E=E, nu=nu, p=p, device=device)
        if not force_local:  # compute_rhs of the reference indexes with the (possibly empty) key list
            self.model.compute_rhs = lambda: self.model.rhs
        self.ops = HipLocalOps(self.model.hctx)
        self.iface = Interface(self.ops, pt, group)
        torch = self.ops.torch
        self.own_w = self.ops.from_numpy(pt.own_weight, torch.float64)
        self.inv_mult = self.ops.from_numpy(1.0 / pt.multiplicity, torch.float64)
        self._own_nodes = self.ops.from_numpy(pt.owned_mask_nodes.astype(np.float64), torch.float64)
        self._niface = None
        self.last_iters = None

    @property
    def nelems_local(self):
        return self.part.conn_local.shape[0]

    def compute_jacobian(self, rho=1.0, out=None):
        """Local block of K; rho is the GLOBAL nodal density (or a scalar)."""
        if hasattr(rho, "__len__"):
            rho = np.asarray(rho)[self.part.nodes]
        return self.model.compute_jacobian(rho, out=out)

    def apply_dirichlet_bcs(self, K, rhs_local_consistent):
        """BCs on the local block.  rhs is consistent on entry and on return."""
        m = self.model
        b = rhs_local_consistent * self.inv_mult  # split so that the interface sum restores it
        m.hctx.apply_dirichlet(K.vals, b, m._fixed_mask, m._fixed_vals, True, diag_scale=self.own_w)
        K.invalidate()
        self.iface.sum(b)
        fixed = self.ops.from_numpy(m.dof_fixed, self.ops.torch.int64)
        if fixed.numel():
            b[fixed] = 0.0 if m._fixed_vals is None else m._fixed_vals[fixed]
        return K, b

    def solve(self, rho=1.0, rtol=1e-10, maxiter=None, check_every=None, precond="jacobi"):
        """Assembler.solve(method="cg") over the partition (pyfem.py:2299-2317).  precond: "jacobi" (single-reduction
        PCG, one all-reduce per iteration) or "amg" (additive Schwarz with one local AMG V-cycle per rank, two
        all-reduces per iteration) or "amg2" (the same plus a global rigid-body coarse space on a uniform grid of bins:
        a third fewer iterations for a 27-product setup and a dense 3 nbins eigen-decomposition per matrix)."""
        K = self.compute_jacobian(rho)
        rhs = self.ops.from_numpy(self.model.compute_rhs(), self.ops.torch.float64)
        K, b = self.apply_dirichlet_bcs(K, rhs)
        if precond in ("amg", "amg2"):
            coarse = None
            if precond == "amg2" and self.world > 1:  # two-level: global rigid-body coarse space on a uniform grid
                if self._coarse is None:
                    free = 1 - self.model._fixed_mask.to(self.ops.torch.int32)
                    self._coarse = BinCoarseSpace(self.ops, self.model.X, free, self._bbox, self._nelems_global)
                coarse = self._coarse
            x, it, rr = dist_pcg_schwarz(self.ops, K.vals, b, self.inv_mult, self.own_w, self.iface, rtol=rtol,
                                         maxiter=maxiter, check_every=check_every or 1, coarse=coarse)
            self.last_iters, self.last_relres = it, rr
            return x
        if precond != "jacobi":
            raise ValueError("precond must be 'jacobi', 'amg' (one-level Schwarz) or 'amg2' (with the bin coarse space)")
        check_every = check_every or 50
        dinv = 1.0 / self.iface.sum(self.ops.diag(K.vals))
        x, it, rr = dist_cg(self.ops, K.vals, b, dinv, self.own_w, self.iface, rtol=rtol, maxiter=maxiter,
                            check_every=check_every)
        self.last_iters, self.last_relres = it, rr
        return x

    # -- topology-optimisation callbacks over the partition (BASELINE configuration 5; pyfem.py:1798-1870) ------------
    def _node_iface(self):
        """Interface sums of NODAL vectors (densities, sensitivities): the dof interface reduced to one slot per node."""
        if self._niface is None:
            self._niface = Interface(self.ops, _NodeView(self.part), self.group)
        return self._niface

    def local_nodal(self, v_global):
        """This rank's entries of a global nodal vector, as a device tensor."""
        return self.ops.from_numpy(np.asarray(v_global, dtype=float)[self.part.nodes], self.ops.torch.float64)

    def compliance(self, rho, precond="amg", rtol=1e-10):
        """(c, u_local): c = f.u summed over the owned dofs of every rank (one scalar all-reduce); rho is the global
        nodal density (numpy) or this rank's part (device tensor)."""
        torch = self.ops.torch
        if isinstance(rho, torch.Tensor):
            self._rho_local = rho
            K = self.model.compute_jacobian(rho)
        else:
            self._rho_local = None
            K = self.compute_jacobian(rho)
        rhs = self.ops.from_numpy(self.model.compute_rhs(), torch.float64)
        K, b = self.apply_dirichlet_bcs(K, rhs)
        if precond == "jacobi":
            dinv = 1.0 / self.iface.sum(self.ops.diag(K.vals))
            u, it, rr = dist_cg(self.ops, K.vals, b, dinv, self.own_w, self.iface, rtol=rtol)
        else:
            u, it, rr = dist_pcg_schwarz(self.ops, K.vals, b, self.inv_mult, self.own_w, self.iface, rtol=rtol)
        self.last_iters, self.last_relres = it, rr
        c = torch.dot(b * self.own_w, u).reshape(1)
        _all_reduce(c, self.group)
        return float(c.item()), u

    def compliance_grad(self, rho, u_local):
        """-d(u^T K u)/d rho on this rank's nodes (pyfem.py:1837-1848): sensitivities of the local elements, summed over
        the ranks that share a node.  rho as in compliance(); returns a device tensor over the local nodes."""
        torch = self.ops.torch
        rho_l = rho if isinstance(rho, torch.Tensor) else self.local_nodal(rho)
        g = -self.model._compute_K_dv_sens(rho_l, u_local, u_local)
        return self._node_iface().sum(g)

    def volume(self, rho_local):
        """rho.sum() / nnodes over the whole mesh (pyfem.py:1851-1859) from this rank's nodes: one scalar all-reduce."""
        v = (rho_local * self._own_nodes).sum().reshape(1)
        _all_reduce(v, self.group)
        return float(v.item()) / self.nnodes_global

    def gather_nodal(self, v_local):
        """Global nodal vector on every rank from consistent local ones (one all-reduce of nnodes values)."""
        torch = self.ops.torch
        out = torch.zeros(self.nnodes_global, dtype=torch.float64, device=v_local.device)
        out[self.ops.from_numpy(self.part.nodes, torch.int64)] = v_local * self._own_nodes
        _all_reduce(out, self.group)
        return out

    def gather_global(self, x_local):
        """Global vector on every rank (tests / small problems)."""
        torch = self.ops.torch
        out = torch.zeros(2 * self.nnodes_global, dtype=torch.float64, device=x_local.device)
        gd = self.ops.from_numpy(self.part.global_dofs(), torch.int64)
        out[gd] = x_local * self.own_w
        _all_reduce(out, self.group)
        return out
