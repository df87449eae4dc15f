"""N>1 host logic under gloo, world_size 2 and 3, on CPU: element-block partition, interface
packing, consistent BCs and the single-reduction CG of pyfem_gpu_testflight_amd/dist.py, with the
numpy test double standing in for the rank-local HIP kernels.  Checked against the oracle's
direct solve of the unpartitioned problem."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _problem(kind):
    from oracle import pyfem_oracle as orc

    if kind == "quad":
        nx, ny = 19, 14
        conn, X = orc.structured_mesh(nx, ny, "quad")
        dof_fixed, force = orc.elasticity_problem(nx, ny)
        return "quad", conn, X, dof_fixed, None, force
    # shuffled triangles: element blocks are scattered, interface nodes shared by up to 3 ranks
    nx, ny = 13, 11
    conn, X = orc.structured_mesh(nx, ny, "tri")
    dof_fixed, force = orc.elasticity_problem(nx, ny)
    rng = np.random.RandomState(11)
    conn = conn[rng.permutation(conn.shape[0])]
    vals = 0.01 * (rng.rand(dof_fixed.size) - 0.5)  # non-zero prescribed displacements
    return "tri", conn, X, dof_fixed, vals, force


def _worker(rank, world, port, kind):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import pyfem_oracle as orc
        from pyfem_gpu_testflight_amd.dist import ElementBlockPartition, Interface, dist_cg
        from tests._cpu_ops import NumpyLocalOps

        etype, conn, X, dof_fixed, vals, force = _problem(kind)
        part = ElementBlockPartition(conn, X.shape[0], world, rank, 2)
        # every element is owned exactly once; local numbering is monotone
        assert part.conn_local.shape[0] == part.elem_range[1] - part.elem_range[0]
        assert np.all(np.diff(part.nodes) > 0)
        assert np.array_equal(part.nodes[part.conn_local], conn[part.elem_range[0]:part.elem_range[1]])
        Xl = X[part.nodes]
        Kl = orc.assemble_csr(orc.elasticity_Ke(Xl, part.conn_local.astype(np.int64), etype),
                              orc.create_conn_dof(part.conn_local.astype(np.int64), 2))
        n = 2 * part.nnodes_local
        assert Kl.shape == (n, n)
        ops = NumpyLocalOps(n)
        iface = Interface(ops, part)
        own_w = torch.from_numpy(part.own_weight)
        # ownership: every global dof owned exactly once
        tot = torch.zeros(2 * X.shape[0], dtype=torch.float64)
        tot[torch.from_numpy(part.global_dofs())] += own_w
        dist.all_reduce(tot)
        assert bool((tot == 1).all())
        # interface sum of the local partial products == global product
        rng = np.random.RandomState(0)
        vg = rng.rand(2 * X.shape[0])
        Kg = orc.assemble_csr(orc.elasticity_Ke(X, conn, etype), orc.create_conn_dof(conn, 2))
        yl = torch.from_numpy(Kl.dot(vg[part.global_dofs()]))
        iface.sum(yl)
        assert np.abs(yl.numpy() - Kg.dot(vg)[part.global_dofs()]).max() < 1e-12 * np.abs(Kg.data).max()
        # consistent rhs + BCs on the local block (what PartitionedElasticity.apply_dirichlet_bcs does)
        rhs_g = orc.elasticity_rhs(2 * X.shape[0], force)
        b = torch.from_numpy(rhs_g[part.global_dofs()] / part.multiplicity)
        fixed_local, present = part.local_dofs_of(dof_fixed)
        vals_local = None if vals is None else vals[present]
        mask = np.zeros(n, dtype=bool)
        mask[fixed_local] = True
        fv = np.zeros(n)
        if vals_local is not None:
            fv[fixed_local] = vals_local
        Kl = Kl.tolil()
        bl = b.numpy()
        Kd = Kl.toarray()
        bl[~mask] -= Kd[np.ix_(~mask, mask)] @ fv[mask]
        Kd[mask, :] = 0.0
        Kd[:, mask] = 0.0
        Kd[mask, mask] = part.own_weight[mask]  # unit diagonal only on the owner
        from scipy import sparse

        Kl = sparse.csr_matrix(Kd)
        iface.sum(b)
        b[torch.from_numpy(fixed_local)] = torch.from_numpy(fv[fixed_local])
        dinv = 1.0 / iface.sum(ops.diag(Kl))
        x, it, relres = dist_cg(ops, Kl, b, dinv, own_w, iface, rtol=1e-12, check_every=7)
        assert it > 5 and relres <= 1e-12
        ug = torch.zeros(2 * X.shape[0], dtype=torch.float64)
        ug[torch.from_numpy(part.global_dofs())] = x * own_w
        dist.all_reduce(ug)
        Kbc, rhs_bc = orc.apply_dirichlet_bcs(Kg, rhs_g, dof_fixed, vals, True)
        from scipy.sparse.linalg import spsolve

        u_ref = spsolve(Kbc.tocsc(), rhs_bc)
        err = np.abs(ug.numpy() - u_ref).max() / np.abs(u_ref).max()
        assert err < 1e-8, err
        # consistency: interface copies agree across ranks
        assert np.abs(x.numpy() - ug.numpy()[part.global_dofs()]).max() < 1e-9 * np.abs(u_ref).max()
        # non-convergence raises like the reference (pyfem.py:2421-2422)
        with pytest.raises(RuntimeError, match="cg failed with code 3"):
            dist_cg(ops, Kl, b, dinv, own_w, iface, rtol=1e-14, maxiter=3)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,kind", [(2, "quad"), (2, "tri"), (3, "tri")])
def test_partitioned_cg_gloo(world, kind):
    mp.spawn(_worker, args=(world, _free_port(), kind), nprocs=world, join=True)


def test_partition_single_rank_is_identity():
    from oracle import pyfem_oracle as orc
    from pyfem_gpu_testflight_amd.dist import ElementBlockPartition

    conn, X = orc.structured_mesh(7, 5, "quad")
    part = ElementBlockPartition(conn, X.shape[0], 1, 0, 2)
    assert part.n_iface_dofs == 0 and np.array_equal(part.conn_local, conn)
    assert np.array_equal(part.nodes, np.arange(X.shape[0])) and part.own_weight.min() == 1.0


def test_centroid_order_shortens_interfaces_of_shuffled_meshes():
    """sort elements by centroid before cutting blocks when element ids carry no locality."""
    from oracle import pyfem_oracle as orc
    from pyfem_gpu_testflight_amd.dist import ElementBlockPartition, centroid_element_order

    conn, X = orc.structured_mesh(41, 33, "tri")
    rng = np.random.RandomState(5)
    conn = conn[rng.permutation(conn.shape[0])]
    perm = centroid_element_order(X, conn)
    assert np.array_equal(np.sort(perm), np.arange(conn.shape[0]))
    world = 4
    g_index = ElementBlockPartition(conn, X.shape[0], world, 0, 2).n_iface_dofs
    parts = [ElementBlockPartition(conn, X.shape[0], world, r, 2, elem_perm=perm) for r in range(world)]
    g_centroid = parts[0].n_iface_dofs
    assert g_centroid * 5 < g_index, (g_centroid, g_index)
    # still a partition: every element exactly once, local meshes consistent with the global one
    seen = np.concatenate([perm[p.elem_range[0]:p.elem_range[1]] for p in parts])
    assert np.array_equal(np.sort(seen), np.arange(conn.shape[0]))
    for p in parts:
        assert np.array_equal(p.nodes[p.conn_local], conn[perm][p.elem_range[0]:p.elem_range[1]])
    # a structured mesh keeps strip-like blocks under either order (no regression for ProblemCreator meshes)
    conn2, X2 = orc.structured_mesh(33, 33, "quad")
    assert ElementBlockPartition(conn2, X2.shape[0], 4, 1, 2, elem_perm=centroid_element_order(X2, conn2)).n_iface_dofs \
        <= 3 * ElementBlockPartition(conn2, X2.shape[0], 4, 1, 2).n_iface_dofs


@pytest.mark.parametrize("kind,world,d", [("tri_shuffled", 3, 2), ("quad_morton", 4, 2), ("quad_strips", 5, 1)])
def test_exchange_tables_line_up_without_a_handshake(kind, world, d):
    """The neighbour-exchange tables of all ranks, built independently from the partition alone, must describe the same
    messages from both ends: simulate every rank's pack -> the receiver's ordered sum in one process and compare with
    the global sum.  Covers nodes shared by 3 and 4 ranks (scattered triangle blocks, the centre of 2 x 2 Morton
    blocks) and the rank-ascending summation order that keeps interface copies bitwise identical."""
    from oracle import pyfem_oracle as orc
    from pyfem_gpu_testflight_amd.dist import ElementBlockPartition, centroid_element_order

    rng = np.random.RandomState(11)
    if kind == "tri_shuffled":
        conn, X = orc.structured_mesh(13, 11, "tri")
        conn = conn[rng.permutation(conn.shape[0])]
        perm = None
    else:
        conn, X = orc.structured_mesh(18, 18, "quad")
        perm = centroid_element_order(X, conn) if kind == "quad_morton" else None
    parts = [ElementBlockPartition(conn, X.shape[0], world, r, d, elem_perm=perm) for r in range(world)]
    tabs = [p.exchange_tables() for p in parts]
    assert max(np.diff(t["src_ptr"]).max() for t in tabs) >= (2 if kind == "quad_strips" else 3)
    glob = np.zeros((world, d * X.shape[0]))
    loc = []
    for r, p in enumerate(parts):
        v = rng.rand(d * p.nnodes_local)
        loc.append(v)
        gd = (p.nodes[:, None] * d + np.arange(d)).ravel()
        glob[r, gd] = v
    results = []
    for r, (p, t) in enumerate(zip(parts, tabs)):
        msgs = {}
        for i, s_ in enumerate(t["nbr_rank"]):
            ts = tabs[s_]
            k = list(ts["nbr_rank"]).index(r)  # r must be a neighbour of s_ as well
            msgs[s_] = loc[s_][ts["send_idx"][ts["nbr_ptr"][k]:ts["nbr_ptr"][k + 1]]]
            assert msgs[s_].size == t["nbr_ptr"][i + 1] - t["nbr_ptr"][i]  # symmetric message sizes
        out = loc[r].copy()
        for i, li in enumerate(t["if_idx"]):
            ranks = t["src_rank"][t["src_ptr"][i]:t["src_ptr"][i + 1]]
            assert np.all(np.diff(ranks) > 0) and r in ranks  # ascending, own rank included
            acc = 0.0
            for q in range(t["src_ptr"][i], t["src_ptr"][i + 1]):
                sr = t["src_rank"][q]
                acc += loc[r][li] if sr == r else msgs[sr][t["src_pos"][q]]
            out[li] = acc
        results.append(out)
    # every rank's result equals the sum in ascending rank order -- bit for bit, on every sharer
    expect = np.zeros(d * X.shape[0])
    for r in range(world):
        expect = expect + glob[r]  # adding exact zeros for non-sharers does not change the bits
    for r, p in enumerate(parts):
        gd = (p.nodes[:, None] * d + np.arange(d)).ravel()
        assert np.array_equal(results[r], expect[gd]), r


def _fused_tables(part, d, tile_nodes):
    """The fused plan's tables from libpyfem_hip.so's HOST builder (pf_host_fuse_tables_c: no GPU involved) for this
    rank's exchange tables."""
    import ctypes as C

    from pyfem_gpu_testflight_amd import _lib

    lib = _lib.load()
    t = part.exchange_tables(d)
    i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)  # noqa: E731
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    nbr, nbr_ptr, send, ifx = i32(t["nbr_rank"]), i32(t["nbr_ptr"]), i32(t["send_idx"]), i32(t["if_idx"])
    nn = part.nnodes_local
    own_w = np.repeat(part.owned_mask_nodes.astype(np.float64), d)
    nt = -(-nn // tile_nodes)
    out = {"ifslot": np.empty(nn, np.int32), "sh_ptr": np.empty(ifx.size // d + 1, np.int32),
           "sh_peer": np.empty(max(send.size // d, 1), np.int32), "sh_pos": np.empty(max(send.size // d, 1), np.int32),
           "tperm": np.empty(nt, np.int32), "flags8": np.empty(nn * d, np.uint8)}
    dims = (C.c_int64 * 4)()
    rc = lib.pf_host_fuse_tables_c(nn, d, tile_nodes, int(nbr.size), p(nbr), p(nbr_ptr), p(send), int(ifx.size), p(ifx), None,
                                   p(own_w), p(out["ifslot"]), p(out["sh_ptr"]), p(out["sh_peer"]), p(out["sh_pos"]),
                                   p(out["tperm"]), p(out["flags8"]), dims)
    assert rc == 0, lib.pf_last_error(None)
    out.update(n_ifnodes=int(dims[0]), n_iftiles=int(dims[1]), ntiles=int(dims[2]), nsharers=int(dims[3]), tables=t)
    out["sh_peer"], out["sh_pos"] = out["sh_peer"][:out["nsharers"]], out["sh_pos"][:out["nsharers"]]
    return out


def _fused_worker(rank, world, port, kind):
    """One FUSED exchange replayed over gloo with the product's own tables: the 'product' walks the tiles in tperm order
    and, for interface nodes, drops the raw row into the message slot of every sharer (sh_peer / sh_pos: what
    k_spmv<..., SEND> stores over xGMI); the 'consumer' sums in ascending rank order through the plan's src tables
    (k_dcg_update / k_halo_recv_apply).  Must equal, bit for bit, dist.py's packed exchange and the global product."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import pyfem_oracle as orc
        from pyfem_gpu_testflight_amd.dist import ElementBlockPartition, Interface
        from tests._cpu_ops import NumpyLocalOps

        etype, conn, X, dof_fixed, vals, force = _problem(kind)
        d, tile = 2, 64
        part = ElementBlockPartition(conn, X.shape[0], world, rank, d)
        F = _fused_tables(part, d, tile)
        t = F["tables"]
        n = d * part.nnodes_local
        Kl = orc.assemble_csr(orc.elasticity_Ke(X[part.nodes], part.conn_local.astype(np.int64), etype),
                              orc.create_conn_dof(part.conn_local.astype(np.int64), d))
        vg = np.random.RandomState(3).rand(d * X.shape[0])
        raw = Kl.dot(vg[part.global_dofs()])  # this rank's product: partial sums on interface rows
        # --- product side: tile walk, interface tiles first; rows of interface nodes go to the sharers' slots
        assert sorted(F["tperm"]) == list(range(F["ntiles"]))
        msgs = {int(r): np.full(int(t["nbr_ptr"][i + 1] - t["nbr_ptr"][i]), np.nan) for i, r in enumerate(t["nbr_rank"])}
        sent_nodes = 0
        for pos, tl in enumerate(F["tperm"]):
            for node in range(tl * tile, min(part.nnodes_local, (tl + 1) * tile)):
                s = F["ifslot"][node]
                if s < 0:
                    continue
                assert pos < F["n_iftiles"], "an interface node outside the tiles that are walked first"
                sent_nodes += 1
                for k in range(F["sh_ptr"][s], F["sh_ptr"][s + 1]):
                    msgs[int(F["sh_peer"][k])][F["sh_pos"][k]:F["sh_pos"][k] + d] = raw[d * node:d * node + d]
        assert sent_nodes == F["n_ifnodes"] == t["if_idx"].size // d
        assert all(not np.isnan(m).any() for m in msgs.values())  # every slot of every message was written exactly by the walk
        # --- the wire (xGMI stores in the product): one message per neighbour
        recv = {r: np.empty_like(m) for r, m in msgs.items()}
        ops_ = []
        for r in msgs:
            ops_.append(dist.P2POp(dist.isend, torch.from_numpy(msgs[r]), r))
            ops_.append(dist.P2POp(dist.irecv, torch.from_numpy(recv[r]), r))
        if ops_:
            for w in dist.batch_isend_irecv(ops_):
                w.wait()
        # --- consumer side: ordered sums through the plan's tables
        y = raw.copy()
        for i, li in enumerate(t["if_idx"]):
            acc = 0.0
            for q in range(t["src_ptr"][i], t["src_ptr"][i + 1]):
                sr = int(t["src_rank"][q])
                acc += raw[li] if sr == rank else recv[sr][t["src_pos"][q]]
            y[li] = acc
        # the packed exchange of dist.py on the same input: same bits
        ops = NumpyLocalOps(n)
        ref = torch.from_numpy(raw.copy())
        Interface(ops, part).sum(ref)
        assert np.array_equal(y, ref.numpy())
        Kg = orc.assemble_csr(orc.elasticity_Ke(X, conn, etype), orc.create_conn_dof(conn, d))
        assert np.abs(y - Kg.dot(vg)[part.global_dofs()]).max() < 1e-12 * np.abs(Kg.data).max()
        # flags: bit 0 owner (lowest sharing rank), bit 1 interface; every global dof owned exactly once
        fl = F["flags8"]
        assert np.array_equal(fl & 1, part.own_weight.astype(np.uint8)) and np.array_equal((fl >> 1) & 1, (part.multiplicity > 1).astype(np.uint8))
        tot = torch.zeros(d * X.shape[0], dtype=torch.float64)
        tot[torch.from_numpy(part.global_dofs())] += torch.from_numpy((fl & 1).astype(np.float64))
        dist.all_reduce(tot)
        assert bool((tot == 1).all())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,kind", [(2, "quad"), (3, "tri")])
def test_fused_exchange_replayed_over_gloo(world, kind):
    mp.spawn(_fused_worker, args=(world, _free_port(), kind), nprocs=world, join=True)
