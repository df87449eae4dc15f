"""Replay of the fused assembly kernel's index arithmetic on the host (run with -m gpu; the maps are built on the
device by pf_pattern_build).

k_assemble_v4 (csrc/pf_assemble.hip) prefetches UNCONDITIONALLY: every lane of every tile loads a visit record, a row
pointer, P coordinate pairs and P densities whether or not the lane has a pair to integrate (round 1, commit 7bf4609,
fixed a gather 15 nodes past the end of X that such a lane produced).  Records and row pointers are read through
per-tile buffer resources (lanes past the tile's pairs get zeros from the hardware's range check), the gathers
through resources over X / rho.  This test downloads tile_desc / vis_rec / n2e_ptr / nptr through
pf_pattern_get_tiles and re-derives, in numpy, every index the kernel forms -- v4_load_tile, v4_load_rec,
v4_load_visit, the LDS adds of v4_tile and its copy-out -- for all 64 lanes of all tiles, asserting that each one lies
inside its array and that the rotated records describe the mesh they were built from.  Meshes: ragged sizes around
the 16-node / 64-pair tile limits, shuffled triangles, a valence-63 hub, runs of unused nodes, a last tile whose
slots reach past the last node."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pf():
    import torch

    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    from pyfem_gpu_testflight_amd import pyfem

    return pyfem


def _replay(maps, nnodes, P, max_row_blocks, d, conn):
    td, vr, n2e, nptr, npt = maps["tile_desc"], maps["vis_rec"], maps["n2e_ptr"], maps["nptr"], maps["npt"]
    ntiles, nvis = td.shape[0], vr.shape[0]
    assert nvis == n2e[-1] and npt <= 32
    plane = maps["max_tile_blocks"] | 1  # doubles per value plane of the LDS image (launch_gather: `plane`)
    lanes = np.arange(64)
    radix = nnodes ** np.arange(P, dtype=np.int64)  # test meshes are small: nnodes**P fits int64
    conn_keys = conn.astype(np.int64) @ radix
    covered_nodes = covered_vis = covered_blocks = 0
    # the kernel reads descriptors up to 4 steps ahead, clamped to the wave's last one: always inside [0, ntiles).
    # Tiles are WALKED in descriptor order, which is node order or (scattered numberings) a Morton order of the tiles'
    # first nodes; the partition of nodes / visits / blocks is checked in node order
    for t in np.argsort(td[:, 0], kind="stable"):
        n0, vb, base, w = (int(v) for v in td[t])
        nn_t, nv_t, nblk, deep = w & 63, (w >> 8) & 255, (w >> 16) & 0xFFFF, (w >> 7) & 1
        assert 0 <= n0 and n0 + nn_t <= nnodes and 1 <= nn_t <= npt
        assert nv_t <= 64 and base == nptr[n0] and base + nblk == nptr[n0 + nn_t] and nblk <= maps["max_tile_blocks"]
        assert deep == int((np.diff(n2e[n0:n0 + nn_t + 1]) > 8).any())  # fourth step of the diagonal pre-reduction
        assert n0 == covered_nodes and vb == covered_vis == n2e[n0] and base == covered_blocks  # tiles partition everything
        assert vb + nv_t == n2e[n0 + nn_t] and (d != 2 or nblk <= 160 or nn_t == 1)
        covered_nodes, covered_vis, covered_blocks = n0 + nn_t, vb + nv_t, base + nblk
        # v4_load_rec: all 64 lanes; the record resource covers [vb, vb + nv_t), the pointer resource nn_t + 1 entries
        jj = vb + lanes
        active = lanes < nv_t
        rec = np.where(active[:, None], vr[np.minimum(jj, nvis - 1)], 0)  # out of range: the hardware returns zeros
        rp = np.where(lanes <= nn_t, nptr[np.minimum(n0 + lanes, nnodes)], 0)
        # v4_load_visit: all 64 lanes (gathers are unconditional): own node first, then the rotated element's others
        wv = rec[:, 3].astype(np.int64) & 0xFFFFFFFF
        a, ln = (wv >> 24) & 3, (wv >> 26) & 31
        n = n0 + ln
        nd = np.concatenate([n[:, None], rec[:, :3].astype(np.int64)], axis=1)[:, :P]
        assert ((nd >= 0) & (nd < nnodes)).all(), f"tile {t}: gather outside X / rho"
        assert (ln <= nn_t).all()  # the __shfl source lane holds a row pointer of this tile
        # lanes with a pair: the record belongs to node n0 + ln of this tile, positions stay inside the tile's image
        la = np.flatnonzero(active)
        assert (ln[la] < nn_t).all()
        own = n0 + ln[la]
        assert ((n2e[own] <= jj[la]) & (jj[la] < n2e[own + 1])).all(), "visit record filed under the wrong node"
        assert (a[la] < P).all()
        # the rotated record is the element: (n, o1, .., o_{P-1}) is a cyclic rotation by a of a row of conn
        elem = nd[la][:, :P]
        back = np.stack([elem[np.arange(la.size), (k - a[la]) % P] for k in range(P)], axis=1)
        assert np.isin(back @ radix, conn_keys).all(), "record is not a rotation of an element"
        roff = rp[ln[la]] - base
        for b in range(P):
            pos = (wv[la] >> (6 * b)) & 63
            blk = roff + pos
            assert ((blk >= 0) & (blk < nblk)).all(), f"tile {t}: LDS add outside the tile's segment"
            assert (pos < nptr[own + 1] - nptr[own]).all(), "position byte beyond the node's row"
            assert (blk < plane).all()
        # copy-out: the store resource ends with the segment (nblk blocks); LDS reads run over by < 32 NOUT doubles,
        # which the launcher allocates (pf_assemble.hip launch_gather)
    assert covered_nodes == nnodes and covered_vis == nvis and covered_blocks == nptr[-1]
    # the compact records (used whenever they exist) must decode to exactly the 16-byte ones
    r8, r12 = maps.get("rec8"), maps.get("rec12")
    own = np.repeat(np.arange(nnodes), np.diff(n2e))  # the node of every visit
    w16 = vr[:, 3].astype(np.int64) & 0xFFFFFFFF
    if r8 is not None and P == 4:
        d0, d1 = r8[:, 0].astype(np.int64), r8[:, 1].astype(np.int64)
        sext = lambda v: ((v & 8191) ^ 4096) - 4096  # noqa: E731
        offs = [sext(d0), sext(d0 >> 13), sext((d0 >> 26) | ((d1 & 127) << 6))]
        for k in range(P - 1):
            assert np.array_equal(own + offs[k], vr[:, k].astype(np.int64)), f"rec8: node {k + 1} decodes differently"
        for k in range(P):
            assert np.array_equal((d1 >> (7 + 5 * k)) & 31, (w16 >> (6 * k)) & 63)
        assert np.array_equal((d1 >> 27) & 31, (w16 >> 26) & 31)
    if r8 is not None and P == 3:
        d0, d1 = r8[:, 0].astype(np.int64), r8[:, 1].astype(np.int64)
        sext = lambda v: ((v & 0xFFFF) ^ 0x8000) - 0x8000  # noqa: E731
        assert np.array_equal(own + sext(d0), vr[:, 0].astype(np.int64)) and np.array_equal(own + sext(d0 >> 16), vr[:, 1].astype(np.int64))
        for k in range(P):
            assert np.array_equal((d1 >> (5 * k)) & 31, (w16 >> (6 * k)) & 63)
        assert np.array_equal((d1 >> 15) & 31, (w16 >> 26) & 31) and np.array_equal((d1 >> 20) & 3, (w16 >> 24) & 3)
    if r12 is not None:
        assert P == 3 and np.array_equal(r12.view(np.int32)[:, :2], vr[:, :2]) and np.array_equal(r12[:, 2], vr[:, 3].view(np.uint32))
    return 8 if r8 is not None else 12 if r12 is not None else 16


def _fan_mesh(k):
    ang = 2 * np.pi * np.arange(k) / k
    X = np.concatenate([[[0.0, 0.0]], np.stack([np.cos(ang), np.sin(ang)], axis=1)])
    conn = np.stack([np.zeros(k, dtype=np.int64), 1 + np.arange(k), 1 + (np.arange(k) + 1) % k], axis=1)
    return X, conn


def _meshes(pf):
    out = []
    for nx, ny, et in ((13, 9, "quad"), (16, 16, "quad"), (17, 5, "quad"), (3, 2, "quad"), (11, 8, "tri"), (33, 2, "tri")):
        c = pf.ProblemCreator(nnodes_x=nx, nnodes_y=ny, element_type=et)
        conn, X, _, _ = c.create_linear_elasticity_problem()
        out.append((f"{et}{nx}x{ny}", X, conn))
    # shuffled triangles: valence / row lengths vary per wave
    c = pf.ProblemCreator(nnodes_x=19, nnodes_y=14, element_type="tri")
    conn, X, _, _ = c.create_linear_elasticity_problem()
    rng = np.random.RandomState(4)
    perm = rng.permutation(X.shape[0])
    inv = np.empty_like(perm)
    inv[perm] = np.arange(perm.size)
    out.append(("tri_shuffled", X[perm], inv[conn][rng.permutation(conn.shape[0])]))
    out.append(("fan63",) + _fan_mesh(63))
    out.append(("fan30",) + _fan_mesh(30))
    # 32 unused nodes in the middle and 5 at the end: empty tiles, and a last tile whose slots pass the last node
    c = pf.ProblemCreator(nnodes_x=9, nnodes_y=7)
    conn, X, _, _ = c.create_linear_elasticity_problem()
    shift = np.where(np.arange(X.shape[0]) >= 20, 32, 0)
    X2 = np.zeros((X.shape[0] + 32 + 5, 2))
    X2[np.arange(X.shape[0]) + shift] = X
    out.append(("gaps", X2, conn + shift[conn]))
    return out


@pytest.mark.parametrize("d", [1, 2])
def test_every_index_the_assembly_kernel_forms_is_in_range(pf, d):
    from pyfem_gpu_testflight_amd.device import HipContext

    for name, X, conn in _meshes(pf):
        conn = np.asarray(conn)
        if name == "gaps":  # pf_mesh_set asserts conn spans 0..nnodes-1 (pyfem.py:680-681): make the last node used
            conn = conn.copy()
            conn[-1, 2] = X.shape[0] - 1
        h = HipContext(X, conn, d)
        maps = h.assembly_maps()
        compact = _replay(maps, h.nnodes, h.P, h.max_row_blocks, d, conn)
        # compact records: 8 bytes where the numbering is banded; the hub of fan63 has rows of 64 blocks (5-bit positions
        # do not fit) and the shuffled mesh scatters neighbours, but both stay within 16-bit offsets at these sizes
        assert compact == (12 if name == "fan63" else 8), (name, compact)
        h.close()
    # a numbering that scatters neighbours more than 4095 node ids apart falls back to the 16-byte records
    c = pf.ProblemCreator(nnodes_x=5000, nnodes_y=3)
    conn, X, _, _ = c.create_linear_elasticity_problem()
    h = HipContext(X, conn, d)
    assert h.assembly_maps()["rec8"] is None
    h.close()
    # triangles with neighbours more than 32767 node ids apart: the 12-byte records
    c = pf.ProblemCreator(nnodes_x=40000, nnodes_y=3, element_type="tri")
    conn, X, _, _ = c.create_linear_elasticity_problem()
    h = HipContext(X, conn, d)
    maps = h.assembly_maps()
    assert maps["rec8"] is None and maps["rec12"] is not None
    assert np.array_equal(maps["rec12"].view(np.int32)[:, :2], maps["vis_rec"][:, :2])
    h.close()
