"""
CPU oracle for the pyfem hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

This file is a from-scratch numpy/scipy restatement of the reference's
assembly + solve path (per-element Ke quadrature -> COO->CSR scatter ->
Dirichlet BCs -> sparse solve).  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import it; the product package
(pyfem_gpu_testflight_amd) never does.

Parity pin: every function here is checked in tests/test_oracle_golden.py
against fixtures in tests/golden/*.npz that were produced by importing the
reference itself (oracle/gen_golden.py, run in the build container where
/root/reference exists) and against the known-answer values the survey
recorded.

Reference lines each function follows are cited as  ref: <file>:<lines>
(paths relative to the reference checkout).
"""
import numpy as np
from scipy import sparse
from scipy.sparse.linalg import spsolve, cg, LinearOperator


# --------------------------------------------------------------------------
# tables                                            ref: pyfem.py:68-94, 253-284, 341-377
# --------------------------------------------------------------------------
def quadrature(etype):
    """Points and weights.  quad: 2x2 Gauss, order (-,-),(+,-),(+,+),(-,+), w=1
    (ref: pyfem.py:83-94).  tri: one point (1/3,1/3), w = 0.5 (ref: pyfem.py:68-80)."""
    if etype == "quad":
        g = 1.0 / np.sqrt(3.0)
        pts = np.array([[-g, -g], [g, -g], [g, g], [-g, g]])
        return pts, np.ones(4)
    if etype == "tri":
        return np.array([[1.0 / 3, 1.0 / 3]]), np.array([0.5])
    if etype == "block":  # 2x2x2 Gauss, z fastest, x slowest, w=1 (ref: pyfem.py:97-112)
        g = 1.0 / np.sqrt(3.0)
        pts = np.array([[sx * g, sy * g, sz * g] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)])
        return pts, np.ones(8)
    if etype == "tet":  # 5-point rule on the unit tetrahedron, centroid weight -4/5 (ref: pyfem.py:115-134)
        pts = np.full((5, 3), 1.0 / 6)
        pts[0] = 0.25
        pts[1, 0] = pts[2, 1] = pts[3, 2] = 0.5
        return pts, np.array([-0.8, 0.45, 0.45, 0.45, 0.45])
    if etype == "brick20":  # 3x3x3 Gauss, x fastest; the weights are the reference's 10-digit literals (pyfem.py:137-172)
        g = np.sqrt(0.6)
        k, j, i = np.meshgrid([-1, 0, 1], [-1, 0, 1], [-1, 0, 1], indexing="ij")
        pts = g * np.stack([i.ravel(), j.ravel(), k.ravel()], axis=1)
        nzero = (np.stack([i.ravel(), j.ravel(), k.ravel()], axis=1) == 0).sum(axis=1)
        return pts, np.array([0.1714677641, 0.2743484225, 0.4389574760, 0.7023319616])[nzero]
    raise ValueError(f"unknown element_type: {etype}")


BRICK20_NODES = np.array(  # corners CCW on z=-1 then z=+1, mid-edges of the two faces, then the vertical edges
    [[-1, -1, -1], [1, -1, -1], [1, 1, -1], [-1, 1, -1], [-1, -1, 1], [1, -1, 1], [1, 1, 1], [-1, 1, 1],
     [0, -1, -1], [1, 0, -1], [0, 1, -1], [-1, 0, -1], [0, -1, 1], [1, 0, 1], [0, 1, 1], [-1, 0, 1],
     [-1, -1, 0], [1, -1, 0], [1, 1, 0], [-1, 1, 0]], dtype=float)  # ref: pyfem.py:461-484


def shape_tables(etype):
    """N (Q,P) and dN/dxi (Q,P,2) at the quadrature points
    (ref: pyfem.py:222-250 driver, :262-284 quad, :365-377 tri)."""
    pts, _ = quadrature(etype)
    if etype == "tet":  # quadratic tetrahedron in the volume coordinates t = (1-x-y-z, x, y, z)  (ref: pyfem.py:380-445)
        t = np.concatenate([1.0 - pts.sum(axis=1, keepdims=True), pts], axis=1)  # (Q,4)
        dt = np.array([[-1.0, -1.0, -1.0], [1.0, 0, 0], [0, 1.0, 0], [0, 0, 1.0]])  # (4,3)
        edges = [(0, 1), (1, 2), (2, 0), (0, 3), (1, 3), (2, 3)]  # nodes 4..9
        N = np.empty((pts.shape[0], 10))
        dN = np.empty((pts.shape[0], 10, 3))
        N[:, :4] = t * (2.0 * t - 1.0)
        dN[:, :4] = (4.0 * t - 1.0)[:, :, None] * dt[None]
        for n, (a, b) in enumerate(edges):
            N[:, 4 + n] = 4.0 * t[:, a] * t[:, b]
            dN[:, 4 + n] = 4.0 * (t[:, a, None] * dt[b][None] + t[:, b, None] * dt[a][None])
        return N, dN
    if etype == "brick20":  # serendipity brick (ref: pyfem.py:448-632)
        c = BRICK20_NODES
        f = 1.0 + pts[:, None, :] * c[None, :, :]  # (Q,20,3)
        N = np.empty((pts.shape[0], 20))
        dN = np.empty((pts.shape[0], 20, 3))
        s8 = (pts[:, None, :] * c[None, :8, :]).sum(axis=2)  # xi.c for the corners
        N[:, :8] = 0.125 * f[:, :8, 0] * f[:, :8, 1] * f[:, :8, 2] * (s8 - 2.0)
        for d in range(3):
            o1, o2 = (d + 1) % 3, (d + 2) % 3
            dN[:, :8, d] = 0.125 * c[None, :8, d] * f[:, :8, o1] * f[:, :8, o2] * (s8 + pts[:, None, d] * c[None, :8, d] - 1.0)
        for n in range(8, 20):
            z = int(np.flatnonzero(c[n] == 0.0)[0])  # the direction the edge runs along
            o1, o2 = (z + 1) % 3, (z + 2) % 3
            q = 1.0 - pts[:, z] ** 2
            N[:, n] = 0.25 * q * f[:, n, o1] * f[:, n, o2]
            dN[:, n, z] = -0.5 * pts[:, z] * f[:, n, o1] * f[:, n, o2]
            dN[:, n, o1] = 0.25 * q * c[n, o1] * f[:, n, o2]
            dN[:, n, o2] = 0.25 * q * f[:, n, o1] * c[n, o2]
        return N, dN
    if etype == "block":  # trilinear hexahedron, nodes CCW on z=-1 then z=+1 (ref: pyfem.py:287-338)
        sg = np.array([[-1, -1, -1], [1, -1, -1], [1, 1, -1], [-1, 1, -1],
                       [-1, -1, 1], [1, -1, 1], [1, 1, 1], [-1, 1, 1]], dtype=float)
        f = 1.0 + pts[:, None, :] * sg[None, :, :]  # (Q,P,3): 1 + s_l xi
        N = 0.125 * f[..., 0] * f[..., 1] * f[..., 2]
        dN = np.empty((8, 8, 3))
        dN[..., 0] = 0.125 * sg[None, :, 0] * f[..., 1] * f[..., 2]
        dN[..., 1] = 0.125 * sg[None, :, 1] * f[..., 0] * f[..., 2]
        dN[..., 2] = 0.125 * sg[None, :, 2] * f[..., 0] * f[..., 1]
        return N, dN
    if etype == "quad":
        xi, et = pts[:, 0], pts[:, 1]
        N = 0.25 * np.stack(
            [(1 - xi) * (1 - et), (1 + xi) * (1 - et), (1 + xi) * (1 + et), (1 - xi) * (1 + et)],
            axis=1,
        )
        dN = np.zeros((4, 4, 2))
        dN[:, 0] = np.stack([-0.25 * (1 - et), -0.25 * (1 - xi)], axis=1)
        dN[:, 1] = np.stack([0.25 * (1 - et), -0.25 * (1 + xi)], axis=1)
        dN[:, 2] = np.stack([0.25 * (1 + et), 0.25 * (1 + xi)], axis=1)
        dN[:, 3] = np.stack([-0.25 * (1 + et), 0.25 * (1 - xi)], axis=1)
        return N, dN
    L1, L2 = pts[:, 0], pts[:, 1]
    N = np.stack([L1, L2, 1 - L1 - L2], axis=1)
    dN = np.zeros((1, 3, 2))
    dN[0] = np.array([[1.0, 0.0], [0.0, 1.0], [-1.0, -1.0]])
    return N, dN


# --------------------------------------------------------------------------
# meshes                                            ref: pyfem.py:2469-2520, 2728-2773
# --------------------------------------------------------------------------
def structured_mesh(nnodes_x, nnodes_y, etype="quad"):
    """ProblemCreator semantics: Lx=(nx-1)/(ny-1), Ly=1, node id = i + j*nnodes_x,
    quad e=i+j*nex -> [(j,i),(j,i+1),(j+1,i+1),(j+1,i)]; tri splits each quad into
    [(j,i),(j,i+1),(j+1,i+1)] and [(j+1,i+1),(j+1,i),(j,i)]  (ref: pyfem.py:2469-2520)."""
    Lx = (nnodes_x - 1) / (nnodes_y - 1)
    x = np.linspace(0, Lx, nnodes_x)
    y = np.linspace(0, 1.0, nnodes_y)
    X = np.stack([np.tile(x, nnodes_y), np.repeat(y, nnodes_x)], axis=1)
    j, i = np.meshgrid(np.arange(nnodes_y - 1), np.arange(nnodes_x - 1), indexing="ij")
    n00 = (i + j * nnodes_x).ravel()
    n10, n11, n01 = n00 + 1, n00 + 1 + nnodes_x, n00 + nnodes_x
    if etype == "quad":
        conn = np.stack([n00, n10, n11, n01], axis=1)
    elif etype == "tri":
        conn = np.empty((2 * n00.size, 3), dtype=np.int64)
        conn[0::2] = np.stack([n00, n10, n11], axis=1)
        conn[1::2] = np.stack([n11, n01, n00], axis=1)
    else:
        raise ValueError(f"unknown element_type: {etype}")
    return conn.astype(np.int64), X


def structured_mesh3d(nnodes_x, nnodes_y, nnodes_z, Lx=None, Ly=None, Lz=None):
    """ProblemCreator(element_type="block"): node id = i + j*nx + k*nx*ny, element e = i + j*nex + k*nex*ney with
    nodes [(k,j,i),(k,j,i+1),(k,j+1,i+1),(k,j+1,i)] then the same on k+1; default Lx=(nx-1)/(ny-1), Ly=1,
    Lz=(nz-1)/(ny-1)  (ref: pyfem.py:2469-2492, 2531-2548)."""
    Lx = (nnodes_x - 1) / (nnodes_y - 1) if Lx is None else Lx
    Ly = 1.0 if Ly is None else Ly
    Lz = (nnodes_z - 1) / (nnodes_y - 1) if Lz is None else Lz
    x, y, z = np.linspace(0, Lx, nnodes_x), np.linspace(0, Ly, nnodes_y), np.linspace(0, Lz, nnodes_z)
    X = np.stack([np.tile(x, nnodes_y * nnodes_z), np.tile(np.repeat(y, nnodes_x), nnodes_z),
                  np.repeat(z, nnodes_x * nnodes_y)], axis=1)
    k, j, i = np.meshgrid(np.arange(nnodes_z - 1), np.arange(nnodes_y - 1), np.arange(nnodes_x - 1), indexing="ij")
    n0 = (i + j * nnodes_x + k * nnodes_x * nnodes_y).ravel()
    lo = np.stack([n0, n0 + 1, n0 + 1 + nnodes_x, n0 + nnodes_x], axis=1)
    conn = np.concatenate([lo, lo + nnodes_x * nnodes_y], axis=1)
    return conn.astype(np.int64), X


TET_CORNERS = np.array(  # the six tetrahedra of a 2x2x2-cell patch, corner offsets (x,y,z) in cells (ref: pyfem.py:2547-2556)
    [[(0, 0, 0), (2, 0, 0), (2, 2, 0), (2, 2, 2)], [(0, 0, 0), (2, 0, 0), (2, 2, 2), (0, 0, 2)],
     [(2, 0, 0), (2, 2, 2), (0, 0, 2), (2, 0, 2)], [(0, 0, 0), (0, 2, 0), (0, 2, 2), (2, 2, 0)],
     [(0, 0, 0), (0, 2, 2), (0, 0, 2), (2, 2, 2)], [(0, 0, 0), (2, 2, 2), (2, 2, 0), (0, 2, 2)]])


def structured_mesh3d_ho(nnodes_x, nnodes_y, nnodes_z, etype):
    """ProblemCreator(element_type="tet" | "brick20") on the same node grid as the block mesh (odd node counts).
    tet: every 2x2x2-cell patch (patches x fastest) holds six 10-node tetrahedra: corners from TET_CORNERS, then the
    midpoints of the edges (0,1),(1,2),(0,2),(0,3),(1,3),(2,3)  (ref: pyfem.py:2538-2655).
    brick20: one brick per 2x2x2-cell patch, corners, then the mid-edge nodes of the bottom face, the top face and the
    four vertical edges (ref: pyfem.py:2657-2700).  Grid nodes no element uses stay in X."""
    _, X = structured_mesh3d(nnodes_x, nnodes_y, nnodes_z)
    nid = lambda c: c[..., 0] + c[..., 1] * nnodes_x + c[..., 2] * nnodes_x * nnodes_y  # noqa: E731
    px, py, pz = (nnodes_x - 1) // 2, (nnodes_y - 1) // 2, (nnodes_z - 1) // 2
    k, j, i = np.meshgrid(np.arange(pz), np.arange(py), np.arange(px), indexing="ij")
    origin = 2 * np.stack([i.ravel(), j.ravel(), k.ravel()], axis=1)  # (patches,3)
    if etype == "tet":
        corners = origin[:, None, None, :] + TET_CORNERS[None]  # (patches,6,4,3)
        pairs = [(0, 1), (1, 2), (0, 2), (0, 3), (1, 3), (2, 3)]
        mids = np.stack([(corners[:, :, a] + corners[:, :, b]) // 2 for a, b in pairs], axis=2)
        conn = nid(np.concatenate([corners, mids], axis=2)).reshape(-1, 10)
    elif etype == "brick20":
        conn = nid(origin[:, None, :] + (BRICK20_NODES[None].astype(np.int64) + 1))
    else:
        raise ValueError(f"unknown element_type: {etype}")
    return conn.astype(np.int64), X


def elasticity_problem3d(nnodes_x, nnodes_y, nnodes_z):
    """x=0 face fully fixed; point loads [0,-1,0] on the nodes (k, j=0, i=nx-1)  (ref: pyfem.py:2737-2755)."""
    face = (np.arange(nnodes_y)[None, :] * nnodes_x + np.arange(nnodes_z)[:, None] * nnodes_x * nnodes_y).ravel()
    dof_fixed = (3 * face[:, None] + np.arange(3)[None, :]).ravel()
    nodal_force = {int(nnodes_x - 1 + k * nnodes_x * nnodes_y): [0.0, -1.0, 0.0] for k in range(nnodes_z)}
    return dof_fixed, nodal_force


def helmholtz_input3d(nnodes_x, nnodes_y, nnodes_z):
    """0.95 in the octant i<nx/2, j<ny/2, k<nz/2, else 1e-3  (ref: pyfem.py:2758-2773)."""
    i = np.tile(np.arange(nnodes_x), nnodes_y * nnodes_z)
    j = np.tile(np.repeat(np.arange(nnodes_y), nnodes_x), nnodes_z)
    k = np.repeat(np.arange(nnodes_z), nnodes_x * nnodes_y)
    return np.where((i < nnodes_x / 2) & (j < nnodes_y / 2) & (k < nnodes_z / 2), 0.95, 1e-3)


def poisson_fixed3d(nnodes_x, nnodes_y, nnodes_z):
    """Nodes of the x=0 face, k-major (ref: pyfem.py:2728-2735)."""
    return (np.arange(nnodes_y)[None, :] * nnodes_x + np.arange(nnodes_z)[:, None] * nnodes_x * nnodes_y).ravel()


def elasticity_problem(nnodes_x, nnodes_y):
    """x=0 column fully fixed, point load [0,-1] at node (j=0, i=nx-1)
    (ref: pyfem.py:2737-2755)."""
    col0 = np.arange(nnodes_y) * nnodes_x
    dof_fixed = np.stack([2 * col0, 2 * col0 + 1], axis=1).ravel()
    nodal_force = {nnodes_x - 1: [0.0, -1.0]}
    return dof_fixed, nodal_force


def helmholtz_input(nnodes_x, nnodes_y):
    """x = 0.95 where i < nnodes_x/2 and j < nnodes_y/2, else 1e-3 (ref: pyfem.py:2758-2773)."""
    i = np.tile(np.arange(nnodes_x), nnodes_y)
    j = np.repeat(np.arange(nnodes_y), nnodes_x)
    return np.where((i < nnodes_x / 2) & (j < nnodes_y / 2), 0.95, 1e-3)


# --------------------------------------------------------------------------
# element kernels                                   ref: utils.py:154-298
# --------------------------------------------------------------------------
def create_conn_dof(conn, ndof):
    """conn_dof[e, a*ndof+k] = ndof*conn[e,a]+k   (ref: utils.py:268-298)."""
    if ndof == 1:
        return conn
    E, P = conn.shape
    out = np.empty((E, P * ndof), dtype=np.int64)
    for k in range(ndof):
        out[:, k::ndof] = ndof * conn + k
    return out


def geometry(X, conn, dN):
    """Jq = dx_j/dxi_k (ref: utils.py:172-185), detJq (utils.py:189-200; closed form
    here, the reference calls LAPACK det), Ngrad = dN . invJ (utils.py:225-264; 2x2 and 3x3 closed forms)."""
    Xe = X[conn]  # (E,P,ndims)                               ref: utils.py:155-168
    Jq = np.einsum("qlk,ilj->iqjk", dN, Xe)
    inv = np.empty_like(Jq)
    if X.shape[1] == 2:
        detJ = Jq[..., 0, 0] * Jq[..., 1, 1] - Jq[..., 0, 1] * Jq[..., 1, 0]
        inv[..., 0, 0] = Jq[..., 1, 1] / detJ
        inv[..., 0, 1] = -Jq[..., 0, 1] / detJ
        inv[..., 1, 0] = -Jq[..., 1, 0] / detJ
        inv[..., 1, 1] = Jq[..., 0, 0] / detJ
    else:
        J = Jq
        c00 = J[..., 1, 1] * J[..., 2, 2] - J[..., 1, 2] * J[..., 2, 1]
        c01 = J[..., 1, 0] * J[..., 2, 2] - J[..., 1, 2] * J[..., 2, 0]
        c02 = J[..., 1, 0] * J[..., 2, 1] - J[..., 1, 1] * J[..., 2, 0]
        detJ = J[..., 0, 0] * c00 - J[..., 0, 1] * c01 + J[..., 0, 2] * c02
        inv[..., 0, 0] = c00 / detJ
        inv[..., 0, 1] = -(J[..., 0, 1] * J[..., 2, 2] - J[..., 0, 2] * J[..., 2, 1]) / detJ
        inv[..., 0, 2] = (J[..., 0, 1] * J[..., 1, 2] - J[..., 0, 2] * J[..., 1, 1]) / detJ
        inv[..., 1, 0] = -c01 / detJ
        inv[..., 1, 1] = (J[..., 0, 0] * J[..., 2, 2] - J[..., 0, 2] * J[..., 2, 0]) / detJ
        inv[..., 1, 2] = -(J[..., 0, 0] * J[..., 1, 2] - J[..., 0, 2] * J[..., 1, 0]) / detJ
        inv[..., 2, 0] = c02 / detJ
        inv[..., 2, 1] = -(J[..., 0, 0] * J[..., 2, 1] - J[..., 0, 1] * J[..., 2, 0]) / detJ
        inv[..., 2, 2] = (J[..., 0, 0] * J[..., 1, 1] - J[..., 0, 1] * J[..., 1, 0]) / detJ
    Ngrad = np.einsum("jkm,ijml->ijkl", dN, inv)
    return detJ, Ngrad


def elasticity3d_C0(E=10.0, nu=0.3):
    """6x6 isotropic, strain order xx,yy,zz,xy,yz,xz with engineering shear (ref: pyfem.py:1752-1757)."""
    C0 = np.zeros((6, 6))
    C0[0, 0] = C0[1, 1] = C0[2, 2] = 1 - nu
    C0[0, 1] = C0[0, 2] = C0[1, 0] = C0[1, 2] = C0[2, 0] = C0[2, 1] = nu
    C0[3, 3] = C0[4, 4] = C0[5, 5] = 0.5 - nu
    return C0 * E / ((1 + nu) * (1 - 2 * nu))


def _Bmat(Ngrad):
    """Strain-displacement matrix (ref: pyfem.py:1976-2013)."""
    nE, Q, P, nd = Ngrad.shape
    if nd == 2:
        Be = np.zeros((nE, Q, 3, 2 * P))
        Be[:, :, 0, 0::2] = Ngrad[..., 0]
        Be[:, :, 1, 1::2] = Ngrad[..., 1]
        Be[:, :, 2, 0::2] = Ngrad[..., 1]
        Be[:, :, 2, 1::2] = Ngrad[..., 0]
        return Be
    Be = np.zeros((nE, Q, 6, 3 * P))
    Nx, Ny, Nz = Ngrad[..., 0], Ngrad[..., 1], Ngrad[..., 2]
    Be[:, :, 0, 0::3] = Nx
    Be[:, :, 1, 1::3] = Ny
    Be[:, :, 2, 2::3] = Nz
    Be[:, :, 3, 0::3] = Ny
    Be[:, :, 3, 1::3] = Nx
    Be[:, :, 4, 1::3] = Nz
    Be[:, :, 4, 2::3] = Ny
    Be[:, :, 5, 0::3] = Nz
    Be[:, :, 5, 2::3] = Nx
    return Be


def plane_stress_C0(E=10.0, nu=0.3):
    """ref: pyfem.py:1746-1750."""
    return E / (1.0 - nu**2) * np.array([[1.0, nu, 0.0], [nu, 1.0, 0.0], [0.0, 0.0, 0.5 * (1.0 - nu)]])


def ramp(rho_q, p):
    """RAMP penalisation rho/(1+p(1-rho))  (ref: pyfem.py:1923-1945)."""
    return rho_q / (1.0 + p * (1.0 - rho_q))


def elasticity_Ke(X, conn, etype, rho=None, E=10.0, nu=0.3, p=0.0):
    """Element stiffness (E,D,D):  sum_q detJ w Cq  B^T C0 B
    (ref: pyfem.py:1976-2068; einsum string of :2016-2027)."""
    _, wq = quadrature(etype)
    N, dN = shape_tables(etype)
    detJ, Ngrad = geometry(X, conn, dN)
    nE, P = conn.shape
    Q = wq.size
    rho = np.ones(X.shape[0]) if rho is None else np.asarray(rho, dtype=float)
    rho_q = np.einsum("jl,il->ij", N, rho[conn])  # ref: utils.py:204-221
    Cq = ramp(rho_q, p)
    Be = _Bmat(Ngrad)
    C0 = plane_stress_C0(E, nu) if X.shape[1] == 2 else elasticity3d_C0(E, nu)
    return np.einsum("iq,q,iqnj,iq,nm,iqmk->ijk", detJ, wq, Be, Cq, C0, Be, optimize=True)


def elasticity_K_dv_sens(X, conn, etype, rho, phi, psi, E=10.0, nu=0.3, p=0.0):
    """d(phi^T K(rho) psi)/d rho  (nodal), ref: pyfem.py:1873-1920 with the RAMP derivative
    (1+p)/(1+p(1-rho_q))^2 N[q,o] of :1948-1973; einsum strings of :1900 and :1912."""
    _, wq = quadrature(etype)
    N, dN = shape_tables(etype)
    detJ, Ngrad = geometry(X, conn, dN)
    nE, P = conn.shape
    Q = wq.size
    rho = np.asarray(rho, dtype=float)
    rho_q = np.einsum("jl,il->ij", N, rho[conn])
    Cqderiv = np.einsum("ql,iq->iql", N, (1 + p) / (1 + p * (1 - rho_q)) ** 2)
    Be = _Bmat(Ngrad)
    C0 = plane_stress_C0(E, nu) if X.shape[1] == 2 else elasticity3d_C0(E, nu)
    Ke_deriv = np.einsum("iq,q,iqnj,iqo,nm,iqmk->ijko", detJ, wq, Be, Cqderiv, C0, Be, optimize=True)
    conn_dof = create_conn_dof(conn, X.shape[1])
    inner = np.einsum("ij,ik,ijko->io", phi[conn_dof], psi[conn_dof], Ke_deriv)
    out = np.zeros(X.shape[0])
    for i in range(P):
        np.add.at(out, conn[:, i], inner[:, i])
    return out


def poisson_Ke(X, conn, etype, rho=None, p=0.0):
    """Ke = sum_q kappa_q detJ w gradN gradN^T, kappa_q = RAMP(rho_q); `kappa0` of the reference is stored
    but never applied (ref: pyfem.py:1176-1187, 1279-1301)."""
    _, wq = quadrature(etype)
    N, dN = shape_tables(etype)
    detJ, Ngrad = geometry(X, conn, dN)
    rho = np.ones(X.shape[0]) if rho is None else np.asarray(rho, dtype=float)
    kappa_q = ramp(np.einsum("jl,il->ij", N, rho[conn]), p)
    return np.einsum("iq,iq,q,iqjl,iqkl->ijk", kappa_q, detJ, wq, Ngrad, Ngrad, optimize=True)


def poisson_rhs(X, conn, etype, gfunc):
    """Element rhs and its scatter WITH the reference's quirks (ref: pyfem.py:1131-1136, 861-875):
    the einsum indexes N[j,k] with j on the quadrature axis (right only because N is symmetric for the
    2x2 rule on quads) and _assemble_rhs loops over range(nquads) columns, not range(P)."""
    _, wq = quadrature(etype)
    N, dN = shape_tables(etype)
    detJ, _ = geometry(X, conn, dN)
    Xq = np.einsum("jl,ilk->ijk", N, X[conn])  # utils.py:204-221
    fun = gfunc(Xq)
    rhs_e = np.zeros((conn.shape[0], conn.shape[1]))
    rhs_e[...] = np.einsum("ik,k,jk,ik->ij", detJ, wq, N, fun, optimize=True)
    rhs = np.zeros(X.shape[0])
    for n in range(wq.size):
        np.add.at(rhs, conn[:, n], rhs_e[:, n])
    return rhs


# --------------------------------------------------------------------------
# NonlinearPoisson2D:  -div( h(x)(1 + u^2) grad u ) = g                      ref: pyfem.py:1332-1665
# --------------------------------------------------------------------------
def nlpoisson_gfun(Xq):
    """ref: pyfem.py:1427-1447."""
    x, y = Xq[..., 0], Xq[..., 1]
    return 1e4 * x * (1.0 - x) * (1.0 - 2.0 * x) * y * (1.0 - y) * (1.0 - 2.0 * y)


def nlpoisson_hfun(xdv, Xq):
    """h = 1 + 4y(1-y) sum_k xdv[k] C(n-1,k) (1-x)^(n-1-k) x^k   (ref: pyfem.py:1449-1470)."""
    from scipy import special

    x, y = Xq[..., 0], Xq[..., 1]
    n = len(xdv)
    h = np.ones_like(x)
    for k in range(n):
        h = h + xdv[k] * special.binom(n - 1, k) * (1.0 - x) ** (n - 1 - k) * x**k * 4.0 * y * (1.0 - y)
    return h


def nlpoisson_element(X, conn, xdv, u, etype="quad"):
    """Element tangent Ke (E,P,P) and residual re (E,P) at the state u  (ref: pyfem.py:1472-1601):
    Ke[j,i] = sum_q w detJ h (1+uq^2) gradN_j.gradN_i + sum_q 2 w detJ h uq (gradN_j . grad u) N[q,i]
    re[j]   = sum_q w detJ h (1+uq^2) gradN_j . grad u - sum_q w detJ g N[q,j]."""
    _, wq = quadrature(etype)
    N, dN = shape_tables(etype)
    detJ, Ngrad = geometry(X, conn, dN)
    Xq = np.einsum("jl,ilk->ijk", N, X[conn])
    ue = np.asarray(u, dtype=float)[conn]
    uq = np.einsum("jl,il->ij", N, ue)
    h, g = nlpoisson_hfun(xdv, Xq), nlpoisson_gfun(Xq)
    Ke = np.einsum("nq,q,nqjl,nqkl->njk", detJ * h * (1.0 + uq**2), wq, Ngrad, Ngrad)
    Ke += np.einsum("nq,nqjl,nqkl,nk,qi->nji", 2.0 * detJ * h * uq * wq, Ngrad, Ngrad, ue, N)
    re = np.einsum("nq,nqjl,nqkl,nk->nj", detJ * h * (1.0 + uq**2) * wq, Ngrad, Ngrad, ue)
    re -= np.dot(detJ * wq * g, N)
    return Ke, re


def nlpoisson_residual(X, conn, xdv, u, etype="quad"):
    """Global residual ("compute_rhs", ref: pyfem.py:1374-1388 with the range(nquads) scatter of :861-875)."""
    _, re = nlpoisson_element(X, conn, xdv, u, etype)
    rhs = np.zeros(X.shape[0])
    for n in range(quadrature(etype)[1].size):
        np.add.at(rhs, conn[:, n], re[:, n])
    return rhs


def nlpoisson_solve(X, conn, dof_fixed, xdv, etype="quad", tol=1e-10, atol=1e-12, max_iter=10):
    """Newton iteration of Assembler.solve_nonlinear with the direct solver (ref: pyfem.py:2319-2355): rows of the
    fixed dofs replaced by the identity (enforce_symmetric_K=False), u -= K^-1 res."""
    u = np.zeros(X.shape[0])
    res0 = None
    for k in range(max_iter):
        Ke, _ = nlpoisson_element(X, conn, xdv, u, etype)
        K = assemble_csr(Ke, conn)
        res = nlpoisson_residual(X, conn, xdv, u, etype)
        K, res = apply_dirichlet_bcs(K, res, dof_fixed, None, False)
        rn = np.sqrt(res.dot(res))
        if k == 0:
            res0 = rn
        elif rn < tol * res0 or rn < atol:
            break
        u = u - spsolve(K.tocsc(), res)
    return u


def poisson_K_dv_sens(X, conn, etype, rho, phi, psi, p=0.0):
    """d(phi^T K psi)/d rho for LinearPoisson (ref: pyfem.py:1240-1276, 1303-1329)."""
    _, wq = quadrature(etype)
    N, dN = shape_tables(etype)
    detJ, Ngrad = geometry(X, conn, dN)
    rho_q = np.einsum("jl,il->ij", N, np.asarray(rho, dtype=float)[conn])
    kd = np.einsum("ql,iq->iql", N, (1 + p) / (1 + p * (1 - rho_q)) ** 2)
    Ke_deriv = np.einsum("iqo,iq,q,iqjl,iqkl->ijko", kd, detJ, wq, Ngrad, Ngrad, optimize=True)
    inner = np.einsum("ij,ik,ijko->io", phi[conn], psi[conn], Ke_deriv)
    out = np.zeros(X.shape[0])
    for i in range(conn.shape[1]):
        np.add.at(out, conn[:, i], inner[:, i])
    return out


def helmholtz_Ke_Re(X, conn, etype, r0):
    """Re = sum_q detJ w N N^T ;  Ke = r0^2 sum_q detJ w gradN.gradN^T + Re
    (ref: pyfem.py:2127-2177)."""
    _, wq = quadrature(etype)
    N, dN = shape_tables(etype)
    detJ, Ngrad = geometry(X, conn, dN)
    Re = np.einsum("iq,q,qj,qk->ijk", detJ, wq, N, N, optimize=True)
    Ke = np.einsum("iq,q,iqjl,iqkl->ijk", detJ * r0**2, wq, Ngrad, Ngrad, optimize=True)
    return Ke + Re, Re


# --------------------------------------------------------------------------
# global scatter, BCs, solve                        ref: pyfem.py:781-931, 2299-2423
# --------------------------------------------------------------------------
def assemble_csr(Ke, conn_dof):
    """COO triplets in (e, a, b) order, duplicates summed, columns sorted, explicit
    zeros kept  (ref: pyfem.py:838-858 pattern, :921-931 coo->csr)."""
    D = conn_dof.shape[1]
    nz_i = np.repeat(conn_dof, D, axis=1).ravel()
    nz_j = np.tile(conn_dof, (1, D)).ravel()
    return sparse.coo_matrix((Ke.ravel(), (nz_i, nz_j))).tocsr()


def apply_dirichlet_bcs(K, rhs, dof_fixed, dof_fixed_vals=None, enforce_symmetric_K=True):
    """[Krr 0; 0 I][ur;u0] = [fr - Krb u0; u0]; exact zeros then dropped
    (ref: pyfem.py:781-835).  Returns new (K, rhs); inputs are not modified."""
    K = K.tocsr().copy()
    rhs = np.array(rhs, dtype=float)
    n = K.shape[0]
    dof_fixed = np.asarray(dof_fixed, dtype=np.int64)
    fixed = np.zeros(n, dtype=bool)
    fixed[dof_fixed] = True
    free = ~fixed
    Krb = K[free][:, dof_fixed]
    diag = K.diagonal()
    keep_row = sparse.diags(free.astype(float))
    K = keep_row @ K
    if enforce_symmetric_K:
        K = K @ keep_row
    diag[dof_fixed] = 1.0
    K = K.tocsr()
    K.setdiag(diag)
    K.eliminate_zeros()
    if dof_fixed_vals is None:
        rhs[dof_fixed] = 0.0
    else:
        vals = np.asarray(dof_fixed_vals, dtype=float)
        rhs[dof_fixed] = vals
        if enforce_symmetric_K:
            rhs[free] -= Krb.dot(vals)
    return K.tocsr(), rhs


def elasticity_rhs(ndof, nodal_force):
    """Point loads by assignment (ref: pyfem.py:1761-1768)."""
    rhs = np.zeros(ndof)
    for node, f in nodal_force.items():
        d = len(f)
        rhs[d * node : d * node + d] = f
    return rhs


def jacobi_pcg(K, b, rtol=1e-8, maxiter=None, x0=None):
    """Jacobi-preconditioned CG, stop at ||r||_2 <= rtol ||b||_2 (atol 0)
    -- the tolerance convention of ref: pyfem.py:2418 (`cg(K, rhs, tol=1e-8, atol=0.0)`).
    The reference preconditions with pyamg SA-AMG, which is not installed anywhere
    this code runs: the iterative solve is "parity unpinned" at that boundary and is
    pinned instead against the direct solve (same u to the stated tolerance)."""
    d = K.diagonal()
    M = LinearOperator(K.shape, matvec=lambda v: v / d, dtype=float)
    it = [0]

    def cb(_):
        it[0] += 1

    u, info = cg(K, b, rtol=rtol, atol=0.0, M=M, maxiter=maxiter, x0=x0, callback=cb)
    if info:
        raise RuntimeError(f"cg failed with code {info}")
    return u, it[0]


def solve_elasticity(X, conn, etype, dof_fixed, nodal_force, method="direct", rho=None, E=10.0, nu=0.3, p=0.0,
                     rtol=1e-8):
    """Assembler.solve for LinearElasticity (ref: pyfem.py:2299-2317)."""
    nd = X.shape[1]
    conn_dof = create_conn_dof(conn, nd)
    K = assemble_csr(elasticity_Ke(X, conn, etype, rho, E, nu, p), conn_dof)
    rhs = elasticity_rhs(nd * X.shape[0], nodal_force)
    Kbc, rhs = apply_dirichlet_bcs(K, rhs, dof_fixed, None, True)
    if method == "direct":
        return spsolve(Kbc, rhs), K, Kbc, rhs
    u, _ = jacobi_pcg(Kbc, rhs, rtol=rtol)
    return u, K, Kbc, rhs


class HelmholtzFilter:
    """K = r0^2 A + M, R = M;  apply: K rho = R x;  apply_gradient: R^T K^-1 g
    (ref: pyfem.py:2085-2120).  The reference solves with pyamg ruge_stuben to tol 1e-8;
    here a sparse LU (what tests/ref_helmholtz.py:106 of the reference uses)."""

    def __init__(self, r0, X, conn, etype="quad"):
        from scipy.sparse.linalg import factorized

        Ke, Re = helmholtz_Ke_Re(X, conn, etype, r0)
        self.K = assemble_csr(Ke, conn)
        self.R = assemble_csr(Re, conn)
        self._lu = factorized(self.K.tocsc())

    def apply(self, x):
        return self._lu(self.R.dot(x))

    def apply_gradient(self, g):
        return self.R.T.dot(self._lu(g))
