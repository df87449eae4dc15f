"""Host-side mirror of the reference's `pyfem` module surface for the accelerated path.

Same class names, constructor signatures and method names as the reference
(aaronyicongfu/pyfem_gpu_testflight, pyfem.py) so its tests and examples read unchanged:

    creator = pyfem.ProblemCreator(nnodes_x=32, nnodes_y=32)
    conn, X, dof_fixed, nodal_force = creator.create_linear_elasticity_problem()
    quadrature = pyfem.QuadratureBilinear2D(); basis = pyfem.BasisBilinear2D(quadrature)
    model = pyfem.LinearElasticity(X, conn, dof_fixed, None, nodal_force, quadrature, basis)
    u = pyfem.Assembler(model).solve(method="cg")

What runs where: everything with an element or nonzero axis (quadrature, scatter, BCs, SpMV,
CG) runs in hand-written HIP kernels through the C ABI (include/pyfem_hip.h); this file only
holds the host logic the reference has around those calls.  There is no numpy fallback: without
the library or a GPU the constructors raise.

Scope: 4-node quads, 3-node triangles and 8-node blocks as hard-wired kernels
(LinearElasticity with RAMP density and sensitivities, LinearPoisson, Helmholtz filter, Assembler.solve);
10-node tetrahedra, 20-node bricks and any other 3-D basis through the table-driven kernels (LinearElasticity);
complex-step densities (the reference's derivative tests) as a real/imaginary pair of device matrices.
A2DWrapper (external a2d library) raises NotImplementedError here.
"""
import numpy as np

from . import _lib
from .device import DeviceCSR, HipContext
from .utils import time_this

# tolerance used by the device PCG wherever the reference asks scipy/pyamg for tol=1e-8
# (pyfem.py:1827, 2107, 2418).  Tighter than the reference so that ||u - u_direct||_inf stays
# below 1e-8 ||u||_inf (BASELINE.json north_star) on ill-conditioned large meshes: measured against a
# manufactured solution on the 2048x1024 mesh with a RAMP density field in [1e-3, 1], p = 5
# (tests/test_gpu_scale.py), 1e-10 leaves 2.6e-8, 1e-11 one more AMG-PCG iteration and 2e-9.
DEFAULT_RTOL = 1e-11
# method="direct": relative residual the device PCG is driven to in place of the reference's sparse LU
DIRECT_RTOL = 1e-13
DIRECT_ACCEPT = 1e-10   # true relative residual up to which method="direct" accepts (with a warning) what PCG reached
DIRECT_QUIET = 1e-11    # ... and below which it says nothing: a sparse LU of these systems sits at 1e-13 .. 1e-12 itself
DIRECT_BURST = 500      # iterations per refinement round of method="direct" with the AMG preconditioner
DIRECT_ROUNDS = 4

# preconditioner of the device PCG: "amg" (aggregation AMG V-cycle, the counterpart of the reference's
# pyamg preconditioners, pyfem.py:2393-2400 and :2098; positive or negative definite K -- only diagonal entries of
# both signs among the coupled dofs make pf_cg fall back to Jacobi) or "jacobi"
DEFAULT_PRECOND = "amg"


def _vdot(a, b):
    """rhs . u of the compliance functionals (pyfem.py:1068, 1832) without BLAS: a multi-threaded ddot wakes one
    thread per logical CPU, and on a CPU-quota'd GPU box those spinning threads get the whole process throttled for
    tens of milliseconds -- longer than the solve the number belongs to."""
    s = np.add.reduce(np.multiply(a, b))
    return complex(s) if np.iscomplexobj(s) else float(s)


class DirectSolveError(RuntimeError):
    """method="direct" gave up: the message names the iterations spent, the true residual reached and the host
    alternative, and is passed through _solve_linear_system unchanged (a RuntimeError, as pyfem.py:2422 raises)."""


_PRECOND = {"none": _lib.PF_PRECOND_NONE, "jacobi": _lib.PF_PRECOND_JACOBI, "amg": _lib.PF_PRECOND_AMG}


# ------------------------------------------------------------------------------------------
# tables (host copies for API parity; the kernels carry the same constants at compile time)
# ------------------------------------------------------------------------------------------
class QuadratureBase:
    def __init__(self, pts, weights):
        self.pts = np.asarray(pts, dtype=float)
        self.weights = np.asarray(weights, dtype=float)
        self.nquads = self.pts.shape[0]

    def get_nquads(self):
        return self.nquads

    def get_pt(self, idx=None):
        return self.pts[idx] if idx else self.pts

    def get_weight(self, idx=None):
        return self.weights[idx] if idx else self.weights


class QuadratureTriangle2D(QuadratureBase):
    """ref: pyfem.py:68-80 -- one point (1/3, 1/3), weight 1 * 0.5."""

    def __init__(self):
        super().__init__([[1.0 / 3, 1.0 / 3]], [0.5])


class QuadratureBilinear2D(QuadratureBase):
    """ref: pyfem.py:83-94 -- 2x2 Gauss, order (-,-),(+,-),(+,+),(-,+), unit weights."""

    def __init__(self):
        g = 1.0 / np.sqrt(3.0)
        super().__init__([[-g, -g], [g, -g], [g, g], [-g, g]], [1.0, 1.0, 1.0, 1.0])


class QuadratureBlock3D(QuadratureBase):
    """ref: pyfem.py:97-112 -- 2x2x2 Gauss, x slowest / z fastest, unit weights."""

    def __init__(self):
        g = 1.0 / np.sqrt(3.0)
        super().__init__([[sx * g, sy * g, sz * g] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)], [1.0] * 8)


class QuadratureTetrahedron5Point(QuadratureBase):
    """ref: pyfem.py:115-134 -- centroid (weight -4/5) and the four points (1/2, 1/6, 1/6) ... (1/6, 1/6, 1/6) (9/20)."""

    def __init__(self):
        pts = np.full((5, 3), 1.0 / 6)
        pts[0] = 0.25
        pts[1, 0] = pts[2, 1] = pts[3, 2] = 0.5
        super().__init__(pts, [-4 / 5, 9 / 20, 9 / 20, 9 / 20, 9 / 20])


class QuadratureBrick333Point(QuadratureBase):
    """ref: pyfem.py:137-172 -- 3x3x3 Gauss points at 0, +-sqrt(3/5), x fastest; weights are the reference's
    10-digit products of 5/9 and 8/9, by the number of zero coordinates."""

    _W = np.array([0.1714677641, 0.2743484225, 0.4389574760, 0.7023319616])

    def __init__(self):
        k, j, i = np.meshgrid([-1, 0, 1], [-1, 0, 1], [-1, 0, 1], indexing="ij")
        ijk = np.stack([i.ravel(), j.ravel(), k.ravel()], axis=1)
        super().__init__(ijk * np.sqrt(3.0 / 5.0), self._W[(ijk == 0).sum(axis=1)])


class BasisBase:
    def __init__(self, ndims, nnodes_per_elem, quadrature):
        self.ndims = ndims
        self.nnodes_per_elem = nnodes_per_elem
        self.quadrature = quadrature
        self.nquads = quadrature.get_nquads()
        self.N = None
        self.Nderiv = None

    def eval_shape_fun(self):
        if self.N is None:
            self.N = np.array([self._eval_shape_fun_on_quad_pt(q) for q in self.quadrature.get_pt()])
        return self.N

    def eval_shape_fun_deriv(self):
        if self.Nderiv is None:
            self.Nderiv = np.array([self._eval_shape_deriv_on_quad_pt(q) for q in self.quadrature.get_pt()]).reshape(
                (self.nquads, self.nnodes_per_elem, self.ndims))
        return self.Nderiv


class BasisBilinear2D(BasisBase):
    """ref: pyfem.py:253-284."""

    def __init__(self, quadrature):
        super().__init__(2, 4, quadrature)

    def _eval_shape_fun_on_quad_pt(self, q):
        return [0.25 * (1 - q[0]) * (1 - q[1]), 0.25 * (1 + q[0]) * (1 - q[1]),
                0.25 * (1 + q[0]) * (1 + q[1]), 0.25 * (1 - q[0]) * (1 + q[1])]

    def _eval_shape_deriv_on_quad_pt(self, q):
        return [-0.25 * (1 - q[1]), -0.25 * (1 - q[0]), 0.25 * (1 - q[1]), -0.25 * (1 + q[0]),
                0.25 * (1 + q[1]), 0.25 * (1 + q[0]), -0.25 * (1 + q[1]), 0.25 * (1 - q[0])]


class BasisBlock3D(BasisBase):
    """ref: pyfem.py:287-338 -- trilinear 8-node block, nodes CCW on z=-1 then z=+1."""

    _S = np.array([[-1, -1, -1], [1, -1, -1], [1, 1, -1], [-1, 1, -1], [-1, -1, 1], [1, -1, 1], [1, 1, 1], [-1, 1, 1]],
                  dtype=float)

    def __init__(self, quadrature):
        super().__init__(3, 8, quadrature)

    def _eval_shape_fun_on_quad_pt(self, q):
        f = 1.0 + self._S * np.asarray(q)[None, :]
        return list(0.125 * f[:, 0] * f[:, 1] * f[:, 2])

    def _eval_shape_deriv_on_quad_pt(self, q):
        f = 1.0 + self._S * np.asarray(q)[None, :]
        d = 0.125 * self._S * np.stack([f[:, 1] * f[:, 2], f[:, 0] * f[:, 2], f[:, 0] * f[:, 1]], axis=1)
        return list(d.ravel())


class BasisTetrahedron10node(BasisBase):
    """ref: pyfem.py:380-445 -- quadratic tetrahedron: corners t_i (2 t_i - 1) in the volume coordinates
    t = (1 - x - y - z, x, y, z), then 4 t_a t_b on the edges (0,1), (1,2), (2,0), (0,3), (1,3), (2,3)."""

    _EDGES = ((0, 1), (1, 2), (2, 0), (0, 3), (1, 3), (2, 3))
    _DT = np.array([[-1.0, -1.0, -1.0], [1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0]])

    def __init__(self, quadrature):
        super().__init__(3, 10, quadrature)

    def _eval_shape_fun_on_quad_pt(self, q):
        t = np.array([1.0 - q[0] - q[1] - q[2], q[0], q[1], q[2]])
        return np.concatenate([t * (2.0 * t - 1.0), [4.0 * t[a] * t[b] for a, b in self._EDGES]])

    def _eval_shape_deriv_on_quad_pt(self, q):
        t = np.array([1.0 - q[0] - q[1] - q[2], q[0], q[1], q[2]])
        d = np.empty((10, 3))
        d[:4] = (4.0 * t - 1.0)[:, None] * self._DT
        for n, (a, b) in enumerate(self._EDGES):
            d[4 + n] = 4.0 * (t[a] * self._DT[b] + t[b] * self._DT[a])
        return d.ravel()


class BasisBrick20Nodes(BasisBase):
    """ref: pyfem.py:448-632 -- 20-node serendipity brick: corners CCW on z=-1 then z=+1, mid-edge nodes of the two
    faces, then of the vertical edges."""

    _NODES = np.array(
        [[-1, -1, -1], [1, -1, -1], [1, 1, -1], [-1, 1, -1], [-1, -1, 1], [1, -1, 1], [1, 1, 1], [-1, 1, 1],
         [0, -1, -1], [1, 0, -1], [0, 1, -1], [-1, 0, -1], [0, -1, 1], [1, 0, 1], [0, 1, 1], [-1, 0, 1],
         [-1, -1, 0], [1, -1, 0], [1, 1, 0], [-1, 1, 0]], dtype=float)

    def __init__(self, quadrature):
        self.nodecoords = self._NODES.copy()
        super().__init__(3, 20, quadrature)

    def _eval_shape_fun_on_quad_pt(self, q):
        c, q = self.nodecoords, np.asarray(q, dtype=float)
        f = 1.0 + c * q[None, :]
        N = np.empty(20)
        N[:8] = 0.125 * f[:8, 0] * f[:8, 1] * f[:8, 2] * (c[:8] @ q - 2.0)
        for n in range(8, 20):
            z = int(np.flatnonzero(c[n] == 0.0)[0])
            N[n] = 0.25 * (1.0 - q[z] ** 2) * f[n, (z + 1) % 3] * f[n, (z + 2) % 3]
        return N

    def _eval_shape_deriv_on_quad_pt(self, q):
        c, q = self.nodecoords, np.asarray(q, dtype=float)
        f = 1.0 + c * q[None, :]
        s = c[:8] @ q
        d = np.empty((20, 3))
        for k in range(3):
            o1, o2 = (k + 1) % 3, (k + 2) % 3
            d[:8, k] = 0.125 * c[:8, k] * f[:8, o1] * f[:8, o2] * (s + q[k] * c[:8, k] - 1.0)
        for n in range(8, 20):
            z = int(np.flatnonzero(c[n] == 0.0)[0])
            o1, o2 = (z + 1) % 3, (z + 2) % 3
            e = 1.0 - q[z] ** 2
            d[n, z] = -0.5 * q[z] * f[n, o1] * f[n, o2]
            d[n, o1] = 0.25 * e * c[n, o1] * f[n, o2]
            d[n, o2] = 0.25 * e * f[n, o1] * c[n, o2]
        return d.ravel()


class BasisTriangle2D(BasisBase):
    """ref: pyfem.py:341-377."""

    def __init__(self, quadrature):
        super().__init__(2, 3, quadrature)

    def _eval_shape_fun_on_quad_pt(self, q):
        return [q[0], q[1], 1 - q[0] - q[1]]

    def _eval_shape_deriv_on_quad_pt(self, q):
        return [1.0, 0.0, 0.0, 1.0, -1.0, -1.0]


def _check_tables(quadrature, basis):
    """Returns True for a (quadrature, basis) pair that runs on the table-driven 3-D kernels, False for one of the
    three hard-wired pairs; anything else is refused."""
    if (isinstance(quadrature, QuadratureBilinear2D) and isinstance(basis, BasisBilinear2D)) or (
            isinstance(quadrature, QuadratureTriangle2D) and isinstance(basis, BasisTriangle2D)) or (
            isinstance(quadrature, QuadratureBlock3D) and isinstance(basis, BasisBlock3D)):
        return False
    if basis.ndims == 3 and basis.nnodes_per_elem != 8 and 4 <= basis.nnodes_per_elem <= 32 and quadrature.get_nquads() <= 64:
        return True  # tet10, brick20, or any other 3-D basis given by its tables (pf_*_tab)
    raise NotImplementedError(
        "accelerated path: (QuadratureBilinear2D, BasisBilinear2D), (QuadratureTriangle2D, BasisTriangle2D) and "
        "(QuadratureBlock3D, BasisBlock3D) are hard-wired HIP kernels; other 3-D bases with 4..32 nodes and at most "
        "64 quadrature points run on the table-driven kernels; nothing else is implemented")


# ------------------------------------------------------------------------------------------
# complex-step densities
# ------------------------------------------------------------------------------------------
def _split_complex(rho):
    """(real part, imaginary part or None) of a nodal density given as numpy array or torch tensor."""
    if hasattr(rho, "is_complex"):  # torch
        return (rho.real, rho.imag) if rho.is_complex() else (rho, None)
    if np.iscomplexobj(rho):
        rho = np.asarray(rho)
        return rho.real.copy(), rho.imag.copy()
    return rho, None


class ComplexStepCSR:
    """K(rho + i drho) to first order in drho: `real` = K(rho), `imag` = dK/drho [drho], both device matrices.

    The reference differentiates its assembly by a complex step of 1e-30 (tests/test_elasticity.py:96-98, 127-128;
    test_linear_poisson.py:82-83, 112-113; test_elements.py:50-51): `compute_jacobian(rho + 1j * p * h)` returns a complex
    scipy matrix whose imaginary part is exact to first order -- second-order terms (1e-60) are below anything fp64
    keeps next to the real part.  The kernels are fp64-real, so the two parts are assembled separately: the real part
    by the usual kernel, the imaginary part by the table-driven kernel with the coefficient RAMP'(rho_q) (N drho)_q
    (pf_assemble_tab).  Products, Dirichlet conditions and solves act on the pair the way complex arithmetic would."""

    def __init__(self, real, imag):
        self.real, self.imag = real, imag
        self.hctx = real.hctx
        self.shape = real.shape
        self.dtype = np.dtype(np.complex128)

    def dot(self, v):
        v = np.asarray(v)
        if np.iscomplexobj(v):
            vr, vi = np.ascontiguousarray(v.real), np.ascontiguousarray(v.imag)
            return (self.real.dot(vr) - self.imag.dot(vi)) + 1j * (self.real.dot(vi) + self.imag.dot(vr))
        return self.real.dot(v) + 1j * self.imag.dot(v)

    __matmul__ = dot

    def to_scipy(self):
        from scipy import sparse

        indptr, indices = self.hctx.csr_structure()
        data = self.hctx.vals_to_csr(self.real.vals).cpu().numpy() + 1j * self.hctx.vals_to_csr(self.imag.vals).cpu().numpy()
        K = sparse.csr_matrix((data, indices.copy(), indptr.copy()), shape=self.shape)
        if self.real._eliminate:
            K.eliminate_zeros()
        return K

    tocsr = to_scipy


# ------------------------------------------------------------------------------------------
# models
# ------------------------------------------------------------------------------------------
class ModelBase:
    """ref: pyfem.py:634-931.  Holds the host copies the reference exposes (X, conn, dof arrays, rhs)
    and a HipContext with the device mesh + symbolic pattern."""

    def __init__(self, ndof_per_node, X, conn, dof_fixed, dof_fixed_vals, quadrature, basis, device=None):
        self.tabular = _check_tables(quadrature, basis)
        if self.tabular and type(self).__name__ != "LinearElasticity":
            raise NotImplementedError("accelerated path: the table-driven 3-D elements (tet10, brick20) are built for "
                                      "LinearElasticity only")
        self.ndof_per_node = ndof_per_node
        self.X = np.array(X, dtype=float)
        self.conn = np.array(conn, dtype=int)
        self.dof_fixed = np.array(dof_fixed, dtype=int)
        self.dof_fixed_vals = None if dof_fixed_vals is None else np.array(dof_fixed_vals, dtype=float)
        self.quadrature = quadrature
        self.basis = basis
        self.nelems, self.nnodes_per_elem = self.conn.shape
        self.nnodes, self.ndims = self.X.shape
        self.nquads = quadrature.get_nquads()
        if self.nnodes_per_elem != basis.nnodes_per_elem:
            raise ValueError("conn does not match the basis' nodes per element")
        self.nodes = np.arange(self.nnodes)
        # conn range is asserted on device by pf_mesh_set (ref: pyfem.py:680-681)
        tables = (basis.eval_shape_fun(), basis.eval_shape_fun_deriv(), quadrature.get_weight())
        self.hctx = HipContext(self.X, self.conn, ndof_per_node, device=device, tables=tables, tabular=self.tabular)
        self.ndof = self.nnodes * ndof_per_node
        self.dof = np.arange(self.ndof)
        self.dof_each_node = self.dof.reshape(self.nnodes, ndof_per_node) if ndof_per_node > 1 else self.nodes
        self._fixed_mask_host = np.zeros(self.ndof, dtype=np.uint8)
        self._fixed_mask_host[self.dof_fixed] = 1
        self.dof_free = np.flatnonzero(self._fixed_mask_host == 0)
        torch = self.hctx.torch
        self._fixed_mask = self.hctx.to_device(self._fixed_mask_host, torch.uint8)
        if self.dof_fixed_vals is None:
            self._fixed_vals = None
        else:
            fv = np.zeros(self.ndof)
            fv[self.dof_fixed] = self.dof_fixed_vals
            self._fixed_vals = self.hctx.to_device(fv, torch.float64)
        self.rhs = np.zeros(self.ndof)

    @property
    def conn_dof(self):
        """ref: utils.py:268-298 (host view for readers; the kernels derive dofs from conn)."""
        d = self.ndof_per_node
        if d == 1:
            return self.conn
        out = np.empty((self.nelems, self.nnodes_per_elem * d), dtype=int)
        for k in range(d):
            out[:, k::d] = d * self.conn + k
        return out

    @time_this
    def apply_dirichlet_bcs(self, K, rhs, enforce_symmetric_K=True):
        """ref: pyfem.py:781-835.  K: DeviceCSR from compute_jacobian (modified in place, like the
        reference mutates its argument); rhs: numpy array (modified in place) or device tensor."""
        if isinstance(K, ComplexStepCSR):
            return self._apply_dirichlet_bcs_complex(K, rhs, enforce_symmetric_K)
        if not isinstance(K, DeviceCSR) or K.hctx is None:
            # a host matrix (the reference takes any scipy CSR, pyfem.py:781): e.g. `K + K2` formed with scipy.  Its
            # entries must lie inside the model's pattern; the values are uploaded once and the result is device-backed.
            K = DeviceCSR(self.hctx, self.hctx.upload_csr(K))
        torch = self.hctx.torch
        on_device = isinstance(rhs, torch.Tensor)
        rhs_d = rhs if on_device else self.hctx.to_device(rhs, torch.float64)
        if not enforce_symmetric_K:
            # rows only: K stays nonsymmetric and goes to BiCGStab, whose AMG is built from a symmetric positive definite
            # companion on the same pattern -- the tangent's SPD part (NonlinearPoisson2D) or K itself before the
            # conditions -- with the conditions applied symmetrically
            comp = getattr(K, "companion", None)
            if comp is None:
                comp = K.copy()
            self.hctx.apply_dirichlet(comp.vals, rhs_d.clone(), self._fixed_mask, self._fixed_vals, True)
            K.companion = comp
        self.hctx.apply_dirichlet(K.vals, rhs_d, self._fixed_mask, self._fixed_vals, enforce_symmetric_K)
        K.invalidate()
        K._eliminate = True  # the reference ends with K.eliminate_zeros() (pyfem.py:826)
        if on_device:
            return K, rhs_d
        rhs[:] = rhs_d.cpu().numpy()
        return K, rhs

    def _apply_dirichlet_bcs_complex(self, K, rhs, enforce_symmetric_K):
        """The complex matrix after pyfem.py:781-835: unit (real) diagonal on the fixed dofs, so the imaginary part gets
        zero rows, columns and diagonal there; rhs[free] -= K[free, fixed] vals picks up an imaginary part when values
        are prescribed."""
        torch = self.hctx.torch
        rhs_r = self.hctx.to_device(np.asarray(rhs, dtype=float), torch.float64)
        rhs_i = self.hctx.zeros(self.ndof)
        self.hctx.apply_dirichlet(K.real.vals, rhs_r, self._fixed_mask, self._fixed_vals, enforce_symmetric_K)
        self.hctx.apply_dirichlet(K.imag.vals, rhs_i, self._fixed_mask, self._fixed_vals, enforce_symmetric_K,
                                  diag_scale=self.hctx.zeros(self.ndof))
        rhs_i[self._fixed_mask.bool()] = 0.0
        for part in (K.real, K.imag):
            part.invalidate()
            part._eliminate = True
        rhs[:] = rhs_r.cpu().numpy()
        ri = rhs_i.cpu().numpy()
        return K, (rhs + 1j * ri if np.any(ri) else rhs)


class LinearPoisson(ModelBase):
    """ref: pyfem.py:934-1329 (2-D).  -k lap(u) = g with RAMP-penalised conductivity; `kappa0` is stored
    and, as in the reference (:977 vs :1300), never applied.  compute_rhs evaluates the Python `gfunc` on
    the host at the quadrature points the device computes and integrates on the device."""

    @time_this
    def __init__(self, X, conn, dof_fixed, dof_fixed_vals, quadrature, basis, gfunc, kappa0=1.0, p=0.0, device=None):
        super().__init__(1, X, conn, dof_fixed, dof_fixed_vals, quadrature, basis, device=device)
        self.gfunc = gfunc
        self.fun_vals = None
        self.kappa0 = kappa0
        self.p = float(p)
        self._last_rho = None
        self._Xq = None

    def _rho_device(self, rho):
        torch = self.hctx.torch
        if isinstance(rho, torch.Tensor):
            return rho.to(device=self.hctx.device, dtype=torch.float64).contiguous()
        if not hasattr(rho, "__len__"):
            if rho == 1.0:
                return None
            rho = np.ones(self.nnodes) * rho
        return self.hctx.to_device(np.asarray(rho).astype(float), torch.float64)

    @time_this
    def compute_rhs(self):
        """ref: pyfem.py:997-1004, 1138-1173."""
        torch = self.hctx.torch
        if self._Xq is None:  # quadrature-point coordinates: fixed with the mesh (the reference keeps them in self.Xq)
            self._Xq = self.hctx.quad_points().cpu().numpy()
        Xq = self._Xq
        # the reference assigns gfunc's result into an (nelems, nquads) array (pyfem.py:1126-1128), so a scalar or a
        # broadcastable result is fine (examples/heat_conduction.py passes `lambda x: source_temp`)
        self.fun_vals = np.empty(Xq.shape[:2])
        self.fun_vals[...] = self.gfunc(Xq)
        gq = self.hctx.to_device(np.ascontiguousarray(self.fun_vals), torch.float64)
        self.rhs[:] = self.hctx.poisson_rhs(gq).cpu().numpy()
        return self.rhs

    @time_this
    def compute_jacobian(self, rho=1.0, out=None):
        """ref: pyfem.py:1006-1030.  A complex density (complex step) returns a ComplexStepCSR."""
        rho, drho = _split_complex(rho)
        rho_d = self._rho_device(rho)
        self._last_rho = rho_d
        if drho is not None:
            if rho_d is None:
                rho_d = self.hctx.to_device(np.ones(self.nnodes), self.hctx.torch.float64)
            Ki = self.hctx.assemble_tab(_lib.PF_TAB_POISSON, 0.0, 0.0, self.p, rho_d, self._rho_device(drho))
            return ComplexStepCSR(DeviceCSR(self.hctx, self.hctx.assemble_poisson(self.p, rho_d)), DeviceCSR(self.hctx, Ki))
        vals = self.hctx.assemble_poisson(self.p, rho_d, out=None if out is None else out.vals)
        if out is not None:
            out.invalidate()
            out._eliminate = False
            return out
        return DeviceCSR(self.hctx, vals)

    @time_this
    def _einsum_element_jacobian(self):
        """Element matrices (nelems, P, P) on the device for the last density; the name (and its profiler key,
        read by examples/SciTech2023/performance/performance_test.py:103-113) is the reference's, pyfem.py:1176.
        Synchronises, so that the recorded time is the kernel's."""
        Ke = self.hctx.element_matrices_poisson(self.p, self._last_rho)
        self.hctx.torch.cuda.synchronize()
        return Ke

    @time_this
    def _compute_element_jacobian(self, Ke):
        """ref: pyfem.py:1189-1217 -- fills the caller's array in place (performance_test.py:52)."""
        Ke[...] = self._einsum_element_jacobian().cpu().numpy()

    @property
    def Ke_mat(self):
        return self._einsum_element_jacobian().cpu().numpy()

    @time_this
    def compliance(self, rho, solver="cg", weighted=True):
        """ref: pyfem.py:1033-1073."""
        assert solver == "direct" or solver == "cg" or solver == "gmres"
        K = self.compute_jacobian(rho)
        rhs = self.compute_rhs()
        K, rhs = self.apply_dirichlet_bcs(K, rhs, enforce_symmetric_K=True)
        asm = Assembler(self)
        u = asm._solve_linear_system(K, rhs, solver)
        self.last_iters = asm.last_iters  # PCG iterations of this solve
        c = _vdot(rhs, u) if weighted else np.sum(u) / len(u)
        return c, u

    @time_this
    def compliance_grad(self, rho, u, weighted=True):
        """ref: pyfem.py:1076-1104."""
        if weighted:
            psi = u
        else:
            K = self.compute_jacobian(rho)
            rhs = np.ones(len(u))
            K, rhs = self.apply_dirichlet_bcs(K, rhs, enforce_symmetric_K=True)
            psi = Assembler(self)._solve_linear_system(K, rhs, "cg") / len(u)
        return -self._compute_K_dv_sens(rho, psi, u)

    def volume(self, rho):
        return rho.sum() / self.nnodes

    def volume_grad(self, rho):
        return np.ones(self.nnodes) / self.nnodes

    @time_this
    def _compute_K_dv_sens(self, rho, phi, psi):
        """ref: pyfem.py:1240-1276, fused (no Ke_deriv tensor)."""
        torch = self.hctx.torch
        if not hasattr(rho, "__len__"):
            rho = np.ones(self.nnodes) * rho
        rho_d = self._rho_device(rho)
        if rho_d is None:
            rho_d = self.hctx.to_device(np.ones(self.nnodes), torch.float64)
        phi_d = self.hctx.to_device(np.asarray(phi, dtype=float), torch.float64)
        psi_d = self.hctx.to_device(np.asarray(psi, dtype=float), torch.float64)
        return self.hctx.sens_poisson(self.p, rho_d, phi_d, psi_d).cpu().numpy()


class LinearElasticity(ModelBase):
    """ref: pyfem.py:1667-2068 (2-D plane stress branch)."""

    @time_this
    def __init__(self, X, conn, dof_fixed, dof_fixed_vals, nodal_force, quadrature, basis, E=10.0, nu=0.3, p=0.0,
                 device=None):
        X = np.asarray(X)
        super().__init__(X.shape[1], X, conn, dof_fixed, dof_fixed_vals, quadrature, basis, device=device)
        self.nodal_force = nodal_force
        self.E, self.nu, self.p = float(E), float(nu), float(p)
        if self.ndims == 2:  # ref: pyfem.py:1746-1750
            self.C0 = E * np.array([[1.0, nu, 0.0], [nu, 1.0, 0.0], [0.0, 0.0, 0.5 * (1.0 - nu)]])
            self.C0 *= 1.0 / (1.0 - nu**2)
        else:  # ref: pyfem.py:1752-1757
            self.C0 = np.zeros((6, 6))
            self.C0[0, 0] = self.C0[1, 1] = self.C0[2, 2] = 1 - nu
            self.C0[0, 1] = self.C0[0, 2] = self.C0[1, 0] = self.C0[1, 2] = self.C0[2, 0] = self.C0[2, 1] = nu
            self.C0[3, 3] = self.C0[4, 4] = self.C0[5, 5] = 0.5 - nu
            self.C0 *= E / ((1 + nu) * (1 - 2 * nu))
        self.scatter_mode = _lib.PF_SCATTER_GATHER
        self._last_rho = None

    @time_this
    def compute_rhs(self):
        """ref: pyfem.py:1761-1768 -- point loads by assignment into the persistent model.rhs."""
        nodes = list(self.nodal_force.keys())
        self.rhs[self.dof_each_node[nodes].flatten()] = np.array(list(self.nodal_force.values())).flatten()
        return self.rhs

    def compute_rhs_device(self):
        """The right-hand side a fresh model's compute_rhs() returns -- zeros with the point loads assigned
        (pyfem.py:1761-1768) -- as a NEW device tensor, without the 8 bytes per dof of host upload compute_rhs() costs the
        device path (0.6 of the 0.9 ms that assembly + conditions take at configuration 5's size).  Built once from
        nodal_force and cloned; what host code wrote into model.rhs by hand is not in it (compute_rhs() keeps that
        reference behaviour)."""
        torch = self.hctx.torch
        if getattr(self, "_rhs_dev", None) is None:
            t = torch.zeros(self.ndof, dtype=torch.float64, device=self.hctx.device)
            nodes = list(self.nodal_force.keys())
            if nodes:
                idx = torch.as_tensor(np.asarray(self.dof_each_node[nodes]).flatten(), dtype=torch.int64, device=self.hctx.device)
                t[idx] = torch.as_tensor(np.array(list(self.nodal_force.values()), dtype=float).flatten(), device=self.hctx.device)
            self._rhs_dev = t
        return self._rhs_dev.clone()

    def _rho_device(self, rho):
        torch = self.hctx.torch
        if isinstance(rho, torch.Tensor):
            return rho.to(device=self.hctx.device, dtype=torch.float64).contiguous()
        if not hasattr(rho, "__len__"):
            if rho == 1.0:
                return None  # rho == 1 => Cq == 1 for every p (pyfem.py:1780-1781, 1944)
            rho = np.ones(self.nnodes) * rho
        rho = np.asarray(rho)
        if rho.shape != (self.nnodes,):
            raise ValueError("rho must be nodal: shape (nnodes,)")
        return self.hctx.to_device(rho.astype(float), torch.float64)

    @time_this
    def compute_jacobian(self, rho=1.0, out=None):
        """ref: pyfem.py:1771-1795.  Returns a DeviceCSR; Ke never exists in memory.  A complex density (the complex
        step of the reference's derivative tests, pyfem.py:1786-1788) returns a ComplexStepCSR."""
        rho, drho = _split_complex(rho)
        rho_d = self._rho_device(rho)
        self._last_rho = rho_d
        if drho is not None:
            if rho_d is None:
                rho_d = self.hctx.to_device(np.ones(self.nnodes), self.hctx.torch.float64)
            Kr = self.hctx.assemble_elasticity(self.E, self.nu, self.p, rho_d, self.scatter_mode)
            Ki = self.hctx.assemble_tab(_lib.PF_TAB_ELASTICITY, self.E, self.nu, self.p, rho_d, self._rho_device(drho))
            return ComplexStepCSR(DeviceCSR(self.hctx, Kr), DeviceCSR(self.hctx, Ki))
        vals = self.hctx.assemble_elasticity(self.E, self.nu, self.p, rho_d, self.scatter_mode,
                                             out=None if out is None else out.vals)
        if out is not None:
            out.invalidate()
            out._eliminate = False
            return out
        return DeviceCSR(self.hctx, vals)

    @time_this
    def _einsum_element_jacobian(self):
        """Element matrices (nelems, D, D) on the device for the last density; the name (and its profiler key,
        read by examples/SciTech2023/performance/performance_test.py:103-113) is the reference's, pyfem.py:2016.
        Synchronises, so that the recorded time is the kernel's."""
        Ke = self.hctx.element_matrices_elasticity(self.E, self.nu, self.p, self._last_rho)
        self.hctx.torch.cuda.synchronize()
        return Ke

    @time_this
    def _compute_element_jacobian(self, Ke):
        """ref: pyfem.py:2030-2068 -- fills the caller's array in place (performance_test.py:52)."""
        Ke[...] = self._einsum_element_jacobian().cpu().numpy()

    @property
    def Ke_mat(self):
        """(nelems, D, D) element matrices for the last density (ref attribute: pyfem.py:717-723)."""
        return self._einsum_element_jacobian().cpu().numpy()

    @time_this
    def compliance(self, rho, solver="cg"):
        """ref: pyfem.py:1798-1834."""
        assert solver == "direct" or solver == "cg" or solver == "gmres"
        K = self.compute_jacobian(rho)
        torch = self.hctx.torch
        if isinstance(rho, torch.Tensor):  # a device density (the optimisation loop keeps its fields in HBM): nothing
            rhs = self.compute_rhs_device()  # crosses PCIe -- (c, u) with u a device tensor
            K, rhs = self.apply_dirichlet_bcs(K, rhs, enforce_symmetric_K=True)
            asm = getattr(self, "_asm", None)
            if asm is None:  # kept: the amg_lag bookkeeping lives in it (set model.amg_lag = N to opt in, see Assembler)
                asm = self._asm = Assembler(self)
            asm.rtol = getattr(self, "rtol", asm.rtol)
            asm.amg_lag = getattr(self, "amg_lag", asm.amg_lag)
            u = asm._solve_linear_system(K, rhs, solver)
            self.last_iters = asm.last_iters
            return float(rhs.dot(u)), u
        rhs = self.compute_rhs()
        K, rhs = self.apply_dirichlet_bcs(K, rhs, enforce_symmetric_K=True)
        asm = Assembler(self)
        u = asm._solve_linear_system(K, rhs, solver)
        self.last_iters = asm.last_iters  # PCG iterations of this solve
        return _vdot(rhs, u), u

    def volume(self, rho):
        """ref: pyfem.py:1851-1859."""
        return rho.sum() / self.nnodes

    def volume_grad(self, rho):
        """ref: pyfem.py:1862-1870 (device density in -> device tensor out)."""
        torch = self.hctx.torch
        if isinstance(rho, torch.Tensor):
            return torch.full((self.nnodes,), 1.0 / self.nnodes, dtype=torch.float64, device=rho.device)
        return np.ones(self.nnodes) / self.nnodes

    @time_this
    def compliance_grad(self, rho, u):
        """ref: pyfem.py:1837-1848."""
        return -self._compute_K_dv_sens(rho, u, u)

    @time_this
    def _compute_K_dv_sens(self, rho, phi, psi):
        """d(phi^T K psi)/d rho, ref: pyfem.py:1873-1920 -- one fused kernel, Ke_deriv is never formed.
        numpy in -> numpy out; device tensors in -> device tensor out."""
        torch = self.hctx.torch
        on_device = isinstance(rho, torch.Tensor)
        if not on_device and not hasattr(rho, "__len__"):
            rho = np.ones(self.nnodes) * rho
        rho_d = self._rho_device(rho)
        if rho_d is None:
            rho_d = self.hctx.to_device(np.ones(self.nnodes), torch.float64)
        phi_d = phi if isinstance(phi, torch.Tensor) else self.hctx.to_device(np.asarray(phi, dtype=float), torch.float64)
        psi_d = psi if isinstance(psi, torch.Tensor) else self.hctx.to_device(np.asarray(psi, dtype=float), torch.float64)
        out = self.hctx.sens_elasticity(self.E, self.nu, self.p, rho_d, phi_d, psi_d)
        return out if on_device else out.cpu().numpy()


class Helmholtz(ModelBase):
    """ref: pyfem.py:2069-2177.  K = r0^2 A + M and R = M assembled once on device; apply() solves
    K rho = R x with the device PCG (AMG V-cycle by default; the reference uses pyamg ruge_stuben
    V-cycles to tol 1e-8).  K never changes, so the AMG numeric setup is made once and reused."""

    @time_this
    def __init__(self, r0, X, conn, quadrature, basis, device=None):
        super().__init__(1, X, conn, [], None, quadrature, basis, device=device)
        self.r0 = float(r0)
        Kv, Rv = self.hctx.assemble_helmholtz(self.r0)
        self.K = DeviceCSR(self.hctx, Kv)
        self.R = DeviceCSR(self.hctx, Rv)
        self.RT = self.R  # the mass matrix is symmetric (ref keeps R.transpose(), pyfem.py:2096)
        self.rtol = DEFAULT_RTOL
        self.precond = DEFAULT_PRECOND
        self.last_iters = 0
        self.reuse_forward = True   # apply(x) of an unchanged device tensor returns the kept result (see apply)
        self._last_apply = None
        self.last_reused = False

    def _solve(self, b_dev):
        pc = _PRECOND[self.precond]
        if pc == _lib.PF_PRECOND_AMG:
            pc = _lib.PF_PRECOND_AMG_REUSE  # the hierarchy checks that it was set up for these K values
        x, it, _ = self.hctx.cg(self.K.vals, b_dev, rtol=self.rtol, precond=pc)
        self.last_iters = it
        return x

    @time_this
    def apply(self, x):
        """ref: pyfem.py:2102-2107.  The optimisation callbacks filter the SAME design twice per iteration (evalObjCon and
        evalObjConGradient both start with filtr.apply(x), topo.py:52, 71): for a device tensor the last result is kept
        and returned again (a fresh clone) while the tensor object and its version counter are unchanged -- any in-place
        write to x bumps the counter, a new tensor is a new object; numpy inputs are always solved."""
        torch = self.hctx.torch
        on_device = isinstance(x, torch.Tensor)
        if on_device and self.reuse_forward:
            c = self._last_apply
            if c is not None and c[0] is x and c[1] == x._version and c[3] == (self.rtol, self.precond):
                self.last_iters, self.last_reused = c[4], True
                return c[2].clone()
        xd = x if on_device else self.hctx.to_device(x, torch.float64)
        rho = self._solve(self.hctx.spmv(self.R.vals, xd))
        self.last_reused = False
        if on_device and self.reuse_forward:
            self._last_apply = (x, x._version, rho.clone(), (self.rtol, self.precond), self.last_iters)
        return rho if on_device else rho.cpu().numpy()

    @time_this
    def apply_gradient(self, gradrho):
        """ref: pyfem.py:2110-2115."""
        torch = self.hctx.torch
        on_device = isinstance(gradrho, torch.Tensor)
        gd = gradrho if on_device else self.hctx.to_device(gradrho, torch.float64)
        gx = self.hctx.spmv(self.RT.vals, self._solve(gd))
        return gx if on_device else gx.cpu().numpy()

    def apply_gradient_pair(self, g1, g2):
        """(apply_gradient(g1), apply_gradient(g2)) -- the two gradient filters evalObjConGradient issues back to back
        (topo.py:73-74; the reference solves them one after the other, pyfem.py:2110-2115) -- as ONE AMG-PCG over the
        pair (pf_cg2): the two vectors sit next to each other per node, every product, smoother sweep and transfer
        streams its operator once for both and the launch-bound coarse levels are walked once.  Each vector keeps its own
        recurrence and stopping test, so each result is what apply_gradient gives alone (to rounding).  Device tensors
        in, device tensors out; numpy inputs, and preconditioners other than AMG, are solved one after the other."""
        torch = self.hctx.torch
        if not (isinstance(g1, torch.Tensor) and isinstance(g2, torch.Tensor)) or _PRECOND[self.precond] != _lib.PF_PRECOND_AMG:
            return self.apply_gradient(g1), self.apply_gradient(g2)
        try:
            sol, its, _ = self.hctx.cg2(self.K.vals, torch.stack((g1, g2), dim=1), rtol=self.rtol)
        except ValueError:  # PF_E_UNSUPPORTED: a hierarchy the two-vector cycle does not cover
            return self.apply_gradient(g1), self.apply_gradient(g2)
        self.last_iters = max(its)
        self.last_pair_iters = its
        sol = sol.t().contiguous()  # (2, n): one vector after the other for R^T
        return self.hctx.spmv(self.RT.vals, sol[0]), self.hctx.spmv(self.RT.vals, sol[1])

    @time_this
    def compute_rhs(self, x):
        """ref: pyfem.py:2118-2120."""
        self.rhs[:] = self.R.dot(np.asarray(x, dtype=float))
        return self.rhs

    def compute_jacobian(self):
        return self.K

    @time_this
    def _einsum_element_jacobian(self):
        """(Ke, Re) on the device, each (nelems, P, P); reference names pyfem.py:2127 / :2134 (one kernel here)."""
        KR = self.hctx.element_matrices_helmholtz(self.r0)
        self.hctx.torch.cuda.synchronize()
        return KR

    @time_this
    def _compute_element_jacobian_and_rhs(self, Ke, Re):
        """ref: pyfem.py:2139-2177 -- fills the caller's arrays in place."""
        Kd, Rd = self._einsum_element_jacobian()
        Ke[...] = Kd.cpu().numpy()
        Re[...] = Rd.cpu().numpy()

    @property
    def Ke_mat(self):
        return self._einsum_element_jacobian()[0].cpu().numpy()

    @property
    def Re(self):
        return self._einsum_element_jacobian()[1].cpu().numpy()


def _not_on_path(name, where):
    class _Stub:
        def __init__(self, *a, **k):
            raise NotImplementedError(f"{name} ({where}) is outside the accelerated hot path (SURVEY.md section 2)")

    _Stub.__name__ = name
    return _Stub


A2DWrapper = _not_on_path("A2DWrapper", "pyfem.py:2180-2283 (needs the external a2d library)")


class NonlinearPoisson2D(ModelBase):
    """ref: pyfem.py:1332-1665.   -div( h(x) (1 + u^2) grad u ) = g  with the reference's h (Bernstein design variables
    xdv) and g.  compute_jacobian(xdv, u) is the Newton tangent (nonsymmetric), compute_rhs(xdv, u) the residual --
    the reference's names; Assembler.solve_nonlinear drives them.  Both are evaluated by the table-driven kernels
    (pf_nlpoisson_tab) from the basis tables."""

    @time_this
    def __init__(self, X, conn, dof_fixed, dof_fixed_vals, quadrature, basis, device=None):
        if np.asarray(X).shape[1] != 2:
            raise ValueError("NonlinearPoisson2D is two-dimensional")
        super().__init__(1, X, conn, dof_fixed, dof_fixed_vals, quadrature, basis, device=device)
        self._state = None

    def _device_state(self, xdv, u):
        from scipy import special

        torch = self.hctx.torch
        xdv = np.asarray(xdv, dtype=float)
        n = xdv.shape[0]
        ck = self.hctx.to_device(xdv * special.binom(n - 1, np.arange(n)), torch.float64)  # pyfem.py:1464-1468
        u_d = u if isinstance(u, torch.Tensor) else self.hctx.to_device(np.asarray(u, dtype=float), torch.float64)
        self._state = (u_d, ck)
        return u_d, ck

    @time_this
    def compute_jacobian(self, xdv, u):
        """ref: pyfem.py:1390-1405, 1523-1562.  The returned matrix carries its symmetric positive definite part as
        `companion`: the preconditioner of the nonsymmetric solve is built from it."""
        u_d, ck = self._device_state(xdv, u)
        K = DeviceCSR(self.hctx, self.hctx.nlpoisson(_lib.PF_NL_TANGENT, u_d, ck))
        K.companion = DeviceCSR(self.hctx, self.hctx.nlpoisson(_lib.PF_NL_TANGENT_SPD, u_d, ck))
        return K

    @time_this
    def compute_rhs(self, xdv, u):
        """ref: pyfem.py:1374-1388, 1472-1521 -- the residual at u (the reference's name for it is rhs)."""
        u_d, ck = self._device_state(xdv, u)
        self.rhs[...] = self.hctx.nlpoisson(_lib.PF_NL_RESIDUAL, u_d, ck).cpu().numpy()
        return self.rhs

    @property
    def Ke_mat(self):
        """Element tangents (nelems, P, P) at the last state (ref attribute: pyfem.py:717-723)."""
        if self._state is None:
            raise AttributeError("Ke_mat: call compute_jacobian(xdv, u) first")
        return self.hctx.nlpoisson(_lib.PF_NL_ELEMENT_TANGENT, *self._state).view(self.nelems, self.nnodes_per_elem,
                                                                                 self.nnodes_per_elem).cpu().numpy()

    def compute_residual(self, u):
        """ref: pyfem.py:1407-1425 -- unusable in the reference (it writes into self.Re, which is None)."""
        raise TypeError("'NoneType' object does not support item assignment (NonlinearPoisson2D.compute_residual is "
                        "dead code in the reference: self.Re is never allocated, pyfem.py:1372, 1419)")


# ------------------------------------------------------------------------------------------
class Assembler:
    """ref: pyfem.py:2286-2423."""

    @time_this
    def __init__(self, model):
        self.model = model
        self.rtol = DEFAULT_RTOL
        self.precond = DEFAULT_PRECOND
        self.last_iters = None
        self.last_relres = None  # relative residual of the last solve (method="direct": the TRUE ||b - K x|| / ||b||)
        self.direct_strict = False  # True: method="direct" raises above DIRECT_QUIET instead of warning up to DIRECT_ACCEPT
        # amg_lag = N > 0 (opt-in; "cg" / "gmres" with the AMG preconditioner): the numeric setup of the hierarchy is kept
        # for up to N further solves although the matrix has changed (PF_PRECOND_AMG_STALE) -- the V-cycle of an earlier
        # matrix is still a valid preconditioner, PCG converges to the same solution -- and redone as soon as a solve
        # needs 30 % more iterations than the one right after the last setup.  For loops whose matrix moves slowly (late
        # topology-optimisation steps), where the setup is a quarter of the solve; 0 (default): a setup per matrix.
        self.amg_lag = 0
        self.amg_setups = 0  # numeric setups requested so far (amg_lag bookkeeping; tests read it)

    @time_this
    def solve(self, method="gmres"):
        """ref: pyfem.py:2299-2317."""
        assert method == "direct" or method == "cg" or method == "gmres"
        K = self.model.compute_jacobian()
        rhs = self.model.compute_rhs()
        K, rhs = self.model.apply_dirichlet_bcs(K, rhs, enforce_symmetric_K=True)
        return self._solve_linear_system(K, rhs, method)

    @time_this
    def _solve_linear_system(self, K, rhs, method):
        """ref: pyfem.py:2403-2423.  All three methods run on the device.
        "cg" and "gmres": PCG with the smoothed-aggregation AMG V-cycle (self.precond = "amg", default) or Jacobi, to
                  self.rtol.  K is SPD after symmetric BC elimination, so CG reaches the same solution GMRES does.
        "direct": the reference calls scipy spsolve (pyfem.py:2414).  No factorisation is built on the GPU: the same
                  PCG is driven to DIRECT_RTOL (1e-13 in ||r||/||b||, the level an LU residual sits at in fp64) with
                  restarts (_direct_by_refinement); what it does on stagnation and on indefinite matrices is stated
                  there.  Host code that wants the factorisation itself can still hand the returned matrix to scipy
                  (`spsolve(K, rhs)`: DeviceCSR is a csr_matrix)."""
        rtol = DIRECT_RTOL if method == "direct" else self.rtol
        if isinstance(K, ComplexStepCSR):  # (Kr + i Ki)(ur + i ui) = br + i bi, to first order in the imaginary parts
            rhs = np.asarray(rhs)
            ur = self._solve_linear_system(K.real, np.ascontiguousarray(rhs.real, dtype=float), method)
            bi = -K.imag.dot(ur)
            if np.iscomplexobj(rhs):
                bi = bi + rhs.imag
            return ur + 1j * self._solve_linear_system(K.real, bi, method)
        hctx = K.hctx
        torch = hctx.torch
        on_device = isinstance(rhs, torch.Tensor)
        b = rhs if on_device else hctx.to_device(rhs, torch.float64)
        try:
            comp = getattr(K, "companion", None)
            if comp is not None:  # nonsymmetric (rows-only conditions, Newton tangent): BiCGStab, AMG of the companion
                x, it, rr = hctx.bicgstab(K.vals, b, pvals=comp.vals, rtol=rtol, precond=_PRECOND[self.precond])
            elif method == "direct":
                x, it, rr = self._direct_by_refinement(K, b)
            else:
                pc, lag = _PRECOND[self.precond], None
                if pc == _lib.PF_PRECOND_AMG:
                    lag = getattr(hctx, "_amg_lag_state", None)  # kept on the context: that is what owns the hierarchy
                    if self.amg_lag > 0 and lag is not None and lag["age"] < self.amg_lag and not lag["refresh"]:
                        pc = _lib.PF_PRECOND_AMG_STALE
                x, it, rr = hctx.cg(K.vals, b, rtol=rtol, precond=pc)
                if pc == _lib.PF_PRECOND_AMG_STALE:
                    lag["age"] += 1
                    lag["refresh"] = it > 1.3 * lag["fresh_iters"] + 1
                elif pc == _lib.PF_PRECOND_AMG:
                    self.amg_setups += 1
                    hctx._amg_lag_state = {"fresh_iters": it, "age": 0, "refresh": False}
        except (_lib.PyfemHipError, DirectSolveError):
            raise  # a HIP / argument error is not a convergence failure, and method="direct"'s message is final
        except RuntimeError as e:  # PF_E_NOCONV: "<solver> failed with code <iters>" (pyfem.py:2421-2422)
            raise RuntimeError(f"{method} failed with code {str(e).rsplit(' ', 1)[-1]}") from None
        self.last_iters, self.last_relres = it, rr
        return x if on_device else x.cpu().numpy()

    def _direct_by_refinement(self, K, b, burst=None):
        """method="direct" on a symmetric K (pyfem.py:2414 calls spsolve): PCG to DIRECT_RTOL, restarted from the
        iterate reached whenever a run of DIRECT_BURST iterations ends above it.  A restart recomputes r = b - K x, so
        the recurrence residual cannot drift away from the true one (iterative refinement); ill-conditioned systems
        (RAMP densities near the lower bound) whose recurrence stalls get there in two or three rounds.  Acceptance is
        on the TRUE residual ||b - K x|| / ||b||: <= DIRECT_RTOL silently, <= DIRECT_ACCEPT with a RuntimeWarning that
        names the level reached; above that -- an indefinite or singular matrix, which spsolve would have factorised
        -- RuntimeError("direct failed with code <iterations>: ...") naming the residual and the host alternative."""
        import warnings

        hctx = K.hctx
        torch = hctx.torch
        if burst is None:
            burst = DIRECT_BURST if self.precond == "amg" else 0  # Jacobi by request: scipy's default cap (10 n)
        x, total, rr = None, 0, float("inf")
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            for _ in range(DIRECT_ROUNDS):
                x, it, rr = hctx.cg(K.vals, b, x0=x, rtol=DIRECT_RTOL, maxiter=burst, precond=_PRECOND[self.precond],
                                    allow_maxiter=True)
                total += it
                # it == 0: the TRUE residual of the restart was already below the target; a run that converged by its
                # recurrence is restarted once more, because the recurrence drifts from b - K x on ill-conditioned K
                if it == 0 or not np.isfinite(rr):
                    break
        for w in {str(c.message) for c in caught}:
            warnings.warn(w, RuntimeWarning, stacklevel=4)
        bn = float(torch.linalg.vector_norm(b))
        true_rr = float(torch.linalg.vector_norm(b - hctx.spmv(K.vals, x))) / bn if bn > 0 else 0.0
        self.last_iters, self.last_relres = total, true_rr  # readable also when the solve is refused below
        if self.direct_strict and not true_rr <= DIRECT_QUIET:
            raise DirectSolveError(
                f"direct failed with code {total}: strict mode accepts ||b - K x|| / ||b|| <= {DIRECT_QUIET:g}, PCG "
                f"with refinement reached ||b - K x|| / ||b|| = {true_rr:.3e}; "
                "scipy.sparse.linalg.spsolve(K, rhs) factorises the returned matrix on the host")
        if not np.isfinite(true_rr) or true_rr > DIRECT_ACCEPT:
            raise DirectSolveError(
                f"direct failed with code {total}: no factorisation is built on the device and PCG stopped at "
                f"||b - K x|| / ||b|| = {true_rr:.3e} (the matrix may be indefinite or singular); "
                "scipy.sparse.linalg.spsolve(K, rhs) factorises the returned matrix on the host")
        if true_rr > DIRECT_QUIET:
            warnings.warn(f'method="direct": PCG with refinement reached ||b - K x|| / ||b|| = {true_rr:.3e} '
                          f"(target {DIRECT_RTOL:g}) after {total} iterations; accepted", RuntimeWarning, stacklevel=4)
        return x, total, true_rr

    @time_this
    def solve_nonlinear(self, method="gmres", xdv=None, u0=None, tol=1e-10, atol=1e-12, max_iter=10):
        """ref: pyfem.py:2319-2355 -- Newton iteration on model.compute_jacobian(xdv, u) / compute_rhs(xdv, u) with the
        conditions applied to the rows only; each update is a nonsymmetric device solve (BiCGStab, see
        _solve_linear_system)."""
        assert method == "direct" or method == "cg" or method == "gmres"
        u = np.zeros(self.model.nnodes) if u0 is None else u0
        res_norm_init = None
        for k in range(max_iter):
            K = self.model.compute_jacobian(xdv, u)
            res = self.model.compute_rhs(xdv, u)
            self.model.apply_dirichlet_bcs(K, res, enforce_symmetric_K=False)
            res_norm = np.sqrt(_vdot(res, res))
            print("pyfem", "{0:5d} {1:25.15e}".format(k, res_norm))
            if k == 0:
                res_norm_init = res_norm
            elif res_norm < tol * res_norm_init or res_norm < atol:
                break
            u -= self._solve_linear_system(K, res, method)
        return u

    def plot(self, u, ax, **kwargs):
        """Filled contours of a nodal scalar on a 2-D mesh (ref: pyfem.py:2358-2390; the examples end with it).
        Quads are drawn as the two triangles (0,1,2) and (0,2,3)."""
        import matplotlib.tri as tri

        conn, X = self.model.conn, self.model.X
        if self.model.nnodes_per_elem == 4:
            triangles = np.concatenate([conn[:, [0, 1, 2]], conn[:, [0, 2, 3]]], axis=0)
        elif self.model.nnodes_per_elem == 3:
            triangles = conn
        else:
            raise ValueError("unsupported element type")
        ax.set_aspect("equal")
        ax.tricontourf(tri.Triangulation(X[:, 0], X[:, 1], triangles), np.asarray(u), **kwargs)


# ------------------------------------------------------------------------------------------
# corner offsets (x, y, z), in cells, of the six tetrahedra that fill a 2x2x2-cell patch (table of pyfem.py:2547-2556)
_TET_CORNERS = np.array(
    [[(0, 0, 0), (2, 0, 0), (2, 2, 0), (2, 2, 2)], [(0, 0, 0), (2, 0, 0), (2, 2, 2), (0, 0, 2)],
     [(2, 0, 0), (2, 2, 2), (0, 0, 2), (2, 0, 2)], [(0, 0, 0), (0, 2, 0), (0, 2, 2), (2, 2, 0)],
     [(0, 0, 0), (0, 2, 2), (0, 0, 2), (2, 2, 2)], [(0, 0, 0), (2, 2, 2), (2, 2, 0), (0, 2, 2)]])


class ProblemCreator:
    """Vectorised equivalent of the reference's loop-based generator (pyfem.py:2426-2773) for the element
    reference's element types (quad, tri, block, tet, brick20); produces identical conn / X / dof_fixed / nodal_force."""

    def __init__(self, nnodes_x, nnodes_y, nnodes_z=None, Lx=None, Ly=None, Lz=None, element_type="quad"):
        if nnodes_z is None:
            self.ndims = 2
            nnodes_z = 1
            assert element_type == "quad" or element_type == "tri"
        else:
            self.ndims = 3
            assert element_type == "block" or element_type == "tet" or element_type == "brick20"
        Lx = (nnodes_x - 1) / (nnodes_y - 1) if Lx is None else Lx
        Ly = 1.0 if Ly is None else Ly
        Lz = (nnodes_z - 1) / (nnodes_y - 1) if Lz is None else Lz
        x, y, z = np.linspace(0, Lx, nnodes_x), np.linspace(0, Ly, nnodes_y), np.linspace(0, Lz, nnodes_z)
        X = np.stack([np.tile(x, nnodes_y * nnodes_z), np.tile(np.repeat(y, nnodes_x), nnodes_z),
                      np.repeat(z, nnodes_x * nnodes_y)], axis=1)
        self.X = np.ascontiguousarray(X[:, : self.ndims])
        self.nodes3d = np.arange(nnodes_x * nnodes_y * nnodes_z).reshape((nnodes_z, nnodes_y, nnodes_x))
        if self.ndims == 2:
            n00 = self.nodes3d[0, :-1, :-1].ravel()
        else:
            n00 = self.nodes3d[:-1, :-1, :-1].ravel()
        n10, n11, n01 = n00 + 1, n00 + 1 + nnodes_x, n00 + nnodes_x
        if element_type == "quad":
            conn = np.stack([n00, n10, n11, n01], axis=1)
        elif element_type == "tri":
            conn = np.empty((2 * n00.size, 3), dtype=int)
            conn[0::2] = np.stack([n00, n10, n11], axis=1)
            conn[1::2] = np.stack([n11, n01, n00], axis=1)
        elif element_type == "block":
            lo = np.stack([n00, n10, n11, n01], axis=1)
            conn = np.concatenate([lo, lo + nnodes_x * nnodes_y], axis=1)
        else:  # one patch of 2x2x2 cells (27 grid nodes) per brick / per six tetrahedra; node counts must be odd
            px, py, pz = (nnodes_x - 1) // 2, (nnodes_y - 1) // 2, (nnodes_z - 1) // 2
            k, j, i = np.meshgrid(np.arange(pz), np.arange(py), np.arange(px), indexing="ij")
            origin = 2 * np.stack([i.ravel(), j.ravel(), k.ravel()], axis=1)

            def nid(c):
                return c[..., 0] + c[..., 1] * nnodes_x + c[..., 2] * nnodes_x * nnodes_y

            if element_type == "brick20":  # ref: pyfem.py:2657-2700 -- the basis' own node order, shifted to 0..2
                conn = nid(origin[:, None, :] + (BasisBrick20Nodes._NODES.astype(int) + 1)[None])
            else:  # "tet", ref: pyfem.py:2538-2655: six tetrahedra per patch, corners, then the edge midpoints
                corners = origin[:, None, None, :] + _TET_CORNERS[None]
                pairs = ((0, 1), (1, 2), (0, 2), (0, 3), (1, 3), (2, 3))
                mids = np.stack([(corners[:, :, a] + corners[:, :, b]) // 2 for a, b in pairs], axis=2)
                conn = nid(np.concatenate([corners, mids], axis=2)).reshape(-1, 10)
        self.conn = conn.astype(int)
        self.nnodes_x, self.nnodes_y, self.nnodes_z = nnodes_x, nnodes_y, nnodes_z
        self.nnodes = nnodes_x * nnodes_y * nnodes_z

    def create_poisson_problem(self):
        return self.conn, self.X, list(self.nodes3d[:, :, 0].ravel())

    def create_linear_elasticity_problem(self):
        face = self.nodes3d[:, :, 0].ravel()
        nd = self.ndims
        dof_fixed = list((nd * face[:, None] + np.arange(nd)[None, :]).ravel())
        nodal_force = {int(self.nodes3d[k, 0, -1]): [0.0, -1.0, 0.0][0:nd] for k in range(self.nnodes_z)}
        return self.conn, self.X, dof_fixed, nodal_force

    def create_helmhotz_problem(self):
        i = np.tile(np.arange(self.nnodes_x), self.nnodes_y * self.nnodes_z)
        j = np.tile(np.repeat(np.arange(self.nnodes_y), self.nnodes_x), self.nnodes_z)
        k = np.repeat(np.arange(self.nnodes_z), self.nnodes_x * self.nnodes_y)
        inside = (i < self.nnodes_x / 2) & (j < self.nnodes_y / 2) & (k < self.nnodes_z / 2)
        return self.conn, self.X, np.where(inside, 0.95, 1e-3)
