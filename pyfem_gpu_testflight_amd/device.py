"""Thin object layer over the C ABI: one HipContext per model, torch tensors as device
arrays (the library only ever sees their data_ptr()), and DeviceCSR -- the object
`compute_jacobian()` returns in place of the reference's scipy CSR (pyfem.py:921-931):
values stay in HBM; a scipy matrix is materialised only if host code touches it.
"""
import ctypes as C

import numpy as np

from . import _lib


def _torch():
    import torch

    if not torch.cuda.is_available():
        raise RuntimeError(
            "pyfem_gpu_testflight_amd needs a ROCm GPU (MI355X): torch.cuda.is_available() is False "
            "and there is no CPU fallback for the accelerated path")
    return torch


def _ptr(t):
    return C.c_void_p(0) if t is None else C.c_void_p(t.data_ptr())


class HipContext:
    """Owns a pf_ctx: device mesh copy + symbolic pattern.  ref: ModelBase.__init__ (pyfem.py:640-757)."""

    def __init__(self, X, conn, ndof_per_node, device=None, tables=None, tabular=False):
        """tables = (N (Q,P), dN (Q,P,ndims), wq (Q)) of the basis: always used by the table-driven entry points
        (complex-step imaginary parts, cross-checks); tabular=True makes them THE assembly path (elements other than
        the hard-wired 3-node triangle, 4-node quad, 8-node block: 10-node tetrahedra, 20-node bricks, ...)."""
        torch = _torch()
        self.torch = torch
        self.lib = _lib.load()
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        self._ctx = C.c_void_p()
        _lib.check(None, self.lib.pf_ctx_create(self.device.index, C.byref(self._ctx)), "pf_ctx_create")
        X = np.ascontiguousarray(X, dtype=np.float64)
        conn = np.asarray(conn)
        if X.ndim != 2 or X.shape[1] not in (2, 3):
            raise ValueError("X must have shape (nnodes, 2) or (nnodes, 3)")
        self.ndims = X.shape[1]
        self.tabular = bool(tabular)
        if self.tabular:
            ok = tables is not None and conn.ndim == 2 and self.ndims == 3 and conn.shape[1] != 8 and 4 <= conn.shape[1] <= 32
        else:
            ok = conn.ndim == 2 and conn.shape[1] in ((3, 4) if self.ndims == 2 else (8,))
        if not ok:
            raise ValueError(f"unknown element_type: {conn.shape[1] if conn.ndim == 2 else '?'}-node elements in "
                             f"{self.ndims}-D are not on the accelerated path (2-D: 3-node triangles, 4-node quads; "
                             "3-D: 8-node blocks, or 4..32 nodes with the basis tables)")
        conn32 = np.ascontiguousarray(conn, dtype=np.int32)
        self.nnodes, self.nelems, self.P, self.d = X.shape[0], conn.shape[0], conn.shape[1], int(ndof_per_node)
        self._sync_stream()
        self._call("pf_mesh_set" if self.ndims == 2 else "pf_mesh_set3d", _lib._i64(self.nnodes),
                   X.ctypes.data_as(C.c_void_p), _lib._i64(self.nelems),
                   self.P, conn32.ctypes.data_as(C.c_void_p), 0)
        self._call("pf_pattern_build", self.d)
        nrows, nnz, nblocks, maxrow = _lib._i64(), _lib._i64(), _lib._i64(), _lib._i32()
        self._call("pf_pattern_dims", C.byref(nrows), C.byref(nnz), C.byref(nblocks), C.byref(maxrow))
        self.nrows, self.nnz, self.nblocks, self.max_row_blocks = nrows.value, nnz.value, nblocks.value, maxrow.value
        self._csr_structure = None
        self._tab = None
        if tables is not None:
            N, dN, wq = (np.ascontiguousarray(t, dtype=np.float64) for t in tables)
            Q = wq.shape[0]
            if N.shape != (Q, self.P) or dN.shape != (Q, self.P, self.ndims) or Q > 64:
                raise ValueError("basis tables must be N (Q,P), dN (Q,P,ndims), weights (Q), Q <= 64")
            self.Q = Q
            self._tab = tuple(self.to_device(t, torch.float64) for t in (N, dN, wq))
        if self.tabular and self.d != 3:
            raise ValueError("table-driven elements are built for 3 dofs per node (elasticity)")

    def _tab_args(self):
        if self._tab is None:
            raise ValueError("this context was built without the basis tables")
        return (self.Q,) + tuple(_ptr(t) for t in self._tab)

    # -- table-driven kernels (pf_tabular.hip): any element given by its tables; drho selects the coefficient
    #    RAMP'(rho_q) (N drho)_q, i.e. the imaginary part of K(rho + i drho) per unit step --------------------------
    def assemble_tab(self, kind, E, nu, p, rho=None, drho=None, out=None):
        vals = self.empty(self.nnz) if out is None else out
        self._sync_stream()
        self._call("pf_assemble_tab", *self._tab_args(), kind, E, nu, p, _ptr(rho), _ptr(drho), _ptr(vals))
        return vals

    def element_matrices_tab(self, kind, E, nu, p, rho=None, drho=None):
        D = self.P * self.d
        out = self.empty(self.nelems * D * D)
        self._sync_stream()
        self._call("pf_element_matrices_tab", *self._tab_args(), kind, E, nu, p, _ptr(rho), _ptr(drho), _ptr(out))
        return out.view(self.nelems, D, D)

    def sens_tab(self, kind, E, nu, p, rho, phi, psi):
        out = self.empty(self.nnodes)
        self._sync_stream()
        self._call("pf_sens_tab", *self._tab_args(), kind, E, nu, p, _ptr(rho), _ptr(phi), _ptr(psi), _ptr(out))
        return out

    # -- plumbing -------------------------------------------------------------------------
    def _sync_stream(self):
        s = self.torch.cuda.current_stream(self.device).cuda_stream
        self.lib.pf_ctx_set_stream(self._ctx, C.c_void_p(s))

    def _call(self, name, *args):
        rc = getattr(self.lib, name)(self._ctx, *args)
        _lib.check(self._ctx, rc, name)

    def close(self):
        if getattr(self, "_ctx", None) and self._ctx.value:
            self.lib.pf_ctx_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def empty(self, n, dtype=None):
        return self.torch.empty(int(n), dtype=dtype or self.torch.float64, device=self.device)

    def zeros(self, n, dtype=None):
        return self.torch.zeros(int(n), dtype=dtype or self.torch.float64, device=self.device)

    def to_device(self, a, dtype=None):
        torch = self.torch
        if isinstance(a, torch.Tensor):
            return a.to(device=self.device, dtype=dtype or a.dtype).contiguous()
        return torch.from_numpy(np.ascontiguousarray(a)).to(device=self.device, dtype=dtype)

    # -- symbolic -------------------------------------------------------------------------
    def csr_structure(self):
        """(indptr, indices) int32 numpy arrays in scipy's layout (sorted columns)."""
        if self._csr_structure is None:
            torch = self.torch
            indptr = self.empty(self.nrows + 1, torch.int32)
            indices = self.empty(self.nnz, torch.int32)
            self._sync_stream()
            self._call("pf_pattern_get_csr", _ptr(indptr), _ptr(indices))
            self._csr_structure = (indptr.cpu().numpy(), indices.cpu().numpy())
        return self._csr_structure

    def graph(self):
        torch = self.torch
        nptr = self.empty(self.nnodes + 1, torch.int32)
        nidx = self.empty(self.nblocks, torch.int32)
        self._sync_stream()
        self._call("pf_pattern_get_graph", _ptr(nptr), _ptr(nidx))
        return nptr, nidx

    def assembly_maps(self):
        """Host copies of the fused assembly kernel's gather maps (pf_pattern_get_tiles): dict with npt, max_tile_blocks (pf_pattern_tile_info), tile_desc
        (ntiles,4), vis_rec (nvisits,4 | 12), rec8 (nvisits,2) / rec12 (nvisits,3) uint32 or None (the compact records of pf_pattern_get_compact), n2e_ptr
        (nnodes+1), nptr (nnodes+1).  For tests that replay the kernel's index arithmetic."""
        torch = self.torch
        nt, npt, nv = _lib._i64(), _lib._i32(), _lib._i64()
        self._sync_stream()
        self._call("pf_pattern_get_tiles", C.byref(nt), C.byref(npt), C.byref(nv), None, None, None)
        w = 12 if self.P == 8 else 4
        td = self.empty(4 * nt.value, torch.int32)
        vr = self.empty(w * nv.value, torch.int32)
        n2e = self.empty(self.nnodes + 1, torch.int32)
        self._call("pf_pattern_get_tiles", None, None, None, _ptr(td), _ptr(vr), _ptr(n2e))
        nptr, _ = self.graph()
        n8, rb = _lib._i64(), _lib._i32()
        self._call("pf_pattern_get_compact", C.byref(rb), C.byref(n8), None)
        rec8 = rec12 = None
        if n8.value:
            rc = self.empty(rb.value // 4 * n8.value, torch.int32)
            self._call("pf_pattern_get_compact", None, None, _ptr(rc))
            rc = rc.cpu().numpy().view(np.uint32).reshape(-1, rb.value // 4)
            rec8, rec12 = (rc, None) if rb.value == 8 else (None, rc)
        mtb = _lib._i32()
        self._call("pf_pattern_tile_info", C.byref(mtb))
        return {"npt": npt.value, "max_tile_blocks": mtb.value, "rec8": rec8, "rec12": rec12, "tile_desc": td.cpu().numpy().reshape(-1, 4), "vis_rec": vr.cpu().numpy().reshape(-1, w),
                "n2e_ptr": n2e.cpu().numpy(), "nptr": nptr.cpu().numpy()}

    # -- numeric --------------------------------------------------------------------------
    def assemble_elasticity(self, E, nu, p, rho=None, mode=_lib.PF_SCATTER_GATHER, out=None):
        vals = self.empty(self.nnz) if out is None else out
        self._sync_stream()
        if self.tabular:
            self._call("pf_assemble_tab", *self._tab_args(), _lib.PF_TAB_ELASTICITY, E, nu, p, _ptr(rho), None, _ptr(vals))
        elif self.ndims == 3:
            self._call("pf_assemble_elasticity3d", E, nu, p, _ptr(rho), _ptr(vals))
        else:
            self._call("pf_assemble_elasticity2d", E, nu, p, _ptr(rho), mode, _ptr(vals))
        return vals

    def assemble_helmholtz(self, r0):
        K, R = self.empty(self.nnz), self.empty(self.nnz)
        self._sync_stream()
        self._call("pf_assemble_helmholtz2d" if self.ndims == 2 else "pf_assemble_helmholtz3d", r0, _ptr(K), _ptr(R))
        return K, R

    def sens_elasticity(self, E, nu, p, rho, phi, psi):
        out = self.empty(self.nnodes)
        self._sync_stream()
        if self.tabular:
            self._call("pf_sens_tab", *self._tab_args(), _lib.PF_TAB_ELASTICITY, E, nu, p, _ptr(rho), _ptr(phi), _ptr(psi),
                       _ptr(out))
        elif self.ndims == 3:
            self._call("pf_sens3d", 0, E, nu, p, _ptr(rho), _ptr(phi), _ptr(psi), _ptr(out))
        else:
            self._call("pf_sens_elasticity2d", E, nu, p, _ptr(rho), _ptr(phi), _ptr(psi), _ptr(out))
        return out

    def assemble_poisson(self, p, rho=None, out=None):
        vals = self.empty(self.nnz) if out is None else out
        self._sync_stream()
        self._call("pf_assemble_poisson2d" if self.ndims == 2 else "pf_assemble_poisson3d", p, _ptr(rho), _ptr(vals))
        return vals

    def element_matrices_poisson(self, p, rho=None):
        out = self.empty(self.nelems * self.P * self.P)
        self._sync_stream()
        if self.ndims == 3:
            self._call("pf_element_matrices3d", 1, 0.0, 0.0, p, 0.0, _ptr(rho), _ptr(out), None)
        else:
            self._call("pf_element_matrices_poisson2d", p, _ptr(rho), _ptr(out))
        return out.view(self.nelems, self.P, self.P)

    def quad_points(self):
        Q = {3: 1, 4: 4, 8: 8}[self.P]
        out = self.empty(self.nelems * Q * self.ndims)
        self._sync_stream()
        self._call("pf_quad_points" if self.ndims == 2 else "pf_quad_points3d", _ptr(out))
        return out.view(self.nelems, Q, self.ndims)

    def poisson_rhs(self, gq):
        out = self.empty(self.nnodes)
        self._sync_stream()
        self._call("pf_poisson_rhs" if self.ndims == 2 else "pf_poisson_rhs3d", _ptr(gq), _ptr(out))
        return out

    def sens_poisson(self, p, rho, phi, psi):
        out = self.empty(self.nnodes)
        self._sync_stream()
        if self.ndims == 3:
            self._call("pf_sens3d", 1, 0.0, 0.0, p, _ptr(rho), _ptr(phi), _ptr(psi), _ptr(out))
        else:
            self._call("pf_sens_poisson2d", p, _ptr(rho), _ptr(phi), _ptr(psi), _ptr(out))
        return out

    def element_matrices_elasticity(self, E, nu, p, rho=None):
        D = self.P * self.ndims
        out = self.empty(self.nelems * D * D)
        self._sync_stream()
        if self.tabular:
            self._call("pf_element_matrices_tab", *self._tab_args(), _lib.PF_TAB_ELASTICITY, E, nu, p, _ptr(rho), None,
                       _ptr(out))
        elif self.ndims == 3:
            self._call("pf_element_matrices3d", 0, E, nu, p, 0.0, _ptr(rho), _ptr(out), None)
        else:
            self._call("pf_element_matrices_elasticity2d", E, nu, p, _ptr(rho), _ptr(out))
        return out.view(self.nelems, D, D)

    def element_matrices_helmholtz(self, r0):
        Ke, Re = self.empty(self.nelems * self.P * self.P), self.empty(self.nelems * self.P * self.P)
        self._sync_stream()
        if self.ndims == 3:
            self._call("pf_element_matrices3d", 2, 0.0, 0.0, 0.0, r0, None, _ptr(Ke), _ptr(Re))
        else:
            self._call("pf_element_matrices_helmholtz2d", r0, _ptr(Ke), _ptr(Re))
        return Ke.view(self.nelems, self.P, self.P), Re.view(self.nelems, self.P, self.P)

    def vals_to_csr(self, vals):
        out = self.empty(self.nnz)
        self._sync_stream()
        self._call("pf_vals_to_csr", _ptr(vals), _ptr(out))
        return out

    def upload_csr(self, K):
        """Device values (blocked layout) of a HOST scipy matrix whose entries lie inside this context's pattern --
        e.g. `K + K2` or `2.0 * K` computed with scipy on matrices that compute_jacobian() returned.  Entries the
        matrix does not store are zero; an entry outside the pattern is a ValueError.  Boundary path (one upload of
        nnz values), not a hot one."""
        from scipy import sparse

        torch = self.torch
        K = sparse.csr_matrix(K)
        if K.shape != (self.nrows, self.nrows):
            raise ValueError(f"matrix of shape {K.shape} does not fit the model's {self.nrows} dofs")
        if np.iscomplexobj(K.data):
            raise TypeError("complex matrices take the complex-step path (compute_jacobian with a complex density)")
        K.sum_duplicates()
        indptr, indices = self.csr_structure()
        rows = np.repeat(np.arange(self.nrows, dtype=np.int64), np.diff(K.indptr))
        key = rows * self.nrows + K.indices
        pat = np.repeat(np.arange(self.nrows, dtype=np.int64), np.diff(indptr)) * self.nrows + indices
        pos = np.searchsorted(pat, key)
        if key.size and (pos.max(initial=0) >= pat.size or not np.array_equal(pat[np.minimum(pos, pat.size - 1)], key)):
            raise ValueError("the matrix has entries outside the sparsity pattern of this model")
        data = np.zeros(self.nnz)
        data[pos] = K.data
        csr_d = self.to_device(data, torch.float64)
        if self.d == 1:
            return csr_d
        if getattr(self, "_csr_perm", None) is None:  # csr position -> blocked position, from the export kernel itself
            ar = torch.arange(self.nnz, dtype=torch.float64, device=self.device)
            self._csr_perm = self.vals_to_csr(ar).to(torch.int64)
        vals = self.empty(self.nnz)
        vals[self._csr_perm] = csr_d
        return vals

    def apply_dirichlet(self, vals, rhs, mask, fixed_vals=None, symmetric=True, diag_scale=None):
        self._sync_stream()
        self._call("pf_apply_dirichlet", _ptr(vals), _ptr(rhs), _ptr(mask), _ptr(fixed_vals), int(bool(symmetric)),
                   _ptr(diag_scale))

    def spmv(self, vals, x, out=None):
        y = self.empty(self.nrows) if out is None else out
        self._sync_stream()
        self._call("pf_spmv", _ptr(vals), _ptr(x), _ptr(y))
        return y

    def diag(self, vals):
        out = self.empty(self.nrows)
        self._sync_stream()
        self._call("pf_extract_diag", _ptr(vals), _ptr(out))
        return out

    def cg(self, vals, b, x0=None, rtol=1e-8, atol=0.0, maxiter=0, precond=_lib.PF_PRECOND_JACOBI, check_every=0,
           allow_maxiter=False):
        """Jacobi-PCG on device.  Returns (x, iters, relres); raises RuntimeError on non-convergence
        like the reference (pyfem.py:2421-2422) unless allow_maxiter (fixed-iteration benchmarking)."""
        x = self.zeros(self.nrows) if x0 is None else x0.clone()
        iters, relres = _lib._i32(), _lib._f64()
        self._sync_stream()
        rc = self.lib.pf_cg(self._ctx, _ptr(vals), _ptr(b), _ptr(x), rtol, atol, int(maxiter), precond,
                            int(check_every), C.byref(iters), C.byref(relres))
        if not (allow_maxiter and rc == _lib.PF_E_NOCONV):
            _lib.check(self._ctx, rc, "pf_cg")
        self._warn()
        return x, iters.value, relres.value

    def amg_set_power_steps(self, steps):
        """Power-iteration steps of the AMG hierarchy's lambda_max estimates from the next numeric setup on (default 10)."""
        self._sync_stream()
        self._call("pf_amg_set_power_steps", int(steps))

    def cg2(self, vals, b2, x0=None, rtol=1e-8, atol=0.0, maxiter=0, reuse=True):
        """AMG-PCG for two right-hand sides of a scalar matrix at once (pf_cg2): b2 is an (nnodes, 2) device array.
        Returns (x2, (iters0, iters1), (relres0, relres1)); raises like cg() when either fails to converge and
        ValueError (PF_E_UNSUPPORTED) when the pair has to be solved one by one."""
        if b2.dim() != 2 or b2.shape[1] != 2 or b2.shape[0] != self.nnodes or not b2.is_contiguous():
            raise ValueError("cg2: b2 must be a contiguous (nnodes, 2) array")
        x = self.torch.zeros_like(b2) if x0 is None else x0.clone()
        iters, relres = (_lib._i32 * 2)(), (_lib._f64 * 2)()
        self._sync_stream()
        rc = self.lib.pf_cg2(self._ctx, _ptr(vals), _ptr(b2), _ptr(x), rtol, atol, int(maxiter), 1 if reuse else 0, iters, relres)
        _lib.check(self._ctx, rc, "pf_cg2")
        return x, (iters[0], iters[1]), (relres[0], relres[1])

    def _warn(self):
        """Surface the library's non-fatal conditions (a refused preconditioner means a 100x slower solve)."""
        msg = self.lib.pf_last_warning(self._ctx)
        if msg:
            import warnings

            warnings.warn(msg.decode(), RuntimeWarning, stacklevel=3)

    def bicgstab(self, vals, b, pvals=None, x0=None, rtol=1e-8, atol=0.0, maxiter=0, precond=_lib.PF_PRECOND_JACOBI):
        """Nonsymmetric solve on the device (pf_bicgstab); pvals: SPD companion for the AMG preconditioner."""
        x = self.zeros(self.nrows) if x0 is None else x0.clone()
        iters, relres = _lib._i32(), _lib._f64()
        self._sync_stream()
        rc = self.lib.pf_bicgstab(self._ctx, _ptr(vals), _ptr(pvals), _ptr(b), _ptr(x), rtol, atol, int(maxiter), precond,
                                  C.byref(iters), C.byref(relres))
        _lib.check(self._ctx, rc, "pf_bicgstab")
        self._warn()
        return x, iters.value, relres.value

    def nlpoisson(self, what, u, ck):
        """NonlinearPoisson2D pieces at the state u (pf_nlpoisson_tab): tangent values, its SPD part, element tangents
        (nelems,P,P) or the assembled residual (nnodes)."""
        n = {_lib.PF_NL_TANGENT: self.nnz, _lib.PF_NL_TANGENT_SPD: self.nnz,
             _lib.PF_NL_ELEMENT_TANGENT: self.nelems * self.P * self.P, _lib.PF_NL_RESIDUAL: self.nnodes}[what]
        out = self.empty(n)
        self._sync_stream()
        self._call("pf_nlpoisson_tab", *self._tab_args(), what, _ptr(u), _ptr(ck), int(ck.numel()), _ptr(out))
        return out

    def amg_info(self):
        """Hierarchy of the last AMG-preconditioned solve: list of (nodes, dofs per node, omega)."""
        cap = 16
        nl = _lib._i32()
        nodes = (_lib._i64 * cap)()
        dpn = (_lib._i32 * cap)()
        om = (_lib._f64 * cap)()
        self._call("pf_amg_info", C.byref(nl), nodes, dpn, om, cap)
        return [(nodes[i], dpn[i], om[i]) for i in range(nl.value)]

    def hbm_probe(self, mode, nbytes, reps=20, nt=False):
        """GB/s this device sustains now for a pure write / pure read / 1:3 read:write mix of `nbytes` (pf_hbm_probe);
        nt: 0 plain accesses, 1 non-temporal loads and stores, 2 non-temporal stores only."""
        ms = _lib._f64()
        self._sync_stream()
        self._call("pf_hbm_probe", int(mode), int(nt), _lib._i64(int(nbytes)), int(reps), C.byref(ms))
        return nbytes / (ms.value * 1e-3) / 1e9

    def time_kernel(self, which, reps, in0=None, in1=None, out0=None, E=0.0, nu=0.0, p=0.0):
        ms = _lib._f64()
        self._sync_stream()
        self._call("pf_time_kernel", which, reps, _ptr(in0), _ptr(in1), _ptr(out0), E, nu, p, C.byref(ms))
        return ms.value


def _csr_base():
    from scipy import sparse

    return sparse.csr_matrix


class DeviceCSR(_csr_base()):
    """Device-resident sparse matrix on a HipContext pattern that IS a scipy CSR matrix for host code.

    `compute_jacobian()` / `apply_dirichlet_bcs()` of the reference return scipy CSR matrices (pyfem.py:921-931, 835) and
    its tests and examples hand them to scipy (`spsolve(K, rhs)`, `K - K2`, `abs(K)`, `K[i, j]`, `K.diagonal()`,
    `K.nnz`, printing).  This subclass keeps the values in HBM and materialises `data` / `indices` / `indptr` on the
    host only when something reads them (explicit zeros are dropped at that point if `eliminate_zeros` was requested,
    which is what apply_dirichlet_bcs does in the reference, pyfem.py:826).  `dot` / `@` with a real vector run the
    HIP SpMV without materialising anything.  The device values are authoritative for the GPU path: host arrays are a
    read-only view of them (`invalidate()` drops the view after the device values changed).

    Results of scipy arithmetic on it (`K - K2`, `abs(K)`, `K.copy()` of a host-only instance, ...) are ordinary host
    matrices of this class with no device side (`hctx is None`)."""

    def __init__(self, arg1, vals=None, eliminate_zeros=False, shape=None, dtype=None, copy=False):
        if isinstance(arg1, HipContext):
            self.hctx, self.vals, self._eliminate = arg1, vals, eliminate_zeros
            self._shape = (arg1.nrows, arg1.nrows)
            self.maxprint = 50
        else:  # scipy's own constructor signature (used by inherited methods that build their result)
            self.hctx, self.vals, self._eliminate = None, None, False
            _csr_base().__init__(self, arg1, shape=shape, dtype=dtype, copy=copy)

    # -- lazy host view ------------------------------------------------------------------------
    def _materialise(self):
        from scipy import sparse

        indptr, indices = self.hctx.csr_structure()
        data = self.hctx.vals_to_csr(self.vals).cpu().numpy()
        K = sparse.csr_matrix((data, indices.copy(), indptr.copy()), shape=self._shape)
        if self._eliminate:
            K.eliminate_zeros()
        d = self.__dict__
        d["_h_data"], d["_h_indices"], d["_h_indptr"] = K.data, K.indices, K.indptr

    def _host_array(self, name):
        d = self.__dict__
        if name not in d:
            if d.get("hctx") is None:
                raise AttributeError(name)
            self._materialise()
        return d[name]

    data = property(lambda self: self._host_array("_h_data"), lambda self, v: self.__dict__.__setitem__("_h_data", v))
    indices = property(lambda self: self._host_array("_h_indices"),
                       lambda self, v: self.__dict__.__setitem__("_h_indices", v))
    indptr = property(lambda self: self._host_array("_h_indptr"),
                      lambda self, v: self.__dict__.__setitem__("_h_indptr", v))

    @property
    def dtype(self):
        h = self.__dict__.get("_h_data")
        return np.dtype(np.float64) if h is None else h.dtype

    def invalidate(self):
        """Drop the host view (the device values were modified)."""
        if self.__dict__.get("hctx") is not None:
            for k in ("_h_data", "_h_indices", "_h_indptr"):
                self.__dict__.pop(k, None)

    def to_scipy(self):
        """A plain scipy CSR matrix over the host view (no copy)."""
        return _csr_base()((self.data, self.indices, self.indptr), shape=self.shape)

    tocsr = to_scipy

    # -- device paths ----------------------------------------------------------------------------
    def dot(self, v):
        hctx = self.__dict__.get("hctx")
        if hctx is not None:
            torch = hctx.torch
            if isinstance(v, torch.Tensor):
                return hctx.spmv(self.vals, v)
            if isinstance(v, np.ndarray) and v.ndim == 1 and not np.iscomplexobj(v) and v.shape[0] == self.shape[1]:
                return hctx.spmv(self.vals, hctx.to_device(v, torch.float64)).cpu().numpy()
        return _csr_base().dot(self, v)

    def __matmul__(self, v):
        return self.dot(v)

    def copy(self):
        if self.__dict__.get("hctx") is not None:
            return DeviceCSR(self.hctx, self.vals.clone(), self._eliminate)
        return _csr_base().copy(self)
