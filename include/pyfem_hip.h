/*
 * pyfem_hip.h -- C ABI of libpyfem_hip.so, the MI355X (gfx950) implementation of the
 * pyfem hot path: per-element Ke quadrature -> global sparse scatter -> Dirichlet BCs ->
 * CSR SpMV / PCG (Jacobi or smoothed-aggregation AMG V-cycle).
 *
 * The reference (aaronyicongfu/pyfem_gpu_testflight) is pure Python and has no FFI; each
 * entry point below names the reference method(s) it replaces (file:line relative to the
 * reference checkout).  The Python shim that binds these with ctypes is
 * pyfem_gpu_testflight_amd/_lib.py; INTEGRATION.md shows the stub a maintainer of the
 * reference would add.
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error (PF_E_*); the message is available
 *     from pf_last_error(ctx).  Nothing throws across the ABI.
 *   - all array arguments are DEVICE pointers unless the name ends in _host or the doc says
 *     otherwise; the caller owns them (the Python shim passes torch tensor data_ptr()s).
 *   - the library owns the mesh copy and the symbolic pattern it builds (hipMalloc'd,
 *     released by pf_ctx_destroy).
 *   - work is enqueued on the stream given to pf_ctx_set_stream (default: the null stream);
 *     calls that return host scalars synchronise that stream, the others are asynchronous.
 *   - one ctx per host thread; a ctx is bound to one device.
 *
 * Matrix value layout ("vals"): the sparsity pattern is the node graph (nptr/nidx: node n
 * couples to nodes nidx[nptr[n] .. nptr[n+1]), sorted ascending); each coupling carries a
 * dense d x d block (d = dofs per node), stored block-contiguous, row-major inside the
 * block:   vals[(nptr[n]+c)*d*d + k*d + l]  =  K[d*n+k, d*nidx[nptr[n]+c]+l].
 * For d == 1 this IS scipy's CSR data order; for d == 2 pf_vals_to_csr converts to the
 * CSR data order `coo_matrix(...).tocsr()` produces (reference pyfem.py:921-931).
 */
#ifndef PYFEM_HIP_H
#define PYFEM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pf_ctx pf_ctx;

enum {
  PF_OK = 0,
  PF_E_INVALID = -1,   /* bad argument / call order            */
  PF_E_HIP = -2,       /* a HIP runtime call failed             */
  PF_E_UNSUPPORTED = -3, /* element type / size outside the path */
  PF_E_NOCONV = -4     /* CG hit maxiter (reference: RuntimeError, pyfem.py:2421-2422) */
};

enum { PF_QUAD4 = 4, PF_TRI3 = 3, PF_BLOCK8 = 8 };  /* nodes per element selects the element */
enum { PF_TAB_MAX_NODES = 32, PF_TAB_MAX_QUADS = 64 };  /* limits of the table-driven element */
enum { PF_SCATTER_GATHER = 0, PF_SCATTER_ATOMIC = 1 };
enum { PF_PRECOND_NONE = 0, PF_PRECOND_JACOBI = 1, PF_PRECOND_AMG = 2, PF_PRECOND_AMG_REUSE = 3, PF_PRECOND_AMG_STALE = 4 };

/* ---- context -------------------------------------------------------------------------- */
int pf_ctx_create(int device, pf_ctx** out);
int pf_ctx_destroy(pf_ctx* ctx);
const char* pf_last_error(pf_ctx* ctx);       /* ctx may be NULL: last global error      */
/* Non-fatal conditions of the last calls (the call itself returned PF_OK), e.g. "the matrix is not definite, the AMG
 * preconditioner was refused and Jacobi is used" from pf_cg.  Returns "" when there is none; reading clears it. */
const char* pf_last_warning(pf_ctx* ctx);
int pf_ctx_set_stream(pf_ctx* ctx, void* hip_stream);
int pf_version(void);

/* ---- mesh: replaces ModelBase.__init__ storage (pyfem.py:659-661, 705-706) -------------- */
/* X: (nnodes, 2) float64 row-major; conn: (nelems, P) int32 row-major, P in {3,4}.
 * on_device != 0: pointers are device pointers; else host pointers.  Data is copied.
 * Checks conn.min()==0 and conn.max()==nnodes-1 like pyfem.py:680-681 (PF_E_INVALID). */
int pf_mesh_set(pf_ctx* ctx, int64_t nnodes, const double* X, int64_t nelems, int P,
                const int32_t* conn, int on_device);
/* 3-D: X (nnodes, 3), conn (nelems, 8) in BasisBlock3D node order (pyfem.py:287-338; ProblemCreator
 * element_type="block", pyfem.py:2531-2548).  pf_pattern_build then takes ndof_per_node 1 or 3.
 * Any other P from 4 to PF_TAB_MAX_NODES selects the table-driven element (pf_*_tab below): the 10-node
 * tetrahedron and the 20-node brick of pyfem.py:380-632 / ProblemCreator "tet", "brick20" (pyfem.py:2538-2700);
 * the pattern then holds the node rows only (up to 160 neighbours per node). */
int pf_mesh_set3d(pf_ctx* ctx, int64_t nnodes, const double* X, int64_t nelems, int P /* PF_BLOCK8 or 4..32 */,
                  const int32_t* conn, int on_device);

/* ---- symbolic pattern: replaces utils.create_dof (utils.py:268-298) +
 *      ModelBase._compute_nz_pattern (pyfem.py:838-858) + the structure half of
 *      coo_matrix(...).tocsr() (pyfem.py:930-931) ------------------------------------------- */
int pf_pattern_build(pf_ctx* ctx, int ndof_per_node /* 1 or the space dimension */);
/* nrows = d*nnodes, nnz = d*d*nblocks */
int pf_pattern_dims(pf_ctx* ctx, int64_t* nrows, int64_t* nnz, int64_t* nblocks, int32_t* max_row_blocks);
/* node graph (device out): nptr int32[nnodes+1], nidx int32[nblocks]; either may be NULL */
int pf_pattern_get_graph(pf_ctx* ctx, int32_t* nptr, int32_t* nidx);
/* The compact visit records the assembly kernel reads instead of the 16-byte ones, if any (HOST out rec_bytes: 8, 12 or
 * 0 = none; nvisits; DEVICE out recs uint32[rec_bytes/4 * nvisits], may be NULL).  With signed node offsets from the
 * pair's own node:
 *   quads, 8 bytes      d0 = off1 | off2<<13 | (off3&63)<<26, d1 = off3>>6 | pos0<<7 | pos1<<12 | pos2<<17 | pos3<<22 | slot<<27
 *   triangles, 8 bytes  d0 = (off1&0xffff) | off2<<16,        d1 = pos0 | pos1<<5 | pos2<<10 | slot<<15 | a<<20
 *   triangles, 12 bytes {o1, o2, w} = the 16-byte record without its unused third node (numberings the offsets do not fit)
 * For tests. */
int pf_pattern_get_compact(pf_ctx* ctx, int32_t* rec_bytes, int64_t* nvisits, uint32_t* recs);
/* scipy-layout CSR structure (device out): indptr int32[nrows+1], indices int32[nnz] */
int pf_pattern_get_csr(pf_ctx* ctx, int32_t* indptr, int32_t* indices);
/* Gather maps of the fused assembly kernel (2-D meshes and 8-node blocks), for tests that replay its index
 * arithmetic on the host.  Sizes come back through the HOST outputs ntiles / npt / nvisits (any may be NULL);
 * the DEVICE outputs may be NULL: tile_desc int32[4*ntiles] = {first node, first visit, first block,
 * nodes | (a node with > 8 incident elements)<<7 | visits<<8 | blocks<<16} per tile, vis_rec int32[4*nvisits] (12*nvisits for blocks) = the visit records,
 * n2e_ptr int32[nnodes+1] = first visit of each node.  No counterpart in the reference (its scatter is
 * coo_matrix(...).tocsr(), pyfem.py:921-931). */
int pf_pattern_get_tiles(pf_ctx* ctx, int64_t* ntiles, int32_t* npt, int64_t* nvisits, int32_t* tile_desc,
                         int32_t* vis_rec, int32_t* n2e_ptr);
/* max_tile_blocks: blocks of the largest tile's segment = the plane size of the assembly kernel's LDS image (HOST out) */
int pf_pattern_tile_info(pf_ctx* ctx, int32_t* max_tile_blocks);

/* ---- numeric assembly --------------------------------------------------------------------
 * pf_assemble_elasticity2d replaces LinearElasticity.compute_jacobian (pyfem.py:1771-1795):
 *   _update_material_property (:1923-1945, RAMP rho/(1+p(1-rho)) at quadrature points),
 *   _compute_element_jacobian (:2030-2068 = utils.compute_jtrans/jdet/basis_grad,
 *   _compute_element_Bmat :1976, _einsum_element_jacobian :2016) and
 *   ModelBase._assemble_jacobian (:921-931), fused: Ke is never written to memory.
 * rho: nodal density float64[nnodes] or NULL (== 1.0 everywhere, pyfem.py:1780-1781).
 * mode: PF_SCATTER_GATHER (owner-computes rows, no atomics, deterministic) or
 *       PF_SCATTER_ATOMIC (one lane per element, fp64 atomics; cross-check only).       */
int pf_assemble_elasticity2d(pf_ctx* ctx, double E, double nu, double p, const double* rho,
                             int mode, double* Kvals);
/* Helmholtz.__init__ / _compute_element_jacobian_and_rhs (pyfem.py:2085-2099, 2139-2177):
 * Kvals = r0^2*stiffness + mass, Rvals = mass (either may be NULL).  d must be 1.        */
int pf_assemble_helmholtz2d(pf_ctx* ctx, double r0, double* Kvals, double* Rvals);

/* Sensitivity of phi^T K(rho) psi w.r.t. the nodal density: LinearElasticity._compute_K_dv_sens
 * (pyfem.py:1873-1920) with _update_material_property_deriv (:1948-1973), fused -- the (E,D,D,P)
 * tensor Ke_deriv is never formed.  rho float64[nnodes] (required), phi/psi float64[2*nnodes],
 * out float64[nnodes].  compliance_grad (pyfem.py:1837-1848) is minus this with phi = psi = u.           */
int pf_sens_elasticity2d(pf_ctx* ctx, double E, double nu, double p, const double* rho,
                         const double* phi, const double* psi, double* out);

/* LinearPoisson, 2-D (pyfem.py:934-1329):
 *   pf_assemble_poisson2d      compute_jacobian (:1006-1030): Ke = sum_q kappa_q detJ w gradN gradN^T with
 *                              kappa_q = RAMP(rho_q) (:1279-1301), fused with the scatter; d must be 1
 *   pf_quad_points             Xq (nelems, Q, 2): the points compute_rhs evaluates the Python gfunc at (:1160)
 *   pf_poisson_rhs             compute_rhs (:997-1004): gq = gfunc(Xq) (nelems, Q) -> rhs (nnodes), including
 *                              the reference's column quirk for triangles (:873)
 *   pf_sens_poisson2d          _compute_K_dv_sens (:1240-1276), fused like the elasticity one           */
int pf_assemble_poisson2d(pf_ctx* ctx, double p, const double* rho, double* Kvals);
int pf_element_matrices_poisson2d(pf_ctx* ctx, double p, const double* rho, double* Ke_out);
int pf_quad_points(pf_ctx* ctx, double* Xq_out);
int pf_poisson_rhs(pf_ctx* ctx, const double* gq, double* rhs_out);
int pf_sens_poisson2d(pf_ctx* ctx, double p, const double* rho, const double* phi, const double* psi,
                      double* out);

/* Element matrices to memory, (nelems, D, D) row-major -- model.Ke_mat / model.Re readers
 * (tests; examples/SciTech2023/performance/performance_test.py:52).                       */
int pf_element_matrices_elasticity2d(pf_ctx* ctx, double E, double nu, double p, const double* rho,
                                     double* Ke_out);
int pf_element_matrices_helmholtz2d(pf_ctx* ctx, double r0, double* Ke_out, double* Re_out);

/* ---- 3-D 8-node block element: QuadratureBlock3D / BasisBlock3D (pyfem.py:97-112,
 *      287-338), 3x3 geometry (utils.py:172-264), 3-D C0 and Bmat (pyfem.py:1752-1757, 1999-2011).
 *      Fused quadrature + scatter like the 2-D kernels; value layout: 3x3 (or 1x1) blocks in node-graph order.
 *      rho: nodal density (RAMP with p) or NULL. ------------------------------------------------------------ */
int pf_assemble_elasticity3d(pf_ctx* ctx, double E, double nu, double p, const double* rho, double* Kvals);
int pf_assemble_poisson3d(pf_ctx* ctx, double p, const double* rho, double* Kvals);        /* pyfem.py:1176-1187 */
int pf_assemble_helmholtz3d(pf_ctx* ctx, double r0, double* Kvals, double* Rvals);         /* pyfem.py:2127-2177 */
/* kind: 0 elasticity (Ke (E,24,24)), 1 Poisson (Ke (E,8,8)), 2 Helmholtz (Ke and Re (E,8,8)); model.Ke_mat users */
int pf_element_matrices3d(pf_ctx* ctx, int kind, double E, double nu, double p, double r0, const double* rho,
                          double* Ke_out, double* Re_out);
/* d(phi^T K(rho) psi)/d rho (nnodes,): scalar == 0 elasticity (phi, psi of length 3 nnodes; pyfem.py:1873-1920),
 * scalar != 0 LinearPoisson (pyfem.py:1240-1276; E, nu ignored).  Fused: Ke_deriv is never formed. */
int pf_sens3d(pf_ctx* ctx, int scalar, double E, double nu, double p, const double* rho, const double* phi,
              const double* psi, double* out);
/* LinearPoisson.compute_rhs pieces on blocks: Xq (nelems, 8, 3); rhs from g at those points (nelems, 8), with the
 * reference's transposed shape-table indexing reproduced (pyfem.py:1131-1136, 861-875) */
int pf_quad_points3d(pf_ctx* ctx, double* Xq_out);
int pf_poisson_rhs3d(pf_ctx* ctx, const double* gq, double* rhs_out);

/* ---- the element given as DATA: N (Q,P) = basis.eval_shape_fun(), dN (Q,P,ndims) = basis.eval_shape_fun_deriv(),
 *      wq (Q) = quadrature.get_weight() (pyfem.py:222-250, 57-65), device arrays; P is that of pf_mesh_set[3d],
 *      Q <= PF_TAB_MAX_QUADS.  kind: PF_TAB_ELASTICITY (pattern with ndims dofs per node; plane stress in 2-D) or
 *      PF_TAB_POISSON (1 dof per node).  Serves
 *        - any (Quadrature, Basis) pair other than the hard-wired ones: QuadratureTetrahedron5Point +
 *          BasisTetrahedron10node, QuadratureBrick333Point + BasisBrick20Nodes (pyfem.py:115-172, 380-632;
 *          tests/test_elements.py, examples/element_plot);
 *        - drho != NULL: the coefficient RAMP'(rho_q) (N drho)_q instead of RAMP(rho_q), i.e. Im K(rho + i drho) per
 *          unit step -- what the reference's complex-step derivative tests read (tests/test_elasticity.py:96-98);
 *        - an independent second implementation of the hard-wired kernels for the parity tests.
 *      Same value layout as the hard-wired assemblies (d x d blocks in node-graph order); the scatter runs colour by
 *      colour over classes of elements that share no node (built on the first call): bitwise reproducible.
 *      compute_jacobian (pyfem.py:1771-1795, 1006-1030) / Ke_mat / _compute_K_dv_sens (pyfem.py:1873-1920, 1240-1276). */
enum { PF_TAB_ELASTICITY = 0, PF_TAB_POISSON = 1 };
int pf_assemble_tab(pf_ctx* ctx, int Q, const double* N, const double* dN, const double* wq, int kind, double E,
                    double nu, double p, const double* rho, const double* drho, double* Kvals);
int pf_element_matrices_tab(pf_ctx* ctx, int Q, const double* N, const double* dN, const double* wq, int kind, double E,
                            double nu, double p, const double* rho, const double* drho, double* Ke_out);
int pf_sens_tab(pf_ctx* ctx, int Q, const double* N, const double* dN, const double* wq, int kind, double E, double nu,
                double p, const double* rho, const double* phi, const double* psi, double* out);
/* NonlinearPoisson2D, -div(h(x)(1 + u^2) grad u) = g with the reference's h and g (pyfem.py:1332-1665), on a 2-D pattern
 * with one dof per node; tables as above.  u: nodal state; ck[k] = xdv[k] * binom(nck - 1, k), k < nck: the design
 * coefficients of h (pyfem.py:1449-1470).  what:
 *   PF_NL_TANGENT          compute_jacobian(xdv, u), nonsymmetric (pyfem.py:1390-1405, 1523-1601)  -> nnz values
 *   PF_NL_TANGENT_SPD      its first, symmetric positive definite term only (what the preconditioner is built from)
 *   PF_NL_ELEMENT_TANGENT  the element matrices (nelems, P, P) (model.Ke_mat)
 *   PF_NL_RESIDUAL         compute_rhs(xdv, u): the assembled residual (pyfem.py:1374-1388, 1472-1521) -> nnodes values */
enum { PF_NL_TANGENT = 0, PF_NL_TANGENT_SPD = 1, PF_NL_ELEMENT_TANGENT = 2, PF_NL_RESIDUAL = 3 };
int pf_nlpoisson_tab(pf_ctx* ctx, int Q, const double* N, const double* dN, const double* wq, int what, const double* u,
                     const double* ck, int nck, double* out);

/* block layout -> scipy CSR data order (identity copy for d == 1) */
int pf_vals_to_csr(pf_ctx* ctx, const double* vals, double* csr_data);

/* ---- boundary conditions: ModelBase.apply_dirichlet_bcs (pyfem.py:781-835) ---------------
 * fixed_mask uint8[nrows] (1 = fixed dof); fixed_vals float64[nrows] (value at fixed dofs,
 * ignored elsewhere) or NULL (all zero, pyfem.py:829-830).  In place on Kvals and rhs:
 * rhs[free] -= K[free,fixed] vals (symmetric only), rows (and columns if symmetric) of
 * fixed dofs zeroed, unit diagonal (diag_scale[dof] instead of 1 when diag_scale != NULL:
 * used by the multi-GPU partition so a shared fixed dof sums to 1 over ranks), rhs[fixed]=vals.
 * Exact zeros stay in the pattern (the reference's eliminate_zeros() happens on export).  */
int pf_apply_dirichlet(pf_ctx* ctx, double* Kvals, double* rhs, const uint8_t* fixed_mask,
                       const double* fixed_vals, int symmetric, const double* diag_scale);

/* ---- sparse linear algebra: scipy CSR matvec / cg call sites (pyfem.py:2119, 2403-2423) -- */
int pf_spmv(pf_ctx* ctx, const double* vals, const double* x, double* y);
int pf_extract_diag(pf_ctx* ctx, const double* vals, double* diag);

/* Preconditioned CG on the ctx pattern:  stop when ||r||_2 <= max(rtol*||b||_2, atol)
 * (the reference's `cg(K, rhs, tol=1e-8, atol=0.0)`, pyfem.py:2418).  x holds the initial
 * guess on entry.  Returns PF_E_NOCONV when maxiter is hit (iters/relres still written).
 * iters, relres: HOST out pointers.  check_every: residual is read back to the host every
 * that many iterations (<=0: default, 50 for Jacobi / 1 for AMG); iteration count is exact to that granularity. */
int pf_cg(pf_ctx* ctx, const double* vals, const double* b, double* x, double rtol, double atol,
          int maxiter, int precond, int check_every, int32_t* iters_host, double* relres_host);
/* AMG-PCG for TWO right-hand sides of one SCALAR matrix at once: b2 and x2 are (nnodes, 2) row-major device arrays (x2 =
 * the initial guesses on entry).  The reference solves the two gradient filters of a topology iteration one after the other
 * with one fixed K (examples/topology_optimization/topo.py:73-74, pyfem.py:2102-2115); here they share every matrix stream
 * and every launch.  The two recurrences are independent: each vector has pf_cg's stopping rule and gets the iterates
 * pf_cg(PF_PRECOND_AMG) would give it alone.  reuse != 0: keep the hierarchy's numeric setup if it was made for `vals`
 * (PF_PRECOND_AMG_REUSE).  iters_host, relres_host: HOST arrays of two.  PF_E_UNSUPPORTED: more than one dof per node, a
 * matrix that is not positive definite, or a hierarchy level the two-vector cycle does not cover (solve one by one). */
int pf_cg2(pf_ctx* ctx, const double* vals, const double* b2, double* x2, double rtol, double atol, int maxiter, int reuse,
           int32_t* iters_host, double* relres_host);
/* BiCGStab for the nonsymmetric systems of the reference: K with Dirichlet conditions on the rows only
 * (enforce_symmetric_K=False) and the Newton tangent of NonlinearPoisson2D, which it solves with scipy gmres or spsolve
 * (pyfem.py:2340-2349, 2414-2420).  Same stopping rule and outputs as pf_cg.  precond: PF_PRECOND_NONE, _JACOBI (diagonal
 * of vals) or _AMG: the V-cycle of pvals, a symmetric positive definite matrix on the same pattern (e.g. the same
 * operator with symmetric boundary conditions; PF_NL_TANGENT_SPD); pvals may be NULL otherwise. */
int pf_bicgstab(pf_ctx* ctx, const double* vals, const double* pvals, const double* b, double* x, double rtol,
                double atol, int maxiter, int precond, int32_t* iters_host, double* relres_host);
/* PF_PRECOND_AMG: smoothed-aggregation AMG V(2,2) cycle -- the device counterpart of the reference's
 * pyamg.smoothed_aggregation_solver(K).aspreconditioner() (pyfem.py:2393-2400) and of
 * pyamg.ruge_stuben_solver (pyfem.py:2098).  Aggregates and coarse graphs are built once per pattern,
 * the coarse operators per call.  The matrix may be positive or negative definite (arm.inp: every detJ < 0); if the
 * coupled dofs carry diagonal entries of both signs pf_cg uses Jacobi instead.  pf_amg_info reports the hierarchy of the last AMG solve (HOST outputs,
 * arrays of length cap).
 * PF_PRECOND_AMG_REUSE keeps the numeric setup when it was made for exactly this `vals` buffer (the Helmholtz filter: one K,
 * many solves).  PF_PRECOND_AMG_STALE keeps WHATEVER numeric setup the hierarchy holds, even one made for other values: the
 * V-cycle of the previous matrix is still a fixed symmetric positive definite operator, so PCG converges to the same
 * solution, in more iterations the further the matrix has moved -- for loops whose matrix changes slowly (late topology
 * optimisation steps), where the setup is a quarter of the solve.  The caller decides when that stops paying
 * (Assembler.amg_lag in the Python mirror); without an earlier setup it is PF_PRECOND_AMG. */
int pf_amg_info(pf_ctx* ctx, int32_t* nlevels, int64_t* nodes, int32_t* dofs_per_node, double* omega, int cap);
/* Power-iteration steps of the hierarchy's lambda_max(D^-1 A) estimates (default 10, 2..200), from the next numeric setup on;
 * builds the graph part of the hierarchy if it does not exist yet.  Too few steps under-estimate lambda_max beyond the 1.1
 * margin and the smoother over-relaxes; pf_cg notices (an AMG-PCG solve still running after 150 iterations), sets the
 * hierarchy up again with three times the steps -- kept from then on --, restarts from the iterate reached and reports it
 * through pf_last_warning.  A tuning knob for meshes whose top eigenvector the ten steps do not reach. */
int pf_amg_set_power_steps(pf_ctx* ctx, int steps);
/* The same V-cycle as a stand-alone operator for drivers with their own Krylov loop (dist.py: one local AMG per
 * rank in an additive Schwarz preconditioner).  prepare: numeric setup for vals (reuse as in PF_PRECOND_AMG_REUSE;
 * PF_E_UNSUPPORTED if the diagonal signs are mixed).  vcycle: z = M^-1 r (device vectors of length nrows, z != r),
 * asynchronous on the ctx stream. */
int pf_amg_prepare(pf_ctx* ctx, const double* vals, int reuse);
/* vals[diagonal entry of dof q] += add[q] (add: nrows doubles): the Schwarz driver completes the interface
 * diagonals of a rank's block so that its local problem is definite */
int pf_add_to_diag(pf_ctx* ctx, double* vals, const double* add);
int pf_amg_vcycle(pf_ctx* ctx, const double* r, double* z);

/* Building blocks for the one-process-per-GPU driver (pyfem_gpu_testflight_amd/dist.py): the
 * single-reduction (Chronopoulos-Gear) Jacobi-PCG.  ONE all-reduce per iteration carries the
 * interface partial sums of w = A z together with gamma = r.z, delta = z.w and r.r.  All calls are
 * asynchronous on the ctx stream; scalars stay on the device.  Vector length is the ctx's nrows.
 * Reductions never use same-address atomics: kernels leave per-block partial sums in the ctx and
 * pf_cgcg_reduce / pf_dot sum them in a fixed order (bitwise reproducible).
 *   pf_dot:             out[0] = sum_i w_i a_i b_i   (w NULL => 1), any n
 *   pf_copy_indexed:    dst[dst_idx[i]] = src[src_idx[i]], i < n (either list NULL = identity):
 *                       packs interface rows into the all-reduce buffer / unpacks the sums
 *   pf_spmv_dot:        y = A x ; partials of x . y
 *   pf_cgcg_update_xrz: alpha = state[0]; x += alpha p; r -= alpha s; z = dinv*r (dinv NULL: z = r);
 *                       partials of sum w r z and sum w r r
 *   pf_cgcg_reduce:     zero[0..nzero) = 0 (the interface part of the all-reduce buffer), then
 *                       red_out[0..2] = [gamma, delta, rr] from the partials of the two calls above
 *   pf_cgcg_update_ps:  state = [alpha, beta, gamma]; from state_in and the REDUCED red = [gamma_new, delta, rr]:
 *                       beta = first ? 0 : gamma_new/gamma; alpha = gamma_new/(delta - beta*gamma_new/alpha);
 *                       p = z + beta p ; s = w + beta s ; the new state goes to state_out (!= state_in)      */
int pf_dot(pf_ctx* ctx, int64_t n, const double* a, const double* b, const double* w, double* out);
int pf_copy_indexed(pf_ctx* ctx, int64_t n, const int32_t* src_idx, const int32_t* dst_idx,
                    const double* src, double* dst);
int pf_spmv_dot(pf_ctx* ctx, const double* vals, const double* x, double* y);
int pf_cgcg_update_xrz(pf_ctx* ctx, const double* state, const double* p, const double* s,
                       const double* dinv, const double* w, double* x, double* r, double* z);
int pf_cgcg_reduce(pf_ctx* ctx, double* red_out, double* zero, int64_t nzero);
int pf_cgcg_update_ps(pf_ctx* ctx, const double* state_in, double* state_out, const double* red,
                      int first, const double* z, const double* w, double* p, double* s);

/* ---- peer-store halo exchange + native distributed CG (one process per GPU of ONE node; no counterpart in the
 *      reference, which is single process).  BASELINE north_star: "all-reduce over xGMI for the CG dot-products and
 *      halo residual sum".  Every rank owns a window in fine-grained device memory that its peers map through HIP
 *      IPC; an exchange is two kernels on the ctx stream: pack + store into the peers' windows + release flag, then
 *      wait (acquire, bounded by a timeout) + sum in ascending rank order (bitwise identical on every sharer).
 *        pf_halo_create:   window for `world` ranks with message slots of cap_doubles doubles (the same on all ranks)
 *        pf_halo_handle:   HOST out, PF_HALO_HANDLE_BYTES bytes: the IPC handle of this rank's window
 *        pf_halo_connect:  map peer `peer`'s window from its handle (all peers must be connected before an exchange)
 *        pf_halo_plan_create (all HOST arrays): neighbours nbr_rank[nnbr] ascending, their packed local indices
 *                          send_idx[nbr_ptr[k]:nbr_ptr[k+1]]; the n_if interface entries if_idx (local vector index)
 *                          with, per entry, the contributing ranks src_rank (ascending, this rank included) and the
 *                          position src_pos of the entry inside that rank's message to this rank (ignored for the own
 *                          rank); returns a plan id
 *        pf_halo_exchange: v[interface entries] <- sum over the sharing ranks (plan >= 0; plan < 0: no vector) and
 *                          scal_out[0..ns) <- sum over all ranks of scal_in[0..ns)  (device pointers, ns <= PF_HALO_NSCAL)
 *        pf_halo_status:   synchronises; HOST out 1 if a wait timed out since creation (then every later exchange
 *                          returns at once and the results are garbage: the caller must raise)
 *        pf_dist_cg:       Chronopoulos-Gear single-reduction Jacobi-PCG over the partition, x0 = 0, entirely in C
 *                          (interface sums of K z and the scalars r.z, z.Kz, r.r per iteration; see the fused
 *                          exchange below); stops at ||r|| <= max(rtol ||b||, atol), read back every check_every
 *                          iterations.  dinv: inverse of the ASSEMBLED diagonal; own_w: 1 on the dofs this rank owns
 *                          (each dof owned once).  PF_E_NOCONV like pf_cg; PF_E_HIP if a peer never arrived.      */
/* ---- the AMG hierarchy's pieces for a DISTRIBUTED finest level (dist.py; counterpart of the reference's pyamg
 *      preconditioner, pyfem.py:2393-2400, when the mesh is split over ranks).  Every rank sets the SAME hierarchy up
 *      with pf_amg_prepare on a context that holds the whole mesh (the numeric setup is ~10 ms; replicating it costs
 *      less than distributing it at these sizes); the finest level's smoothing and residuals then run on each rank's
 *      rows with pf_halo interface sums, and only these four calls touch the replicated part:
 *        pf_amg_level0:       the finest level's two Chebyshev-Jacobi factors per dof (device out, nrows each; may be
 *                             NULL) and the number of level-1 dofs (HOST out)
 *        pf_amg_restrict0:    b1 = P0^T res      (res: nrows, zero outside the caller's owned dofs; b1: level-1 dofs)
 *        pf_amg_coarse_solve: x1 = one V-cycle of levels >= 1 applied to b1 (b1 summed over ranks by the caller)
 *        pf_amg_prolong0:     c += P0 x1         (c: nrows)
 *      pf_vec_smooth: out = x + d .* (b - w) on any vectors of length n (x, d, w may be NULL: 0, 1, 0): the sweeps of
 *      the distributed finest level between two interface sums.  PF_E_UNSUPPORTED: single-level hierarchy.          */
int pf_amg_level0(pf_ctx* ctx, double* dinv1, double* dinv2, int64_t* n1_dofs_host);
int pf_amg_restrict0(pf_ctx* ctx, const double* res, double* b1);
int pf_amg_coarse_solve(pf_ctx* ctx, const double* b1, double* x1);
int pf_amg_prolong0(pf_ctx* ctx, const double* x1, double* c);
int pf_vec_smooth(pf_ctx* ctx, int64_t n, const double* x, const double* d, const double* b, const double* w, double* out);
/* The two transfers over the CALLER'S rows of P0 only, so that a rank does not stream the whole prolongator:
 *   pf_amg_p0_pattern:     HOST out: entries np and coarse nodes nc of P0; DEVICE out (may be NULL): its row pattern
 *                          pptr int32[nnodes+1], pidx int32[np] (coarse node ids, ascending per row)
 *   pf_amg_prolong0_rows:  c_local[i] += P0[grow[i], :] x1 for the nl local nodes whose global ids are grow[i]
 *   pf_amg_restrict0_rows: b1 = sum over the caller's OWNED rows of P0^T res_local, given the transpose pattern of
 *                          those rows: ltptr int32[nc+1], ltnode (local node index, ascending per column), ltslot (the
 *                          entry's index in P0's row-major storage) -- built by the caller from pf_amg_p0_pattern.
 * 2-D problems (scalar or plane: the levels that store a smoothed prolongator).                                      */
int pf_amg_p0_pattern(pf_ctx* ctx, int64_t* np_host, int64_t* nc_host, int32_t* pptr, int32_t* pidx);
int pf_amg_prolong0_rows(pf_ctx* ctx, int64_t nl, const int32_t* grow, const double* x1, double* c_local);
int pf_amg_restrict0_rows(pf_ctx* ctx, const int32_t* ltptr, const int32_t* ltnode, const int32_t* ltslot,
                          const double* res_local, double* b1);
/* The same restriction streamed: pf_amg_p0_gather copies the caller's nlt entries of P0 (ltslot) into ltvals in transpose
 * order (once per numeric setup; nlt * dofs * coarse dofs doubles), pf_amg_restrict0_stream walks (ltptr, ltnode,
 * ltvals) like the single-GPU restriction walks its transposed copy -- 78 instead of 218 us on configuration 5.     */
int pf_amg_p0_gather(pf_ctx* ctx, int64_t nlt, const int32_t* ltslot, double* ltvals);
int pf_amg_restrict0_stream(pf_ctx* ctx, const int32_t* ltptr, const int32_t* ltnode, const double* ltvals,
                            const double* res_local, double* b1);

#define PF_HALO_NSCAL 8
#define PF_HALO_HANDLE_BYTES 64
typedef struct pf_halo pf_halo;
int pf_halo_create(pf_ctx* ctx, int world, int rank, int64_t cap_doubles, pf_halo** out);
int pf_halo_handle(pf_halo* halo, void* handle_out);
int pf_halo_connect(pf_halo* halo, int peer, const void* handle);
int pf_halo_plan_create(pf_halo* halo, int nnbr, const int32_t* nbr_rank, const int32_t* nbr_ptr,
                        const int32_t* send_idx, int64_t n_if, const int32_t* if_idx, const int32_t* src_ptr,
                        const int32_t* src_rank, const int32_t* src_pos, int32_t* plan_id);
int pf_halo_exchange(pf_halo* halo, int plan, double* v, const double* scal_in, int ns, double* scal_out);
int pf_halo_status(pf_halo* halo, int32_t* err_host);
int pf_halo_set_timeout(pf_halo* halo, double seconds);
int pf_halo_destroy(pf_halo* halo);
/* FUSED exchange (round 3): the product kernel stores the rows of interface nodes into the neighbours' windows from its
 * epilogue (interface tiles first, the interior while the rows travel) and the kernel that consumes the product sums
 * them in ascending rank order -- no pack launch, no exchange launch.  pf_dist_cg and the smoother sweeps of
 * pf_dist_pcg_amg run on it.
 *   pf_halo_plan_fuse:     prepare plan `plan` (vectors with the context's dofs per node; 2-D models) for it: per-node
 *                          sharer tables, the tile walk of the product, per-dof owner / interface flags from own_w
 *                          (DEVICE, 1 on the dofs this rank owns).  Idempotent.  All fused plans of one window must
 *                          have the same neighbour list.
 *   pf_halo_fuse_get:      HOST copies of those tables (any output may be NULL); dims = {interface nodes, interface
 *                          tiles, tiles, sharer entries, nodes per tile, dofs per node}.  Tests replay them.
 *   pf_halo_set_colocated: how many ranks of the job run on this rank's device (1 in production; the tests rehearse
 *                          several ranks on one GPU, where waiting kernels must leave room for the peers' kernels)
 *   pf_dist_cg:            TWO launches per iteration: product + send, then scalars / wait / interface sums / fused
 *                          vector update with the convergence test on the device (the result does not depend on
 *                          check_every).  The plan must have been fused.                                            */
int pf_halo_plan_fuse(pf_halo* halo, int plan, const double* own_w);
int pf_halo_fuse_get(pf_halo* halo, int plan, int64_t* dims_host, int32_t* ifslot, int32_t* sh_ptr, int32_t* sh_peer,
                     int32_t* sh_pos, int32_t* tperm, uint8_t* flags8);
int pf_halo_set_colocated(pf_halo* halo, int nranks);
/* The fused plan's tables from HOST arrays, no GPU involved (csrc/pf_hostlogic.h; the CPU tests replay a fused exchange
 * over gloo with them).  Outputs sized by the caller: ifslot nnodes, sh_ptr n_if/d + 1, sh_peer / sh_pos nbr_ptr[nnbr]/d,
 * tperm ceil(nnodes / tile_nodes), flags8 nnodes*d; any may be NULL.  dims = {interface nodes, interface tiles, tiles,
 * sharer entries}.  base_perm (tile walk to refine) and own_w may be NULL. */
int pf_host_fuse_tables_c(int64_t nnodes, int d, int tile_nodes, int nnbr, const int32_t* nbr, const int32_t* nbr_ptr,
                          const int32_t* send_idx, int64_t n_if, const int32_t* if_idx, const int32_t* base_perm,
                          const double* own_w, int32_t* ifslot, int32_t* sh_ptr, int32_t* sh_peer, int32_t* sh_pos,
                          int32_t* tperm, uint8_t* flags8, int64_t* dims);
/* y = K x over the partition through the fused exchange (plan prepared with pf_halo_plan_fuse): product + send, then the
 * interface sums in ascending rank order.  Same result, bit for bit, as pf_spmv followed by pf_halo_exchange. */
int pf_dist_spmv(pf_ctx* ctx, pf_halo* halo, int plan, const double* vals, const double* x, double* y);
int pf_dist_cg(pf_ctx* ctx, pf_halo* halo, int plan, const double* vals, const double* b, const double* dinv,
               const double* own_w, double* x, double rtol, double atol, int maxiter, int check_every,
               int32_t* iters_host, double* relres_host);

/* ---- AMG-PCG over the partition, the whole loop in C (pf_dist.hip).  `local`: the rank's context (K, vectors, halo);
 *      `global`: a context that holds the WHOLE mesh and the hierarchy pf_amg_prepare built for the assembled,
 *      BC-applied matrix (replicated on every rank); both bound to one stream.  plan_dof: the halo plan of the dof
 *      vectors (window `halo`); plan_coarse: a plan on a second window, `halo_coarse`, over the level-1 vector, in which
 *      every rank sends the entries it contributes to to every other rank (dist.py builds both).  d1, d2: the finest level's Chebyshev-Jacobi factors on the rank's dofs
 *      (pf_amg_level0, gathered); grow / ltptr / ltnode / ltslot: as for pf_amg_prolong0_rows / pf_amg_restrict0_rows;
 *      n1: level-1 dofs.  work: 9 nrows_local + 2 n1 + 64 doubles of scratch.  The dof plan must have been prepared
 *      with pf_halo_plan_fuse: every product sends its interface rows from its epilogue.
 *        pf_dist_vcycle:   z = M^-1 r (work: 4 nrows_local + 2 n1 doubles)
 *        pf_dist_pcg_amg:  single-reduction PCG, x0 = 0, stop and errors as pf_dist_cg                                   */
typedef struct {
  pf_ctx* local;
  pf_ctx* global;
  pf_halo* halo;         /* window of the dof-vector exchange (plan_dof) */
  pf_halo* halo_coarse;  /* window of the level-1 exchange (plan_coarse): its messages are ~10x larger */
  int32_t plan_dof, plan_coarse;
  const double* K;
  const double* own_w;
  const double* d1;
  const double* d2;
  int64_t nl;
  const int32_t* grow;
  const int32_t* ltptr;
  const int32_t* ltnode;
  const int32_t* ltslot;
  int64_t n1;
  const float* K32;      /* fp32 copy of K for the cycle's products (pf_vals_to_f32), or NULL: the cycle reads K */
  const double* ltvals;  /* the rank's entries of P0 in transpose order (pf_amg_p0_gather), or NULL: gathered through ltslot */
} pf_dist_amg;
int pf_dist_vcycle(const pf_dist_amg* a, const double* r, double* z, double* work);
/* out[i] = (float) vals[i], i < n (device arrays): the fp32 operator copy the cycle's products read */
int pf_vals_to_f32(pf_ctx* ctx, int64_t n, const double* vals, float* out);
int pf_dist_pcg_amg(const pf_dist_amg* a, const double* b, double* x, double* work, double rtol, double atol,
                    int maxiter, int check_every, int32_t* iters_host, double* relres_host);

/* Measurement helpers of bench.py (pf_time_kernel, pf_hbm_probe) are NOT part of the drop-in boundary: they are declared in
 * pyfem_hip_bench.h and live in their own translation unit (csrc/pf_bench.hip). */

#ifdef __cplusplus
}
#endif
#endif /* PYFEM_HIP_H */
