// AMG-preconditioned PCG over the element-block partition, the whole loop in C (gfx950; no counterpart in the
// reference, which is single process: its solve is pyfem.py:2393-2423, cg + pyamg V-cycle on one matrix).
//
// Round 3: built on the FUSED exchange (pf_halo.hip / pf_solve.hip).  Every product with the rank's matrix is ONE launch
// that also sends the interface rows from its epilogue (interface tiles first); the Chebyshev-Jacobi sweeps and the
// residual are epilogues of that product for all rows that are not on an interface -- exactly the single-GPU cycle's
// launches -- and a small kernel over the interface entries (k_halo_recv_apply, skipped on a rank without neighbours)
// finishes those once the neighbours' rows are there.  The cycle's products read an fp32 copy of the rank's matrix, as the
// single-GPU cycle does (a->K32; NULL keeps fp64).
//   per iteration   V-cycle, k_dot (own r.z), product + send (w = K z, partials of z.w), k_dcg_update (scalars to all ranks,
//                   interface sums of w, p / s / x / r in one pass, convergence test on the device)
//   V-cycle         finest level on the rank's rows: first sweep from zero, then [product+send, finish interface] x 2
//                   (second sweep, residual), restriction of the rows the rank owns, level-1 right-hand side summed over
//                   ranks (packed exchange on the coarse window), levels >= 1 redundantly on the replicated hierarchy
//                   of the whole-mesh context, prolongation to the rank's rows, [product+send, finish] x 2 on the way up.
// Two contexts take part: `local` (the rank's mesh block: K, the vectors, the halo) and `global` (the whole mesh: the
// hierarchy); the caller binds both to one stream.
#include "pf_common.h"

static int dist_vcycle(const pf_dist_amg* a, const double* r, double* z, double* c, double* c2, double* t, double* b1,
                       double* x1) {
  pf_ctx* L = a->local;
  pf_ctx* G = a->global;
  const int64_t n = (int64_t)L->d * L->nnodes;
  int rc;
#define STEP(call) do { if ((rc = (call))) return rc; } while (0)
  // product with the smoother epilogue on the rows off the interface, raw rows on it (sent), then the interface entries
  auto sweep = [&](const double* xin, const double* d, double* out) -> int {
    if (int e = pf_dist_product(L, a->halo, a->plan_dof, a->K, a->K32, xin, out, false, d, r, nullptr)) return e;
    return pf_halo_recv_apply(a->halo, a->plan_dof, xin, d, r, out);
  };
  STEP(pf_vec_smooth(L, n, nullptr, a->d1, r, nullptr, c));            // first sweep from a zero guess
  STEP(sweep(c, a->d2, c2));                                           // second sweep (second Chebyshev factor)
  STEP(sweep(c2, nullptr, t));                                         // residual t = r - K c2
  if (a->ltvals) STEP(pf_amg_restrict0_stream(G, a->ltptr, a->ltnode, a->ltvals, t, b1));
  else STEP(pf_amg_restrict0_rows(G, a->ltptr, a->ltnode, a->ltslot, t, b1));
  if (pf_halo_world(a->halo_coarse) > 1)  // level-1 right-hand side summed over all ranks
    STEP(pf_halo_exchange_impl(a->halo_coarse, a->plan_coarse, b1, nullptr, 0, nullptr));
  STEP(pf_amg_coarse_solve(G, b1, x1));
  STEP(pf_amg_prolong0_rows(G, a->nl, a->grow, x1, c2));               // c2 += P x1
  STEP(sweep(c2, a->d2, c));                                           // post-smoothing in reverse order: symmetric cycle
  STEP(sweep(c, a->d1, z));
#undef STEP
  return PF_OK;
}

extern "C" int pf_dist_vcycle(const pf_dist_amg* a, const double* r, double* z, double* work) {
  if (!a || !a->local) return pf_fail(nullptr, PF_E_INVALID, "pf_dist_vcycle: NULL arguments");
  pf_ctx* ctx = a->local;
  PF_REQUIRE(a->global && a->halo && a->halo_coarse && r && z && work && r != z, "pf_dist_vcycle: NULL argument");
  const int64_t n = (int64_t)ctx->d * ctx->nnodes;
  double *c = work, *c2 = c + n, *t = c2 + 2 * n, *b1 = t + n, *x1 = b1 + a->n1;
  PF_REQUIRE(pf_halo_fuse_of(a->halo, a->plan_dof), "pf_dist_vcycle: the dof plan was not prepared (pf_halo_plan_fuse)");
  return dist_vcycle(a, r, z, c, c2, t, b1, x1);
}

extern "C" int pf_dist_pcg_amg(const pf_dist_amg* a, const double* b, double* x, double* work, double rtol, double atol,
                               int maxiter, int check_every, int32_t* iters_host, double* relres_host) {
  if (!a || !a->local) return pf_fail(nullptr, PF_E_INVALID, "pf_dist_pcg_amg: NULL arguments");
  pf_ctx* ctx = a->local;
  PF_REQUIRE(a->global && a->halo && a->halo_coarse && pf_halo_ctx(a->halo) == ctx && pf_halo_ctx(a->halo_coarse) == ctx,
             "pf_dist_pcg_amg: both halos must belong to the local context");
  PF_REQUIRE(ctx->d && a->global->d == ctx->d && a->K && a->own_w && a->d1 && a->d2 && a->grow && a->ltptr && a->ltnode &&
                 a->ltslot && b && x && work && a->n1 > 0 && a->nl == ctx->nnodes,
             "pf_dist_pcg_amg: incomplete arguments");
  PF_REQUIRE(a->global->stream == ctx->stream, "pf_dist_pcg_amg: bind both contexts to one stream (pf_ctx_set_stream)");
  PF_REQUIRE(pf_halo_fuse_of(a->halo, a->plan_dof), "pf_dist_pcg_amg: the dof plan was not prepared (pf_halo_plan_fuse)");
  PF_HIP(hipSetDevice(ctx->device));
  struct MultiStream {  // an AMG iteration streams several operators: products read their values non-temporally (pf_launch_spmv)
    pf_ctx *l, *g;
    bool wl, wg;
    MultiStream(pf_ctx* l_, pf_ctx* g_) : l(l_), g(g_), wl(l_->multi_stream), wg(g_->multi_stream) { l->multi_stream = g->multi_stream = true; }
    ~MultiStream() { l->multi_stream = wl; g->multi_stream = wg; }
  } multi_stream_guard(a->local, a->global);
  const int64_t n = (int64_t)ctx->d * ctx->nnodes;
  if (maxiter <= 0) maxiter = (int)std::min<int64_t>(10 * n, INT32_MAX);
  if (check_every <= 0) check_every = 1;
  // work: r z w p s | c c2 (unused) t | b1 x1 | 64 small
  double *r = work, *z = r + n, *w = z + n, *p = w + n, *s = p + n;
  double *c = s + n, *c2 = c + n, *t = c2 + 2 * n, *b1 = t + n, *x1 = b1 + a->n1, *small = x1 + a->n1;
  double *stA = small, *stB = small + 8, *red = small + 16, *redg = small + 24;
  hipStream_t st = ctx->stream;
  PF_HIP(hipMemsetAsync(x, 0, sizeof(double) * n, st));
  PF_HIP(hipMemsetAsync(p, 0, sizeof(double) * 2 * n, st));
  PF_HIP(hipMemsetAsync(small, 0, sizeof(double) * 64, st));
  PF_HIP(hipMemcpyAsync(r, b, sizeof(double) * n, hipMemcpyDeviceToDevice, st));
  int rc = pf_dot(ctx, n, b, b, a->own_w, red);
  if (rc) return rc;
  if ((rc = pf_halo_exchange_impl(a->halo, -1, nullptr, red, 1, redg))) return rc;
  double h[8];
  int32_t herr = 0;
  PF_HIP(hipMemcpyAsync(h, redg, sizeof(double), hipMemcpyDeviceToHost, st));
  PF_HIP(hipMemcpyAsync(&herr, pf_halo_err_ptr(a->halo), sizeof(herr), hipMemcpyDeviceToHost, st));
  PF_HIP(hipStreamSynchronize(st));
  if (herr) return pf_fail(ctx, PF_E_HIP, "pf_dist_pcg_amg: a peer did not arrive at the halo exchange (timeout)");
  const double bnorm = sqrt(h[0]);
  const double tol = std::max(rtol * bnorm, atol);
  PF_HIP(hipMemsetAsync(redg, 0, sizeof(double) * 8, st));
  double rnorm = INFINITY;
  int it = 0;
  double *state = stA, *state_next = stB;
  // partials of own r.r for the first reduction (zeroed state: x, r unchanged; the z it writes is overwritten by the cycle)
  if ((rc = pf_dist_init_partials(ctx, state, p, s, nullptr, a->own_w, x, r, z))) return rc;
  // Convergence is tested on the device by every update (redg[3]); the host looks at it once per iteration WITHOUT stalling
  // the device: the copy to pinned memory is queued right behind the update, the next iteration's V-cycle behind that, and
  // only then does the host wait for the copy.  When the residual's recent rate says that this update probably ends the
  // solve, the host waits BEFORE queuing the next cycle instead, so that no cycle is wasted (as in pf_cg).  check_every is
  // accepted for compatibility; the result never depended on it.
  (void)check_every;
  bool stopped = false, pending = false, failed = false;
  double rnorm_prev = INFINITY;
  int herr_state = 0;
  auto queue_look = [&]() -> int {
    if (int e = pf_pin_copy(ctx, 0, redg, sizeof(double) * 5)) return e;
    if (int e = pf_pin_copy(ctx, 8, pf_halo_err_ptr(a->halo), sizeof(int32_t))) return e;
    if (int e = pf_pin_copy(ctx, 9, pf_halo_err_ptr(a->halo_coarse), sizeof(int32_t))) return e;
    return pf_pin_mark(ctx);
  };
  auto look = [&]() -> int {
    if (int e = pf_pin_wait(ctx)) return e;
    int32_t e0, e1;
    memcpy(&e0, ctx->pin + 8, sizeof(e0));
    memcpy(&e1, ctx->pin + 9, sizeof(e1));
    if (e0) { herr_state = 1; failed = true; return PF_OK; }
    if (e1) { herr_state = 2; failed = true; return PF_OK; }
    rnorm_prev = rnorm;
    rnorm = sqrt(ctx->pin[2]);
    if (ctx->pin[3] != 0.0) {  // converged on the device after pin[4] updates
      it = (int)ctx->pin[4];
      stopped = true;
    }
    if (!(rnorm == rnorm)) failed = true;
    return PF_OK;
  };
  while (it < maxiter && !stopped && !failed) {
    if ((rc = dist_vcycle(a, r, z, c, c2, t, b1, x1))) return rc;                              // z = M^-1 r
    if (pending) {  // the previous update's verdict, with this iteration's cycle already queued behind it
      pending = false;
      if ((rc = look())) return rc;
      if (stopped || failed) break;  // (the queued cycle is wasted; its update is not queued, x is the converged iterate)
    }
    if ((rc = pf_dist_dot_gamma(ctx, r, z, a->own_w))) return rc;                              // partials of own r.z
    if ((rc = pf_dist_product(ctx, a->halo, a->plan_dof, a->K, nullptr, z, w, true, nullptr, nullptr, redg + 3))) return rc;
    if ((rc = pf_dist_update(ctx, a->halo, a->plan_dof, w, nullptr, true, true, it == 0, x, r, z, p, s, state, state_next, redg,
                             it == 0, tol * tol, it)))
      return rc;
    std::swap(state, state_next);
    ++it;
    if ((rc = queue_look())) return rc;
    const bool early = rnorm * std::min(1.0, rnorm / rnorm_prev) <= 4.0 * tol || it >= maxiter;
    if (early) {
      if ((rc = look())) return rc;
    } else {
      pending = true;
    }
  }
  if (pending && (rc = look())) return rc;
  if (herr_state == 1) return pf_fail(ctx, PF_E_HIP, "pf_dist_pcg_amg: a peer did not arrive at the fused exchange (timeout)");
  if (herr_state == 2) return pf_fail(ctx, PF_E_HIP, "pf_dist_pcg_amg: a peer did not arrive at the level-1 exchange (timeout)");
  if (iters_host) *iters_host = it;
  if (relres_host) *relres_host = bnorm > 0 ? rnorm / bnorm : rnorm;
  if (!(rnorm <= tol)) return pf_fail(ctx, PF_E_NOCONV, "cg failed with code " + std::to_string(it));  // pyfem.py:2421-2422
  return PF_OK;
}
