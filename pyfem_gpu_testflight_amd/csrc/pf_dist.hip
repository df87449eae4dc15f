// AMG-preconditioned PCG over the element-block partition, the whole loop in C (gfx950; no counterpart in the
// reference, which is single process: its solve is pyfem.py:2393-2423, cg + pyamg V-cycle on one matrix).
//
// dist.py's dist_pcg_amg / ReplicatedHierarchy.vcycle is the same algorithm driven from Python (about 25 C-ABI calls, a
// handful of torch scalar operations and one torch.distributed all-reduce per iteration); here the iteration is nothing
// but kernel launches on the solver's stream, with BOTH exchanges -- the interface sums of the finest level and the
// sum of the level-1 right-hand side over all ranks -- going through the peer-store windows of pf_halo.hip:
//   per iteration   k_cgcg_update_xrz, the V-cycle, k_spmv (+ partials of z.Kz), k_finalize, k_dot, one exchange
//                   (interface sums of w = K z with r.z, z.w, r.r riding along), k_cgcg_update_ps
//   V-cycle         finest level on the rank's rows (four products with interface sums, Chebyshev-Jacobi sweeps as
//                   pf_vec_smooth), restriction of the rows the rank owns (pf_amg_restrict0_rows), level-1 right-hand
//                   side summed over ranks (exchange on the coarse plan), levels >= 1 redundantly on the replicated
//                   hierarchy of the whole-mesh context (pf_amg_coarse_solve), prolongation to the rank's rows.
// Two contexts take part: `local` (the rank's mesh block: K, the vectors, the halo) and `global` (the whole mesh: the
// hierarchy); the caller binds both to one stream.
#include "pf_common.h"

static int dist_vcycle(const pf_dist_amg* a, const double* r, double* z, double* c, double* c2, double* w, double* t,
                       double* b1, double* x1) {
  pf_ctx* L = a->local;
  pf_ctx* G = a->global;
  const int64_t n = (int64_t)L->d * L->nnodes;
  int rc;
#define STEP(call) do { if ((rc = (call))) return rc; } while (0)
  STEP(pf_vec_smooth(L, n, nullptr, a->d1, r, nullptr, c));            // first sweep from a zero guess
  STEP(pf_spmv(L, a->K, c, w));
  STEP(pf_halo_exchange_impl(a->halo, a->plan_dof, w, nullptr, 0, nullptr));
  STEP(pf_vec_smooth(L, n, c, a->d2, r, w, c2));                       // second sweep (second Chebyshev factor)
  STEP(pf_spmv(L, a->K, c2, w));
  STEP(pf_halo_exchange_impl(a->halo, a->plan_dof, w, nullptr, 0, nullptr));
  STEP(pf_vec_smooth(L, n, nullptr, nullptr, r, w, t));                // residual
  STEP(pf_amg_restrict0_rows(G, a->ltptr, a->ltnode, a->ltslot, t, b1));
  STEP(pf_halo_exchange_impl(a->halo_coarse, a->plan_coarse, b1, nullptr, 0, nullptr));  // level-1 right-hand side over all ranks
  STEP(pf_amg_coarse_solve(G, b1, x1));
  STEP(pf_amg_prolong0_rows(G, a->nl, a->grow, x1, c2));               // c2 += P x1
  STEP(pf_spmv(L, a->K, c2, w));
  STEP(pf_halo_exchange_impl(a->halo, a->plan_dof, w, nullptr, 0, nullptr));
  STEP(pf_vec_smooth(L, n, c2, a->d2, r, w, c));                       // post-smoothing in reverse order: symmetric cycle
  STEP(pf_spmv(L, a->K, c, w));
  STEP(pf_halo_exchange_impl(a->halo, a->plan_dof, w, nullptr, 0, nullptr));
  STEP(pf_vec_smooth(L, n, c, a->d1, r, w, z));
#undef STEP
  return PF_OK;
}

extern "C" int pf_dist_vcycle(const pf_dist_amg* a, const double* r, double* z, double* work) {
  if (!a || !a->local) return pf_fail(nullptr, PF_E_INVALID, "pf_dist_vcycle: NULL arguments");
  pf_ctx* ctx = a->local;
  PF_REQUIRE(a->global && a->halo && a->halo_coarse && r && z && work && r != z, "pf_dist_vcycle: NULL argument");
  const int64_t n = (int64_t)ctx->d * ctx->nnodes;
  double *c = work, *c2 = c + n, *w = c2 + n, *t = w + n, *b1 = t + n, *x1 = b1 + a->n1;
  return dist_vcycle(a, r, z, c, c2, w, t, b1, x1);
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
  PF_HIP(hipSetDevice(ctx->device));
  const int64_t n = (int64_t)ctx->d * ctx->nnodes;
  if (maxiter <= 0) maxiter = (int)std::min<int64_t>(10 * n, INT32_MAX);
  if (check_every <= 0) check_every = 1;
  // work: r z w p s | c c2 wv t | b1 x1 | 64 small
  double *r = work, *z = r + n, *w = z + n, *p = w + n, *s = p + n;
  double *c = s + n, *c2 = c + n, *wv = c2 + n, *t = wv + n, *b1 = t + n, *x1 = b1 + a->n1, *small = x1 + a->n1;
  double *stA = small, *stB = small + 8, *red = small + 16, *redg = small + 24, *g = small + 32;
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
  double rnorm = INFINITY;
  int it = 0;
  double *state = stA, *state_next = stB;
  while (it < maxiter) {
    if ((rc = pf_cgcg_update_xrz(ctx, state, p, s, nullptr, a->own_w, x, r, t))) return rc;  // x += alpha p; r -= alpha s
    if ((rc = dist_vcycle(a, r, z, c, c2, wv, t, b1, x1))) return rc;                          // z = M^-1 r
    if ((rc = pf_spmv_dot(ctx, a->K, z, w))) return rc;                                        // w = K z, partials of z.w
    if ((rc = pf_cgcg_reduce(ctx, red, nullptr, 0))) return rc;                                // [., delta, rr]
    if ((rc = pf_dot(ctx, n, r, z, a->own_w, g))) return rc;                                   // gamma = r.z over owned dofs
    PF_HIP(hipMemcpyAsync(red, g, sizeof(double), hipMemcpyDeviceToDevice, st));
    if ((rc = pf_halo_exchange_impl(a->halo, a->plan_dof, w, red, 3, redg))) return rc;
    if (it % check_every == 0 || it + 1 == maxiter) {
      PF_HIP(hipMemcpyAsync(h, redg, sizeof(double) * 3, hipMemcpyDeviceToHost, st));
      PF_HIP(hipMemcpyAsync(&herr, pf_halo_err_ptr(a->halo), sizeof(herr), hipMemcpyDeviceToHost, st));
      PF_HIP(hipStreamSynchronize(st));
      if (herr) return pf_fail(ctx, PF_E_HIP, "pf_dist_pcg_amg: a peer did not arrive at the halo exchange (timeout)");
      PF_HIP(hipMemcpyAsync(&herr, pf_halo_err_ptr(a->halo_coarse), sizeof(herr), hipMemcpyDeviceToHost, st));
      PF_HIP(hipStreamSynchronize(st));
      if (herr) return pf_fail(ctx, PF_E_HIP, "pf_dist_pcg_amg: a peer did not arrive at the level-1 exchange (timeout)");
      rnorm = sqrt(h[2]);
      if (!(rnorm == rnorm) || rnorm <= tol) break;
    }
    if ((rc = pf_cgcg_update_ps(ctx, state, state_next, redg, it == 0, z, w, p, s))) return rc;
    std::swap(state, state_next);
    ++it;
  }
  if (iters_host) *iters_host = it;
  if (relres_host) *relres_host = bnorm > 0 ? rnorm / bnorm : rnorm;
  if (!(rnorm <= tol)) return pf_fail(ctx, PF_E_NOCONV, "cg failed with code " + std::to_string(it));  // pyfem.py:2421-2422
  return PF_OK;
}
