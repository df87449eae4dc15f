// Dirichlet BCs, SpMV and Jacobi-PCG on the blocked value layout (gfx950).
//
// Replaces (reference file:line)
//   ModelBase.apply_dirichlet_bcs                pyfem.py:781-835
//   scipy CSR matvec call sites                   pyfem.py:2119 (Helmholtz rhs = R x), inside cg
//   Assembler._solve_linear_system "cg"           pyfem.py:2403-2423  (scipy.sparse.linalg.cg;
//       the pyamg SA-AMG preconditioner of :2393-2400 is replaced by Jacobi -- see DESIGN.md)
//
// SpMV design: HBM-bound streaming of the value array (8 B/nnz) plus one 4-byte node index per
// d x d block (so 9 B/nnz at d=2 instead of CSR's 12).  One wave owns 64 consecutive nodes whose
// blocks are ONE contiguous range: lanes stream that range block-per-lane (32 B values + 4 B index,
// fully coalesced), gather x[node] as one 16-byte load, and drop the d-vector product into an LDS
// strip; then lane-per-node sums its row's strip entries (conflict-free: consecutive rows start
// 9 x 16 B apart) and writes y coalesced.  Waves are persistent (grid-stride over 64-node tiles)
// so p.Ap for CG costs one atomic per wave.
#include "pf_common.h"

static constexpr int SPMV_TILE = 64;       // nodes per wave tile
static constexpr int SPMV_CHUNK = 1024;    // blocks staged in LDS per pass (d=2: 16 KB)

template <int D>
struct Vec;
template <>
struct Vec<1> {
  using T = double;
};
template <>
struct Vec<2> {
  using T = double2;
};

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// y = A x ; optionally dot_out[0] += sum_i w_i x_i y_i   (w == nullptr -> 1)
// zero_a / zero_b: device scalars this launch clears for the NEXT kernel in the CG pipeline
template <int D>
__global__ __launch_bounds__(64) void k_spmv(int64_t nnodes, const int32_t* __restrict__ nptr,
                                             const int32_t* __restrict__ nidx, const double* __restrict__ vals,
                                             const double* __restrict__ x, double* __restrict__ y,
                                             double* __restrict__ dot_out, const double* __restrict__ dot_w,
                                             double* __restrict__ zero_a, double* __restrict__ zero_b) {
  using V = typename Vec<D>::T;
  __shared__ V prod[SPMV_CHUNK];
  const int lane = threadIdx.x;
  if (blockIdx.x == 0 && lane == 0) {
    if (zero_a) *zero_a = 0.0;
    if (zero_b) *zero_b = 0.0;
  }
  double dacc = 0.0;
  const int64_t ntiles = (nnodes + SPMV_TILE - 1) / SPMV_TILE;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t n = tile * SPMV_TILE + lane;
    const bool valid = n < nnodes;
    const int32_t rb = valid ? nptr[n] : 0;
    const int32_t re = valid ? nptr[n + 1] : 0;
    const int32_t base = __shfl(rb, 0);
    const int64_t nlast = min(nnodes, (tile + 1) * SPMV_TILE);
    const int32_t end = nptr[nlast];  // uniform
    V acc;
    if constexpr (D == 1) acc = 0.0; else acc = make_double2(0.0, 0.0);
    for (int32_t c0 = base; c0 < end; c0 += SPMV_CHUNK) {
      const int32_t c1 = min(end, c0 + SPMV_CHUNK);
      // phase A: stream blocks [c0, c1)
      for (int32_t g = c0 + lane; g < c1; g += 64) {
        const int32_t m = nidx[g];
        if constexpr (D == 1) {
          prod[g - c0] = vals[g] * x[m];
        } else {
          const double2 k0 = *reinterpret_cast<const double2*>(vals + 4 * (int64_t)g);
          const double2 k1 = *reinterpret_cast<const double2*>(vals + 4 * (int64_t)g + 2);
          const double2 xv = *reinterpret_cast<const double2*>(x + 2 * (int64_t)m);
          prod[g - c0] = make_double2(k0.x * xv.x + k0.y * xv.y, k1.x * xv.x + k1.y * xv.y);
        }
      }
      __syncthreads();
      // phase B: lane-per-node partial row sums over this chunk
      const int32_t lo = max(rb, c0), hi = min(re, c1);
      for (int32_t g = lo; g < hi; ++g) {
        const V pv = prod[g - c0];
        if constexpr (D == 1) acc += pv; else { acc.x += pv.x; acc.y += pv.y; }
      }
      __syncthreads();
    }
    if (valid) {
      if constexpr (D == 1) {
        y[n] = acc;
        if (dot_out) dacc += (dot_w ? dot_w[n] : 1.0) * x[n] * acc;
      } else {
        *reinterpret_cast<double2*>(y + 2 * n) = acc;
        if (dot_out) {
          const double2 xv = *reinterpret_cast<const double2*>(x + 2 * n);
          if (dot_w) {
            const double2 wv = *reinterpret_cast<const double2*>(dot_w + 2 * n);
            dacc += wv.x * xv.x * acc.x + wv.y * xv.y * acc.y;
          } else {
            dacc += xv.x * acc.x + xv.y * acc.y;
          }
        }
      }
    }
  }
  if (dot_out) {
    dacc = wave_sum(dacc);
    if (lane == 0) unsafeAtomicAdd(dot_out, dacc);
  }
}

// ------------------------------------------------------------------------------------------
template <int D>
__global__ void k_extract_diag(int64_t nnodes, const int32_t* __restrict__ nptr, const uint8_t* __restrict__ diagpos,
                               const double* __restrict__ vals, double* __restrict__ diag, int invert) {
  const int64_t n = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (n >= nnodes) return;
  const int64_t b = ((int64_t)nptr[n] + diagpos[n]) * D * D;
#pragma unroll
  for (int k = 0; k < D; ++k) {
    const double v = vals[b + k * D + k];
    diag[D * n + k] = invert ? 1.0 / v : v;
  }
}

// apply_dirichlet_bcs (pyfem.py:781-835); one lane owns the d rows of one node
template <int D>
__global__ void k_apply_dirichlet(int64_t nnodes, const int32_t* __restrict__ nptr, const int32_t* __restrict__ nidx,
                                  double* __restrict__ vals, double* __restrict__ rhs,
                                  const uint8_t* __restrict__ mask, const double* __restrict__ fvals, int symmetric,
                                  const double* __restrict__ diag_scale) {
  const int64_t n = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (n >= nnodes) return;
  bool fr[D];
  double acc[D];
#pragma unroll
  for (int k = 0; k < D; ++k) {
    fr[k] = mask[D * n + k] != 0;
    acc[k] = 0.0;
  }
  for (int32_t g = nptr[n]; g < nptr[n + 1]; ++g) {
    const int64_t m = nidx[g];
#pragma unroll
    for (int l = 0; l < D; ++l) {
      const bool fc = mask[D * m + l] != 0;
      const double cv = (fc && fvals) ? fvals[D * m + l] : 0.0;
#pragma unroll
      for (int k = 0; k < D; ++k) {
        double* kp = vals + (int64_t)g * D * D + k * D + l;
        if (fr[k]) {
          // zeroed row, unit diagonal (pyfem.py:811-812, 822-823)
          *kp = (m == n && k == l) ? (diag_scale ? diag_scale[D * n + k] : 1.0) : 0.0;
        } else if (fc && symmetric) {
          acc[k] += *kp * cv;  // Krb . u0  (pyfem.py:806-807, 833-834)
          *kp = 0.0;           // zeroed column (pyfem.py:815-819)
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < D; ++k) {
    if (fr[k]) rhs[D * n + k] = fvals ? fvals[D * n + k] : 0.0;  // pyfem.py:829-832
    else if (symmetric && fvals) rhs[D * n + k] -= acc[k];
  }
}

// ------------------------------------------------------------------------------------------
// vector kernels (grid-stride, double2 vectorised where n is even)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_dot(int64_t n, const double* __restrict__ a, const double* __restrict__ b,
                                             const double* __restrict__ w, double* __restrict__ out) {
  double s = 0.0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    s += (w ? w[i] : 1.0) * a[i] * b[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) unsafeAtomicAdd(out, s);
}

// alpha = rz/pAp; x += alpha p; r -= alpha Ap; z = dinv r; rz_new += w r z; rr += w r r
__global__ __launch_bounds__(256) void k_cg_update_xr(int64_t n, const double* __restrict__ rz,
                                                      const double* __restrict__ pAp, const double* __restrict__ p,
                                                      const double* __restrict__ Ap, const double* __restrict__ dinv,
                                                      const double* __restrict__ w, double* __restrict__ x,
                                                      double* __restrict__ r, double* __restrict__ z,
                                                      double* __restrict__ rz_new, double* __restrict__ rr) {
  const double alpha = rz[0] / pAp[0];
  double s1 = 0.0, s2 = 0.0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    x[i] += alpha * p[i];
    const double ri = r[i] - alpha * Ap[i];
    r[i] = ri;
    const double zi = dinv ? dinv[i] * ri : ri;
    z[i] = zi;
    const double wi = w ? w[i] : 1.0;
    s1 += wi * ri * zi;
    s2 += wi * ri * ri;
  }
  s1 = wave_sum(s1);
  s2 = wave_sum(s2);
  if ((threadIdx.x & 63) == 0) {
    unsafeAtomicAdd(rz_new, s1);
    unsafeAtomicAdd(rr, s2);
  }
}

// beta = rz_new/rz_old; p = z + beta p ; clears `zero_a` for the next SpMV's p.Ap
__global__ __launch_bounds__(256) void k_cg_update_p(int64_t n, const double* __restrict__ rz_new,
                                                     const double* __restrict__ rz_old, const double* __restrict__ z,
                                                     double* __restrict__ p, double* __restrict__ zero_a) {
  const double beta = rz_new[0] / rz_old[0];
  if (zero_a && blockIdx.x == 0 && threadIdx.x == 0) *zero_a = 0.0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    p[i] = z[i] + beta * p[i];
}

// r = b - Ap ; z = dinv r ; p = z ; rz += r z ; bb += b b
__global__ __launch_bounds__(256) void k_cg_init(int64_t n, const double* __restrict__ b, const double* __restrict__ Ap,
                                                 const double* __restrict__ dinv, double* __restrict__ r,
                                                 double* __restrict__ z, double* __restrict__ p,
                                                 double* __restrict__ rz, double* __restrict__ bb,
                                                 double* __restrict__ rr) {
  double s1 = 0.0, s2 = 0.0, s3 = 0.0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const double bi = b[i];
    const double ri = bi - Ap[i];
    r[i] = ri;
    const double zi = dinv ? dinv[i] * ri : ri;
    z[i] = zi;
    p[i] = zi;
    s1 += ri * zi;
    s2 += bi * bi;
    s3 += ri * ri;
  }
  s1 = wave_sum(s1);
  s2 = wave_sum(s2);
  s3 = wave_sum(s3);
  if ((threadIdx.x & 63) == 0) {
    unsafeAtomicAdd(rz, s1);
    unsafeAtomicAdd(bb, s2);
    unsafeAtomicAdd(rr, s3);
  }
}

__global__ void k_gather(int64_t n, const int32_t* __restrict__ idx, const double* __restrict__ src,
                         double* __restrict__ dst) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[idx[i]];
}

__global__ void k_scatter_add(int64_t n, const int32_t* __restrict__ idx, const double* __restrict__ src,
                              double* __restrict__ dst) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) dst[idx[i]] += src[i];  // idx entries are unique within one call
}

__global__ void k_scatter_set(int64_t n, const int32_t* __restrict__ idx, const double* __restrict__ src,
                              double* __restrict__ dst) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) dst[idx[i]] = src[i];
}

__global__ void k_copy_indexed(int64_t n, const int32_t* __restrict__ si, const int32_t* __restrict__ di,
                               const double* __restrict__ src, double* __restrict__ dst) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) dst[di ? di[i] : i] = src[si ? si[i] : i];
}

// ---- single-reduction (Chronopoulos-Gear) PCG steps; state = [alpha, beta, gamma] ----------
__global__ __launch_bounds__(256) void k_cgcg_update_xrz(int64_t n, const double* __restrict__ state,
                                                         const double* __restrict__ p, const double* __restrict__ s,
                                                         const double* __restrict__ dinv, const double* __restrict__ w,
                                                         double* __restrict__ x, double* __restrict__ r,
                                                         double* __restrict__ z, double* __restrict__ gamma_out,
                                                         double* __restrict__ rr_out) {
  const double alpha = state[0];
  double s1 = 0.0, s2 = 0.0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    x[i] += alpha * p[i];
    const double ri = r[i] - alpha * s[i];
    r[i] = ri;
    const double zi = dinv ? dinv[i] * ri : ri;
    z[i] = zi;
    const double wi = w ? w[i] : 1.0;
    s1 += wi * ri * zi;
    s2 += wi * ri * ri;
  }
  s1 = wave_sum(s1);
  s2 = wave_sum(s2);
  if ((threadIdx.x & 63) == 0) {
    unsafeAtomicAdd(gamma_out, s1);
    unsafeAtomicAdd(rr_out, s2);
  }
}

__global__ void k_cgcg_scalars(double* __restrict__ state, const double* __restrict__ red, int first) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const double gamma_new = red[0], delta = red[1];
  double alpha, beta;
  if (first) {
    beta = 0.0;
    alpha = gamma_new / delta;
  } else {
    beta = gamma_new / state[2];
    alpha = gamma_new / (delta - beta * gamma_new / state[0]);
  }
  state[0] = alpha;
  state[1] = beta;
  state[2] = gamma_new;
}

__global__ __launch_bounds__(256) void k_cgcg_update_ps(int64_t n, const double* __restrict__ state,
                                                        const double* __restrict__ z, const double* __restrict__ w,
                                                        double* __restrict__ p, double* __restrict__ s) {
  const double beta = state[1];
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    p[i] = z[i] + beta * p[i];
    s[i] = w[i] + beta * s[i];
  }
}

// ------------------------------------------------------------------------------------------
static inline unsigned vec_grid(int64_t n) { return (unsigned)std::min<int64_t>(cdiv(n, 256), 256 * 8); }
static inline unsigned spmv_grid(int64_t nnodes) { return (unsigned)std::min<int64_t>(cdiv(nnodes, SPMV_TILE), 256 * 16); }

static int launch_spmv_full(pf_ctx* ctx, const double* vals, const double* x, double* y, double* dot_out,
                            const double* dot_w, double* za, double* zb) {
  const unsigned grid = spmv_grid(ctx->nnodes);
  if (ctx->d == 1)
    k_spmv<1><<<grid, 64, 0, ctx->stream>>>(ctx->nnodes, ctx->nptr, ctx->nidx, vals, x, y, dot_out, dot_w, za, zb);
  else
    k_spmv<2><<<grid, 64, 0, ctx->stream>>>(ctx->nnodes, ctx->nptr, ctx->nidx, vals, x, y, dot_out, dot_w, za, zb);
  PF_LAUNCH_CHECK();
  return PF_OK;
}

int pf_launch_spmv(pf_ctx* ctx, const double* vals, const double* x, double* y, double* dot_out, const double* dot_w) {
  return launch_spmv_full(ctx, vals, x, y, dot_out, dot_w, nullptr, nullptr);
}

extern "C" int pf_spmv(pf_ctx* ctx, const double* vals, const double* x, double* y) {
  if (!ctx) return pf_fail(nullptr, PF_E_INVALID, "ctx is NULL");
  PF_REQUIRE(ctx->d, "pf_spmv: no pattern built");
  PF_REQUIRE(vals && x && y && x != y, "pf_spmv: NULL argument or x aliases y");
  return pf_launch_spmv(ctx, vals, x, y, nullptr, nullptr);
}

extern "C" int pf_extract_diag(pf_ctx* ctx, const double* vals, double* diag) {
  if (!ctx) return pf_fail(nullptr, PF_E_INVALID, "ctx is NULL");
  PF_REQUIRE(ctx->d && vals && diag, "pf_extract_diag: no pattern / NULL argument");
  const unsigned g = (unsigned)cdiv(ctx->nnodes, 256);
  if (ctx->d == 1) k_extract_diag<1><<<g, 256, 0, ctx->stream>>>(ctx->nnodes, ctx->nptr, ctx->diagpos, vals, diag, 0);
  else k_extract_diag<2><<<g, 256, 0, ctx->stream>>>(ctx->nnodes, ctx->nptr, ctx->diagpos, vals, diag, 0);
  PF_LAUNCH_CHECK();
  return PF_OK;
}

extern "C" int pf_apply_dirichlet(pf_ctx* ctx, double* Kvals, double* rhs, const uint8_t* fixed_mask,
                                  const double* fixed_vals, int symmetric, const double* diag_scale) {
  if (!ctx) return pf_fail(nullptr, PF_E_INVALID, "ctx is NULL");
  PF_REQUIRE(ctx->d && Kvals && rhs && fixed_mask, "pf_apply_dirichlet: no pattern / NULL argument");
  const unsigned g = (unsigned)cdiv(ctx->nnodes, 128);
  if (ctx->d == 1)
    k_apply_dirichlet<1><<<g, 128, 0, ctx->stream>>>(ctx->nnodes, ctx->nptr, ctx->nidx, Kvals, rhs, fixed_mask, fixed_vals, symmetric, diag_scale);
  else
    k_apply_dirichlet<2><<<g, 128, 0, ctx->stream>>>(ctx->nnodes, ctx->nptr, ctx->nidx, Kvals, rhs, fixed_mask, fixed_vals, symmetric, diag_scale);
  PF_LAUNCH_CHECK();
  return PF_OK;
}

extern "C" int pf_dot(pf_ctx* ctx, int64_t n, const double* a, const double* b, const double* w, double* out) {
  if (!ctx) return pf_fail(nullptr, PF_E_INVALID, "ctx is NULL");
  PF_REQUIRE(a && b && out && n >= 0, "pf_dot: NULL argument");
  PF_HIP(hipMemsetAsync(out, 0, sizeof(double), ctx->stream));
  if (n) k_dot<<<vec_grid(n), 256, 0, ctx->stream>>>(n, a, b, w, out);
  PF_LAUNCH_CHECK();
  return PF_OK;
}

extern "C" int pf_cg_update_xr(pf_ctx* ctx, int64_t n, const double* rz, const double* pAp, const double* p,
                               const double* Ap, const double* dinv, const double* w, double* x, double* r, double* z,
                               double* rz_new, double* rr) {
  if (!ctx) return pf_fail(nullptr, PF_E_INVALID, "ctx is NULL");
  PF_REQUIRE(rz && pAp && p && Ap && x && r && z && rz_new && rr, "pf_cg_update_xr: NULL argument");
  PF_HIP(hipMemsetAsync(rz_new, 0, sizeof(double), ctx->stream));
  PF_HIP(hipMemsetAsync(rr, 0, sizeof(double), ctx->stream));
  k_cg_update_xr<<<vec_grid(n), 256, 0, ctx->stream>>>(n, rz, pAp, p, Ap, dinv, w, x, r, z, rz_new, rr);
  PF_LAUNCH_CHECK();
  return PF_OK;
}

extern "C" int pf_cg_update_p(pf_ctx* ctx, int64_t n, const double* rz_new, const double* rz_old, const double* z,
                              double* p) {
  if (!ctx) return pf_fail(nullptr, PF_E_INVALID, "ctx is NULL");
  PF_REQUIRE(rz_new && rz_old && z && p, "pf_cg_update_p: NULL argument");
  k_cg_update_p<<<vec_grid(n), 256, 0, ctx->stream>>>(n, rz_new, rz_old, z, p, nullptr);
  PF_LAUNCH_CHECK();
  return PF_OK;
}

extern "C" int pf_gather(pf_ctx* ctx, int64_t n, const int32_t* idx, const double* src, double* dst) {
  if (!ctx) return pf_fail(nullptr, PF_E_INVALID, "ctx is NULL");
  if (n == 0) return PF_OK;
  PF_REQUIRE(idx && src && dst, "pf_gather: NULL argument");
  k_gather<<<(unsigned)cdiv(n, 256), 256, 0, ctx->stream>>>(n, idx, src, dst);
  PF_LAUNCH_CHECK();
  return PF_OK;
}

extern "C" int pf_scatter_add(pf_ctx* ctx, int64_t n, const int32_t* idx, const double* src, double* dst) {
  if (!ctx) return pf_fail(nullptr, PF_E_INVALID, "ctx is NULL");
  if (n == 0) return PF_OK;
  PF_REQUIRE(idx && src && dst, "pf_scatter_add: NULL argument");
  k_scatter_add<<<(unsigned)cdiv(n, 256), 256, 0, ctx->stream>>>(n, idx, src, dst);
  PF_LAUNCH_CHECK();
  return PF_OK;
}

extern "C" int pf_scatter_set(pf_ctx* ctx, int64_t n, const int32_t* idx, const double* src, double* dst) {
  if (!ctx) return pf_fail(nullptr, PF_E_INVALID, "ctx is NULL");
  if (n == 0) return PF_OK;
  PF_REQUIRE(idx && src && dst, "pf_scatter_set: NULL argument");
  k_scatter_set<<<(unsigned)cdiv(n, 256), 256, 0, ctx->stream>>>(n, idx, src, dst);
  PF_LAUNCH_CHECK();
  return PF_OK;
}

extern "C" int pf_copy_indexed(pf_ctx* ctx, int64_t n, const int32_t* src_idx, const int32_t* dst_idx,
                               const double* src, double* dst) {
  if (!ctx) return pf_fail(nullptr, PF_E_INVALID, "ctx is NULL");
  if (n == 0) return PF_OK;
  PF_REQUIRE(src && dst, "pf_copy_indexed: NULL argument");
  k_copy_indexed<<<(unsigned)cdiv(n, 256), 256, 0, ctx->stream>>>(n, src_idx, dst_idx, src, dst);
  PF_LAUNCH_CHECK();
  return PF_OK;
}

extern "C" int pf_spmv_dot(pf_ctx* ctx, const double* vals, const double* x, double* y, double* dot_out) {
  if (!ctx) return pf_fail(nullptr, PF_E_INVALID, "ctx is NULL");
  PF_REQUIRE(ctx->d, "pf_spmv_dot: no pattern built");
  PF_REQUIRE(vals && x && y && dot_out && x != y, "pf_spmv_dot: NULL argument or x aliases y");
  return pf_launch_spmv(ctx, vals, x, y, dot_out, nullptr);
}

extern "C" int pf_cgcg_update_xrz(pf_ctx* ctx, int64_t n, const double* state, const double* p, const double* s,
                                  const double* dinv, const double* w, double* x, double* r, double* z,
                                  double* gamma_out, double* rr_out) {
  if (!ctx) return pf_fail(nullptr, PF_E_INVALID, "ctx is NULL");
  PF_REQUIRE(state && p && s && x && r && z && gamma_out && rr_out, "pf_cgcg_update_xrz: NULL argument");
  k_cgcg_update_xrz<<<vec_grid(n), 256, 0, ctx->stream>>>(n, state, p, s, dinv, w, x, r, z, gamma_out, rr_out);
  PF_LAUNCH_CHECK();
  return PF_OK;
}

extern "C" int pf_cgcg_scalars(pf_ctx* ctx, double* state, const double* red, int first) {
  if (!ctx) return pf_fail(nullptr, PF_E_INVALID, "ctx is NULL");
  PF_REQUIRE(state && red, "pf_cgcg_scalars: NULL argument");
  k_cgcg_scalars<<<1, 64, 0, ctx->stream>>>(state, red, first);
  PF_LAUNCH_CHECK();
  return PF_OK;
}

extern "C" int pf_cgcg_update_ps(pf_ctx* ctx, int64_t n, const double* state, const double* z, const double* w,
                                 double* p, double* s) {
  if (!ctx) return pf_fail(nullptr, PF_E_INVALID, "ctx is NULL");
  PF_REQUIRE(state && z && w && p && s, "pf_cgcg_update_ps: NULL argument");
  k_cgcg_update_ps<<<vec_grid(n), 256, 0, ctx->stream>>>(n, state, z, w, p, s);
  PF_LAUNCH_CHECK();
  return PF_OK;
}

// ------------------------------------------------------------------------------------------
// native PCG loop: three launches per iteration, scalars stay on the device, the host only
// reads ||r||^2 back every `check_every` iterations.
// scal layout: [0] rz_a  [1] rz_b  [2] pAp  [3] rr  [4] bb
// ------------------------------------------------------------------------------------------
extern "C" int pf_cg(pf_ctx* ctx, const double* vals, const double* b, double* x, double rtol, double atol,
                     int maxiter, int precond, int check_every, int32_t* iters_host, double* relres_host) {
  if (!ctx) return pf_fail(nullptr, PF_E_INVALID, "ctx is NULL");
  PF_REQUIRE(ctx->d && vals && b && x, "pf_cg: no pattern / NULL argument");
  PF_REQUIRE(precond == PF_PRECOND_NONE || precond == PF_PRECOND_JACOBI, "pf_cg: unknown preconditioner");
  PF_HIP(hipSetDevice(ctx->device));
  const int64_t n = (int64_t)ctx->d * ctx->nnodes;
  if (maxiter <= 0) maxiter = (int)std::min<int64_t>(10 * n, INT32_MAX);  // scipy's default
  if (check_every <= 0) check_every = 50;
  if (ctx->cgw_cap < 5 * n) {
    pf_free(ctx->cgw);
    PF_HIP(hipMalloc(&ctx->cgw, sizeof(double) * 5 * n));
    ctx->cgw_cap = 5 * n;
  }
  double *r = ctx->cgw, *z = r + n, *p = z + n, *Ap = p + n, *dinv = Ap + n;
  double* S = ctx->scal;
  hipStream_t st = ctx->stream;
  const unsigned vg = vec_grid(n);

  if (precond == PF_PRECOND_JACOBI) {
    const unsigned g = (unsigned)cdiv(ctx->nnodes, 256);
    if (ctx->d == 1) k_extract_diag<1><<<g, 256, 0, st>>>(ctx->nnodes, ctx->nptr, ctx->diagpos, vals, dinv, 1);
    else k_extract_diag<2><<<g, 256, 0, st>>>(ctx->nnodes, ctx->nptr, ctx->diagpos, vals, dinv, 1);
    PF_LAUNCH_CHECK();
  } else {
    dinv = nullptr;
  }
  PF_HIP(hipMemsetAsync(S, 0, 8 * sizeof(double), st));
  int rc = launch_spmv_full(ctx, vals, x, Ap, nullptr, nullptr, nullptr, nullptr);  // Ap = A x0
  if (rc) return rc;
  k_cg_init<<<vg, 256, 0, st>>>(n, b, Ap, dinv, r, z, p, S + 0, S + 4, S + 3);
  PF_LAUNCH_CHECK();
  double h[5];
  PF_HIP(hipMemcpyAsync(h, S, sizeof(h), hipMemcpyDeviceToHost, st));
  PF_HIP(hipStreamSynchronize(st));
  const double bnorm = sqrt(h[4]);
  const double tol = std::max(rtol * bnorm, atol);
  double rnorm = sqrt(h[3]);
  int it = 0;
  int cur = 0;  // index of the live rz scalar
  while (rnorm > tol && it < maxiter) {
    const int burst = std::min(check_every, maxiter - it);
    for (int k = 0; k < burst; ++k) {
      double *rz_old = S + cur, *rz_new = S + (cur ^ 1);
      // Ap = A p, pAp = p.Ap ; clears rz_new and rr for the update kernel
      rc = launch_spmv_full(ctx, vals, p, Ap, S + 2, nullptr, rz_new, S + 3);
      if (rc) return rc;
      k_cg_update_xr<<<vg, 256, 0, st>>>(n, rz_old, S + 2, p, Ap, dinv, nullptr, x, r, z, rz_new, S + 3);
      k_cg_update_p<<<vg, 256, 0, st>>>(n, rz_new, rz_old, z, p, S + 2);
      cur ^= 1;
    }
    PF_LAUNCH_CHECK();
    it += burst;
    PF_HIP(hipMemcpyAsync(h, S, sizeof(h), hipMemcpyDeviceToHost, st));
    PF_HIP(hipStreamSynchronize(st));
    rnorm = sqrt(h[3]);
    if (!(rnorm == rnorm)) break;  // NaN: breakdown
  }
  if (iters_host) *iters_host = it;
  if (relres_host) *relres_host = bnorm > 0 ? rnorm / bnorm : rnorm;
  if (!(rnorm <= tol))
    return pf_fail(ctx, PF_E_NOCONV, "cg failed with code " + std::to_string(it));  // pyfem.py:2421-2422
  return PF_OK;
}

// ------------------------------------------------------------------------------------------
extern "C" int pf_time_kernel(pf_ctx* ctx, int which, int reps, const double* in0, const double* in1, double* out0,
                              double E, double nu, double p, double* ms_host) {
  if (!ctx) return pf_fail(nullptr, PF_E_INVALID, "ctx is NULL");
  PF_REQUIRE(ctx->d && reps > 0 && ms_host, "pf_time_kernel: bad arguments");
  hipEvent_t e0, e1;
  PF_HIP(hipEventCreate(&e0));
  PF_HIP(hipEventCreate(&e1));
  int rc = PF_OK;
  for (int pass = 0; pass < 2 && rc == PF_OK; ++pass) {  // pass 0 = warm-up
    const int nrep = pass ? reps : 1;
    PF_HIP(hipEventRecord(e0, ctx->stream));
    for (int i = 0; i < nrep && rc == PF_OK; ++i) {
      if (which == PF_K_ASSEMBLE_ELASTICITY) rc = pf_launch_assemble_elasticity(ctx, E, nu, p, in0, PF_SCATTER_GATHER, out0);
      else if (which == PF_K_SPMV) rc = pf_launch_spmv(ctx, in0, in1, out0, nullptr, nullptr);
      else if (which == PF_K_ASSEMBLE_HELMHOLTZ) rc = pf_launch_assemble_helmholtz(ctx, E, out0, nullptr);
      else rc = pf_fail(ctx, PF_E_INVALID, "pf_time_kernel: unknown kernel");
    }
    PF_HIP(hipEventRecord(e1, ctx->stream));
    PF_HIP(hipEventSynchronize(e1));
    float ms = 0;
    PF_HIP(hipEventElapsedTime(&ms, e0, e1));
    *ms_host = (double)ms / nrep;
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return rc;
}
