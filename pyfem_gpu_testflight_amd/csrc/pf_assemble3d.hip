// NUMERIC phase for the 3-D 8-node block element: fused quadrature + scatter.
//
// Reference path (examples/block/block.py, examples/heat_conduction.py, SciTech2023/performance_test.py):
//   QuadratureBlock3D (pyfem.py:97-112, 2x2x2 Gauss, x slowest / z fastest), BasisBlock3D (pyfem.py:287-338),
//   utils.compute_jtrans / compute_jdet / compute_basis_grad 3x3 branch (utils.py:172-264),
//   LinearElasticity 3-D C0 (pyfem.py:1752-1757) + Bmat (pyfem.py:1999-2011) + einsum Ke (pyfem.py:2016-2027),
//   LinearPoisson Ke (pyfem.py:1176-1187), Helmholtz Ke/Re (pyfem.py:2127-2177), coo->csr (pyfem.py:921-931).
//
// Same design as the 2-D kernel (pf_assemble.hip): lane = incident (node, element) pair, tile = up to 8
// consecutive nodes (<= 64 pairs), LDS image of the tile's contiguous value segment, LDS fp64 adds, one
// straight copy-out; every K value is written to HBM once, Ke (24x24 per element, 4.6 KB) never exists.
// The lane integrates the d rows of its node: 3 x 24 values for elasticity.  With the isotropic C0
//   K_ab[i][j] = sum_q w detJ C_q ( lam g_a[i] g_b[j] + mu g_a[j] g_b[i] + mu delta_ij g_a.g_b )
// so the lane accumulates S_b[i][j] = sum_q (w detJ C_q g_a[i]) g_b[j] (9 FMA per node pair and point) and
// applies lam/mu once at the end.  Unlike 2-D this kernel is VALU-bound (about 2.2 k fp64 FMA per pair against
// 1.9 KB written per node), so it is not software-pipelined; occupancy (2 waves/SIMD at ~250 VGPRs) hides
// the loads.
#include "pf_common.h"

namespace {

constexpr double G3 = 0.57735026918962576451;  // 1/sqrt(3)

enum { HEX_ELASTICITY = 0, HEX_POISSON = 1, HEX_HELMHOLTZ = 2 };
template <int KIND> struct HexV { static constexpr int V = KIND == HEX_ELASTICITY ? 9 : KIND == HEX_HELMHOLTZ ? 2 : 1; };

struct HexParams {
  double lam, mu;     // elasticity: Lame constants of C0 (pyfem.py:1752-1757)
  double p;           // RAMP parameter (pyfem.py:1923-1945 / 1279-1301)
  double r02;         // Helmholtz r0^2
  const double* rho;  // nodal density or NULL
};

// node signs in BasisBlock3D order (pyfem.py:296-307): CCW on z=-1, then z=+1
__host__ __device__ constexpr int hsx(int l) { return ((l ^ (l >> 1)) & 1) ? 1 : -1; }
__host__ __device__ constexpr int hsy(int l) { return (l & 2) ? 1 : -1; }
__host__ __device__ constexpr int hsz(int l) { return (l & 4) ? 1 : -1; }

// RAMP factor rho_q / (1 + p (1 - rho_q)) at quadrature point q (utils.py:204-221, pyfem.py:1923-1945)
__device__ __forceinline__ double hex_ramp(double p, const double (&rho)[8], int q) {
  const double xi = (q & 4) ? G3 : -G3, et = (q & 2) ? G3 : -G3, ze = (q & 1) ? G3 : -G3;
  double rq = 0.0;
#pragma unroll
  for (int l = 0; l < 8; ++l) rq += 0.125 * (1.0 + hsx(l) * xi) * (1.0 + hsy(l) * et) * (1.0 + hsz(l) * ze) * rho[l];
  return rq / (1.0 + p * (1.0 - rq));
}

// rows of node `a` (runtime local index) of the element matrix: v[b][*] = block (a, b).
// cq[q * cq_stride]: RAMP factor of point q (RHO only) -- parked in LDS by the caller, because the eight
// density values would cost the 16 registers that decide between one and two waves per SIMD.
template <int KIND, bool RHO>
__device__ __forceinline__ void hex_rows(const HexParams& pr, const double (&cx)[8], const double (&cy)[8],
                                         const double (&cz)[8], const double* cq, int cq_stride, int a,
                                         double (&v)[8][HexV<KIND>::V]) {
  constexpr int NS = KIND == HEX_ELASTICITY ? 9 : KIND == HEX_HELMHOLTZ ? 2 : 1;
  const double sxa = ((a ^ (a >> 1)) & 1) ? 1.0 : -1.0, sya = (a & 2) ? 1.0 : -1.0, sza = (a & 4) ? 1.0 : -1.0;
  double S[8][NS];
#pragma unroll
  for (int b = 0; b < 8; ++b)
#pragma unroll
    for (int k = 0; k < NS; ++k) S[b][k] = 0.0;

  // The trilinear map in monomial form: x(xi) = c0 + c[0] xi + c[1] eta + c[2] zeta + c[3] xi eta + c[4] eta zeta
  // + c[5] xi zeta + c[6] xi eta zeta.  A Jacobian column is then 3 FMAs per component instead of the 8-term
  // sum of utils.py:172-185 (same polynomial, different rounding), and only 21 coordinate values stay live.
  double mc[3][7];
  {
    const double (*cc[3])[8] = {&cx, &cy, &cz};
#pragma unroll
    for (int d = 0; d < 3; ++d) {
#pragma unroll
      for (int t = 0; t < 7; ++t) mc[d][t] = 0.0;
#pragma unroll
      for (int l = 0; l < 8; ++l) {
        const double xv = 0.125 * (*cc[d])[l];
        mc[d][0] += hsx(l) * xv;
        mc[d][1] += hsy(l) * xv;
        mc[d][2] += hsz(l) * xv;
        mc[d][3] += hsx(l) * hsy(l) * xv;
        mc[d][4] += hsy(l) * hsz(l) * xv;
        mc[d][5] += hsx(l) * hsz(l) * xv;
        mc[d][6] += hsx(l) * hsy(l) * hsz(l) * xv;
      }
    }
  }

#pragma unroll 1
  for (int q = 0; q < 8; ++q) {  // pyfem.py:97-112: q = 4 ix + 2 iy + iz
    const double xi = (q & 4) ? G3 : -G3, et = (q & 2) ? G3 : -G3, ze = (q & 1) ? G3 : -G3;
    const double fx[2] = {1.0 - xi, 1.0 + xi}, fy[2] = {1.0 - et, 1.0 + et}, fz[2] = {1.0 - ze, 1.0 + ze};
    double pyz[2][2], pxz[2][2], pxy[2][2];  // 0.125 x pairs of factors
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        pyz[i][j] = 0.125 * fy[i] * fz[j];
        pxz[i][j] = 0.125 * fx[i] * fz[j];
        pxy[i][j] = 0.125 * fx[i] * fy[j];
      }
    // Jq[j][k] = d x_j / d xi_k
    const double xe = xi * et, ez = et * ze, xz = xi * ze;
    double J[3][3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      J[j][0] = mc[j][0] + mc[j][3] * et + mc[j][5] * ze + mc[j][6] * ez;
      J[j][1] = mc[j][1] + mc[j][3] * xi + mc[j][4] * ze + mc[j][6] * xz;
      J[j][2] = mc[j][2] + mc[j][5] * xi + mc[j][4] * et + mc[j][6] * xe;
    }
    const double c00 = J[1][1] * J[2][2] - J[1][2] * J[2][1];
    const double c01 = J[1][0] * J[2][2] - J[1][2] * J[2][0];
    const double c02 = J[1][0] * J[2][1] - J[1][1] * J[2][0];
    const double det = J[0][0] * c00 - J[0][1] * c01 + J[0][2] * c02;  // utils.py:189-200
    const double rd = 1.0 / det;
    double inv[3][3];  // utils.py:250-260
    inv[0][0] = c00 * rd;
    inv[0][1] = -(J[0][1] * J[2][2] - J[0][2] * J[2][1]) * rd;
    inv[0][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) * rd;
    inv[1][0] = -c01 * rd;
    inv[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) * rd;
    inv[1][2] = -(J[0][0] * J[1][2] - J[0][2] * J[1][0]) * rd;
    inv[2][0] = c02 * rd;
    inv[2][1] = -(J[0][0] * J[2][1] - J[0][1] * J[2][0]) * rd;
    inv[2][2] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) * rd;
    double scale = det;  // weights are 1
    if constexpr (RHO) scale *= cq[q * cq_stride];
    // own node: factors from its runtime signs (no register array is indexed by a)
    const double fxa = 1.0 + sxa * xi, fya = 1.0 + sya * et, fza = 1.0 + sza * ze;
    const double a0 = 0.125 * sxa * fya * fza, a1 = 0.125 * sya * fxa * fza, a2 = 0.125 * sza * fxa * fya;
    double ga[3];  // scale * grad N_a
#pragma unroll
    for (int j = 0; j < 3; ++j) ga[j] = scale * (a0 * inv[0][j] + a1 * inv[1][j] + a2 * inv[2][j]);
    const double Na = 0.125 * fxa * fya * fza;
    // elasticity: the gradients of the eight shape functions sum to zero, so the block of node 7 is minus the
    // sum of the others -- 63 accumulators instead of 72 (the kernel is register-bound: 2 waves per SIMD)
    constexpr int NB = KIND == HEX_ELASTICITY ? 7 : 8;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int ix = hsx(b) > 0, iy = hsy(b) > 0, iz = hsz(b) > 0;
      const double d0 = hsx(b) * pyz[iy][iz], d1 = hsy(b) * pxz[ix][iz], d2 = hsz(b) * pxy[ix][iy];
      double gb[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) gb[j] = d0 * inv[0][j] + d1 * inv[1][j] + d2 * inv[2][j];  // utils.py:263
      if constexpr (KIND == HEX_ELASTICITY) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int j = 0; j < 3; ++j) S[b][3 * i + j] += ga[i] * gb[j];
      } else {
        S[b][0] += ga[0] * gb[0] + ga[1] * gb[1] + ga[2] * gb[2];
        if constexpr (KIND == HEX_HELMHOLTZ) S[b][1] += det * Na * (pxy[ix][iy] * fz[iz]);  // pyfem.py:2134
      }
    }
  }
  if constexpr (KIND == HEX_ELASTICITY) {
#pragma unroll
    for (int k = 0; k < 9; ++k) S[7][k] = -(((S[0][k] + S[1][k]) + (S[2][k] + S[3][k])) + ((S[4][k] + S[5][k]) + S[6][k]));
  }
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    if constexpr (KIND == HEX_ELASTICITY) {
      const double tr = S[b][0] + S[b][4] + S[b][8];
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
          v[b][3 * i + j] = pr.lam * S[b][3 * i + j] + pr.mu * S[b][3 * j + i] + (i == j ? pr.mu * tr : 0.0);
    } else if constexpr (KIND == HEX_HELMHOLTZ) {
      v[b][0] = pr.r02 * S[b][0] + S[b][1];  // pyfem.py:2127-2131, 2176
      v[b][1] = S[b][1];
    } else {
      v[b][0] = S[b][0];
    }
  }
}

__device__ __forceinline__ void hex_load(const double* __restrict__ X, const double* __restrict__ rho, const int4& r0,
                                         const int4& r1, double (&cx)[8], double (&cy)[8], double (&cz)[8],
                                         double (&re)[8]) {
  const int32_t nd[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
#pragma unroll
  for (int l = 0; l < 8; ++l) {
    const double* xp = X + 3 * (int64_t)nd[l];
    cx[l] = xp[0];
    cy[l] = xp[1];
    cz[l] = xp[2];
    re[l] = rho ? rho[nd[l]] : 1.0;
  }
}

// dynamic LDS: double img[V * plane]
template <int KIND, bool RHO>
__global__ __launch_bounds__(64, 2) void k_assemble_hex(int64_t ntiles, int32_t plane, const double* __restrict__ X,
                                                     const int32_t* __restrict__ nptr,
                                                     const int4* __restrict__ tile_desc,
                                                     const int4* __restrict__ vis_rec, HexParams pr,
                                                     double* __restrict__ out0, double* __restrict__ out1) {
  constexpr int V = HexV<KIND>::V;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem3[];
  double* const img = reinterpret_cast<double*>(smem3);
  const int lane = threadIdx.x;
  for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int4 dsc = tile_desc[t];
    const int32_t n0 = dsc.x, vb = dsc.y, base = dsc.z;
    const int32_t nv = (dsc.w >> 8) & 255;
    const int32_t nblk = (int32_t)((uint32_t)dsc.w >> 16);
    const bool active = lane < nv;
    int4 r2 = make_int4(0, 0, 0, 0);
    int32_t roff = 0;
    double v[8][V];
    if (active) {
      const int64_t jj = (int64_t)vb + lane;
      const int4 r0 = vis_rec[3 * jj], r1 = vis_rec[3 * jj + 1];
      r2 = vis_rec[3 * jj + 2];
      const int a = r2.z & 255, slot = (r2.z >> 8) & 255;
      roff = nptr[n0 + slot] - base;
      double cx[8], cy[8], cz[8], re[8];
      hex_load(X, RHO ? pr.rho : nullptr, r0, r1, cx, cy, cz, re);
      if constexpr (RHO) {  // the image is free until the adds below: park the eight RAMP factors there
#pragma unroll
        for (int q = 0; q < 8; ++q) img[q * 64 + lane] = hex_ramp(pr.p, re, q);
      }
      hex_rows<KIND, RHO>(pr, cx, cy, cz, img + lane, 64, a, v);
    }
    // one wave: LDS operations execute in program order, so no barrier between the phases
#pragma unroll
    for (int k = 0; k < V; ++k)
      for (int32_t q = lane; q < nblk; q += 64) img[k * plane + q] = 0.0;
    if (active) {
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        const uint32_t pw = (uint32_t)(b < 4 ? r2.x : r2.y);
        const int blk = roff + (int)((pw >> (8 * (b & 3))) & 255);
#pragma unroll
        for (int k = 0; k < V; ++k)
          __hip_atomic_fetch_add(img + k * plane + blk, v[b][k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      }
    }
    // copy-out
    if constexpr (V == 9) {
      double* dst = out0 + (int64_t)base * 9;
      const int32_t tot = nblk * 9;
      for (int32_t e = lane; e < tot; e += 64) {
        const int32_t blk = e / 9, k = e - 9 * blk;
        dst[e] = img[k * plane + blk];
      }
    } else {
      for (int32_t q = lane; q < nblk; q += 64) {
        out0[base + q] = img[q];
        if constexpr (V == 2) out1[base + q] = img[plane + q];
      }
    }
  }
}

// Ke (E, 8*D, 8*D) row-major for the parity tests (model.Ke_mat of the reference); thread = (element, a)
template <int KIND, bool RHO>
__global__ __launch_bounds__(128) void k_element_matrices_hex(int64_t nelems, const double* __restrict__ X, const int32_t* __restrict__ conn,
                                       HexParams pr, double* __restrict__ Ke, double* __restrict__ Re) {
  constexpr int V = HexV<KIND>::V;
  constexpr int D = KIND == HEX_ELASTICITY ? 3 : 1;
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= nelems * 8) return;
  const int64_t e = i >> 3;
  const int a = (int)(i & 7);
  const int32_t* c = conn + e * 8;
  const int4 r0 = make_int4(c[0], c[1], c[2], c[3]), r1 = make_int4(c[4], c[5], c[6], c[7]);
  double cx[8], cy[8], cz[8], re[8];
  hex_load(X, RHO ? pr.rho : nullptr, r0, r1, cx, cy, cz, re);
  double cq[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) cq[q] = RHO ? hex_ramp(pr.p, re, q) : 1.0;
  double v[8][V];
  hex_rows<KIND, RHO>(pr, cx, cy, cz, cq, 1, a, v);
  constexpr int ED = 8 * D;
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    if constexpr (KIND == HEX_ELASTICITY) {
#pragma unroll
      for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int l = 0; l < 3; ++l) Ke[(e * ED + 3 * a + k) * ED + 3 * b + l] = v[b][3 * k + l];
    } else {
      Ke[(e * ED + a) * ED + b] = v[b][0];
      if constexpr (KIND == HEX_HELMHOLTZ)
        if (Re) Re[(e * ED + a) * ED + b] = v[b][1];
    }
  }
}

// monomial coefficients of a nodal field on the trilinear element (0.125 folded in):
//   f(xi) = m[7] + m[0] xi + m[1] eta + m[2] zeta + m[3] xi eta + m[4] eta zeta + m[5] xi zeta + m[6] xi eta zeta
__device__ __forceinline__ void hex_monomials(const double (&f)[8], double (&m)[8]) {
#pragma unroll
  for (int t = 0; t < 8; ++t) m[t] = 0.0;
#pragma unroll
  for (int l = 0; l < 8; ++l) {
    const double v = 0.125 * f[l];
    m[0] += hsx(l) * v;
    m[1] += hsy(l) * v;
    m[2] += hsz(l) * v;
    m[3] += hsx(l) * hsy(l) * v;
    m[4] += hsy(l) * hsz(l) * v;
    m[5] += hsx(l) * hsz(l) * v;
    m[6] += hsx(l) * hsy(l) * hsz(l) * v;
    m[7] += v;
  }
}
// d f / d xi_k at (xi, eta, zeta)
__device__ __forceinline__ void hex_dlocal(const double (&m)[8], double xi, double et, double ze, double (&d)[3]) {
  d[0] = m[0] + m[3] * et + m[5] * ze + m[6] * (et * ze);
  d[1] = m[1] + m[3] * xi + m[4] * ze + m[6] * (xi * ze);
  d[2] = m[2] + m[5] * xi + m[4] * et + m[6] * (xi * et);
}
__device__ __forceinline__ double hex_value(const double (&m)[8], double xi, double et, double ze) {
  return m[7] + m[0] * xi + m[1] * et + m[2] * ze + m[3] * (xi * et) + m[4] * (et * ze) + m[5] * (xi * ze) +
         m[6] * (xi * et * ze);
}
// J[j][k] = d x_j / d xi_k, its determinant and inverse (utils.py:172-264)
__device__ __forceinline__ double hex_jacobian(const double (&mx)[8], const double (&my)[8], const double (&mz)[8],
                                               double xi, double et, double ze, double (&inv)[3][3]) {
  double J[3][3];
  hex_dlocal(mx, xi, et, ze, J[0]);
  hex_dlocal(my, xi, et, ze, J[1]);
  hex_dlocal(mz, xi, et, ze, J[2]);
  const double c00 = J[1][1] * J[2][2] - J[1][2] * J[2][1];
  const double c01 = J[1][0] * J[2][2] - J[1][2] * J[2][0];
  const double c02 = J[1][0] * J[2][1] - J[1][1] * J[2][0];
  const double det = J[0][0] * c00 - J[0][1] * c01 + J[0][2] * c02;
  const double rd = 1.0 / det;
  inv[0][0] = c00 * rd;
  inv[0][1] = -(J[0][1] * J[2][2] - J[0][2] * J[2][1]) * rd;
  inv[0][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) * rd;
  inv[1][0] = -c01 * rd;
  inv[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) * rd;
  inv[1][2] = -(J[0][0] * J[1][2] - J[0][2] * J[1][0]) * rd;
  inv[2][0] = c02 * rd;
  inv[2][1] = -(J[0][0] * J[2][1] - J[0][1] * J[2][0]) * rd;
  inv[2][2] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) * rd;
  return det;
}

// ------------------------------------------------------------------------------------------
// Sensitivity pass on blocks: out[o] = d(phi^T K(rho) psi)/d rho_o  (pyfem.py:1873-1920 / 1240-1276 with the
// RAMP derivative of :1948-1973 / :1303-1329).  One lane per node o, its incident elements in ascending order:
//   inner[e,o] = sum_q detJ N_o(q) r'(rho_q) eps(phi):C0:eps(psi),  r' = (1+p)/(1+p(1-rho_q))^2.
// Strains come from the displacement gradient U = (d phi / d xi) J^-1, with d phi / d xi from the monomial form
// of the nodal field -- Ke_deriv (E,24,24,8) = 36 KB per element in the reference is never formed.
// ------------------------------------------------------------------------------------------
template <bool SCALAR>
__global__ __launch_bounds__(64) void k_sens_hex(int64_t nnodes, const double* __restrict__ X,
                                                 const int32_t* __restrict__ n2e_ptr, const int4* __restrict__ vis_rec,
                                                 HexParams pr, const double* __restrict__ phi,
                                                 const double* __restrict__ psi, double* __restrict__ out) {
  constexpr int NF = SCALAR ? 1 : 3;
  const int64_t n = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (n >= nnodes) return;
  double acc = 0.0;
  for (int32_t jj = n2e_ptr[n]; jj < n2e_ptr[n + 1]; ++jj) {
    const int4 r0 = vis_rec[3 * (int64_t)jj], r1 = vis_rec[3 * (int64_t)jj + 1], r2 = vis_rec[3 * (int64_t)jj + 2];
    const int a = r2.z & 255;
    const double sxa = ((a ^ (a >> 1)) & 1) ? 1.0 : -1.0, sya = (a & 2) ? 1.0 : -1.0, sza = (a & 4) ? 1.0 : -1.0;
    const int32_t nd[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
    double mx[8], my[8], mz[8], mr[8], mf[NF][8], mg[NF][8];
    {
      double t0[8], t1[8], t2[8], t3[8];
#pragma unroll
      for (int l = 0; l < 8; ++l) {
        const double* xp = X + 3 * (int64_t)nd[l];
        t0[l] = xp[0]; t1[l] = xp[1]; t2[l] = xp[2]; t3[l] = pr.rho[nd[l]];
      }
      hex_monomials(t0, mx); hex_monomials(t1, my); hex_monomials(t2, mz); hex_monomials(t3, mr);
#pragma unroll
      for (int c = 0; c < NF; ++c) {
#pragma unroll
        for (int l = 0; l < 8; ++l) { t0[l] = phi[NF * (int64_t)nd[l] + c]; t1[l] = psi[NF * (int64_t)nd[l] + c]; }
        hex_monomials(t0, mf[c]); hex_monomials(t1, mg[c]);
      }
    }
#pragma unroll 1
    for (int q = 0; q < 8; ++q) {
      const double xi = (q & 4) ? G3 : -G3, et = (q & 2) ? G3 : -G3, ze = (q & 1) ? G3 : -G3;
      double inv[3][3];
      const double det = hex_jacobian(mx, my, mz, xi, et, ze, inv);
      const double rq = hex_value(mr, xi, et, ze);
      const double t = 1.0 + pr.p * (1.0 - rq);
      const double dr = (1.0 + pr.p) / (t * t);
      const double Na = 0.125 * (1.0 + sxa * xi) * (1.0 + sya * et) * (1.0 + sza * ze);
      double U[NF][3], W[NF][3];  // U[i][j] = d phi_i / d x_j
#pragma unroll
      for (int c = 0; c < NF; ++c) {
        double df[3], dg[3];
        hex_dlocal(mf[c], xi, et, ze, df);
        hex_dlocal(mg[c], xi, et, ze, dg);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          U[c][j] = df[0] * inv[0][j] + df[1] * inv[1][j] + df[2] * inv[2][j];
          W[c][j] = dg[0] * inv[0][j] + dg[1] * inv[1][j] + dg[2] * inv[2][j];
        }
      }
      double eCe;
      if constexpr (SCALAR) {
        eCe = U[0][0] * W[0][0] + U[0][1] * W[0][1] + U[0][2] * W[0][2];
      } else {  // strain order xx,yy,zz,xy,yz,xz with engineering shear (pyfem.py:1999-2011)
        const double tu = U[0][0] + U[1][1] + U[2][2], tw = W[0][0] + W[1][1] + W[2][2];
        eCe = pr.lam * tu * tw + 2.0 * pr.mu * (U[0][0] * W[0][0] + U[1][1] * W[1][1] + U[2][2] * W[2][2]) +
              pr.mu * ((U[0][1] + U[1][0]) * (W[0][1] + W[1][0]) + (U[1][2] + U[2][1]) * (W[1][2] + W[2][1]) +
                       (U[0][2] + U[2][0]) * (W[0][2] + W[2][0]));
      }
      acc += det * dr * Na * eCe;
    }
  }
  out[n] = acc;
}

// Xq[e,q,:] = sum_l N[q,l] X[conn[e,l],:]: where LinearPoisson evaluates its Python source term (pyfem.py:1138-1173)
__global__ void k_quad_points_hex(int64_t nelems, const double* __restrict__ X, const int32_t* __restrict__ conn,
                                  double* __restrict__ Xq) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= nelems * 8) return;
  const int64_t e = i >> 3;
  const int q = (int)(i & 7);
  const double xi = (q & 4) ? G3 : -G3, et = (q & 2) ? G3 : -G3, ze = (q & 1) ? G3 : -G3;
  double s[3] = {0.0, 0.0, 0.0};
#pragma unroll
  for (int l = 0; l < 8; ++l) {
    const double N = 0.125 * (1.0 + hsx(l) * xi) * (1.0 + hsy(l) * et) * (1.0 + hsz(l) * ze);
    const double* xp = X + 3 * (int64_t)conn[e * 8 + l];
    s[0] += N * xp[0]; s[1] += N * xp[1]; s[2] += N * xp[2];
  }
  Xq[3 * i + 0] = s[0]; Xq[3 * i + 1] = s[1]; Xq[3 * i + 2] = s[2];
}

// LinearPoisson load vector WITH the reference's indexing (pyfem.py:1131-1136, 861-875):
//   rhs_e[e,j] = sum_k detJ[e,k] w_k N[j,k] g[e,k]   where N[j,k] = shape function of node k at quadrature point j
// (the table is indexed [point, node] but contracted as if it were [node, point]; harmless on 2-D quads where the
// table is symmetric, not here), scattered for all 8 columns since nquads == 8.  Reproduced as is; lane = node.
__global__ __launch_bounds__(64) void k_poisson_rhs_hex(int64_t nnodes, const double* __restrict__ X,
                                                        const int32_t* __restrict__ n2e_ptr,
                                                        const int4* __restrict__ vis_rec, const int32_t* __restrict__ n2e,
                                                        const double* __restrict__ gq, double* __restrict__ rhs) {
  const int64_t n = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (n >= nnodes) return;
  double acc = 0.0;
  for (int32_t jj = n2e_ptr[n]; jj < n2e_ptr[n + 1]; ++jj) {
    const int4 r0 = vis_rec[3 * (int64_t)jj], r1 = vis_rec[3 * (int64_t)jj + 1], r2 = vis_rec[3 * (int64_t)jj + 2];
    const int a = r2.z & 255;
    const int64_t e = n2e[jj] >> 3;
    const int32_t nd[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
    double mx[8], my[8], mz[8];
    {
      double t0[8], t1[8], t2[8];
#pragma unroll
      for (int l = 0; l < 8; ++l) {
        const double* xp = X + 3 * (int64_t)nd[l];
        t0[l] = xp[0]; t1[l] = xp[1]; t2[l] = xp[2];
      }
      hex_monomials(t0, mx); hex_monomials(t1, my); hex_monomials(t2, mz);
    }
    // "quadrature point a" of the transposed table
    const double xa = (a & 4) ? G3 : -G3, ea = (a & 2) ? G3 : -G3, za = (a & 1) ? G3 : -G3;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const double xi = (k & 4) ? G3 : -G3, et = (k & 2) ? G3 : -G3, ze = (k & 1) ? G3 : -G3;
      double inv[3][3];
      const double det = hex_jacobian(mx, my, mz, xi, et, ze, inv);
      const double Nk_at_a = 0.125 * (1.0 + hsx(k) * xa) * (1.0 + hsy(k) * ea) * (1.0 + hsz(k) * za);
      acc += det * Nk_at_a * gq[8 * e + k];
    }
  }
  rhs[n] = acc;
}

template <int KIND, bool RHO>
int launch_hex(pf_ctx* ctx, const HexParams& pr, double* out0, double* out1) {
  constexpr int V = HexV<KIND>::V;
  const int32_t plane = ctx->npt * ctx->max_row_blocks + 1;
  const size_t lds = ((size_t)std::max<int64_t>((int64_t)plane * V, RHO ? 512 : 0) * 8 + 15) & ~(size_t)15;  // RHO: 8 x 64 RAMP factors
  auto k = k_assemble_hex<KIND, RHO>;
  if (lds > 48 * 1024)
    PF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  int wpc = 8;  // persistent waves per CU: ~250 VGPRs -> 2 per SIMD
  if (const char* e = pf_exp_env("PF_HEX_WAVES_PER_CU")) wpc = std::max(1, atoi(e));
  const unsigned grid = (unsigned)std::min<int64_t>(ctx->ntiles, (int64_t)256 * wpc);
  hipLaunchKernelGGL(k, dim3(grid), dim3(64), lds, ctx->stream, ctx->ntiles, plane, ctx->X, ctx->nptr,
                     reinterpret_cast<const int4*>(ctx->tile_desc), reinterpret_cast<const int4*>(ctx->vis_rec), pr,
                     out0, out1);
  PF_LAUNCH_CHECK();
  return PF_OK;
}

int check3d(pf_ctx* ctx, int d, const char* who) {
  if (!ctx) return pf_fail(nullptr, PF_E_INVALID, "ctx is NULL");
  if (ctx->ndims != 3 || ctx->P != 8) return pf_fail(ctx, PF_E_INVALID, std::string(who) + ": needs a pf_mesh_set3d mesh");
  if (d && ctx->d != d)
    return pf_fail(ctx, PF_E_INVALID, std::string(who) + ": pattern was not built with ndof_per_node = " + std::to_string(d));
  if (d && (!ctx->npt || !ctx->tile_desc)) return pf_fail(ctx, PF_E_UNSUPPORTED, std::string(who) + ": node valence above 64");
  return PF_OK;
}

HexParams lame(double E, double nu, double p, const double* rho) {
  HexParams pr{};
  pr.lam = E * nu / ((1.0 + nu) * (1.0 - 2.0 * nu));
  pr.mu = E / (2.0 * (1.0 + nu));
  pr.p = p;
  pr.rho = rho;
  return pr;
}

}  // namespace

extern "C" int pf_assemble_elasticity3d(pf_ctx* ctx, double E, double nu, double p, const double* rho, double* Kvals) {
  if (int rc = check3d(ctx, 3, "pf_assemble_elasticity3d")) return rc;
  PF_REQUIRE(Kvals, "pf_assemble_elasticity3d: Kvals is NULL");
  const HexParams pr = lame(E, nu, p, rho);
  return rho ? launch_hex<HEX_ELASTICITY, true>(ctx, pr, Kvals, nullptr)
             : launch_hex<HEX_ELASTICITY, false>(ctx, pr, Kvals, nullptr);
}

extern "C" int pf_assemble_poisson3d(pf_ctx* ctx, double p, const double* rho, double* Kvals) {
  if (int rc = check3d(ctx, 1, "pf_assemble_poisson3d")) return rc;
  PF_REQUIRE(Kvals, "pf_assemble_poisson3d: Kvals is NULL");
  HexParams pr{};
  pr.p = p;
  pr.rho = rho;
  return rho ? launch_hex<HEX_POISSON, true>(ctx, pr, Kvals, nullptr) : launch_hex<HEX_POISSON, false>(ctx, pr, Kvals, nullptr);
}

extern "C" int pf_assemble_helmholtz3d(pf_ctx* ctx, double r0, double* Kvals, double* Rvals) {
  if (int rc = check3d(ctx, 1, "pf_assemble_helmholtz3d")) return rc;
  PF_REQUIRE(Kvals && Rvals, "pf_assemble_helmholtz3d: NULL output");
  HexParams pr{};
  pr.r02 = r0 * r0;
  return launch_hex<HEX_HELMHOLTZ, false>(ctx, pr, Kvals, Rvals);
}

extern "C" int pf_element_matrices3d(pf_ctx* ctx, int kind, double E, double nu, double p, double r0, const double* rho,
                                     double* Ke_out, double* Re_out) {
  if (int rc = check3d(ctx, 0, "pf_element_matrices3d")) return rc;
  PF_REQUIRE(Ke_out, "pf_element_matrices3d: Ke_out is NULL");
  const unsigned grid = (unsigned)cdiv(ctx->nelems * 8, 128);
  HexParams pr = lame(E, nu, p, rho);
  pr.r02 = r0 * r0;
  if (kind == HEX_ELASTICITY) {
    if (rho) k_element_matrices_hex<HEX_ELASTICITY, true><<<grid, 128, 0, ctx->stream>>>(ctx->nelems, ctx->X, ctx->conn, pr, Ke_out, nullptr);
    else k_element_matrices_hex<HEX_ELASTICITY, false><<<grid, 128, 0, ctx->stream>>>(ctx->nelems, ctx->X, ctx->conn, pr, Ke_out, nullptr);
  } else if (kind == HEX_POISSON) {
    if (rho) k_element_matrices_hex<HEX_POISSON, true><<<grid, 128, 0, ctx->stream>>>(ctx->nelems, ctx->X, ctx->conn, pr, Ke_out, nullptr);
    else k_element_matrices_hex<HEX_POISSON, false><<<grid, 128, 0, ctx->stream>>>(ctx->nelems, ctx->X, ctx->conn, pr, Ke_out, nullptr);
  } else if (kind == HEX_HELMHOLTZ) {
    k_element_matrices_hex<HEX_HELMHOLTZ, false><<<grid, 128, 0, ctx->stream>>>(ctx->nelems, ctx->X, ctx->conn, pr, Ke_out, Re_out);
  } else {
    return pf_fail(ctx, PF_E_INVALID, "pf_element_matrices3d: kind must be 0 (elasticity), 1 (Poisson) or 2 (Helmholtz)");
  }
  PF_LAUNCH_CHECK();
  return PF_OK;
}

extern "C" int pf_sens3d(pf_ctx* ctx, int scalar, double E, double nu, double p, const double* rho, const double* phi,
                         const double* psi, double* out) {
  if (int rc = check3d(ctx, scalar ? 1 : 3, "pf_sens3d")) return rc;
  PF_REQUIRE(rho && phi && psi && out, "pf_sens3d: NULL argument");
  const HexParams pr = lame(E, nu, p, rho);
  const unsigned grid = (unsigned)cdiv(ctx->nnodes, 64);
  if (scalar) k_sens_hex<true><<<grid, 64, 0, ctx->stream>>>(ctx->nnodes, ctx->X, ctx->n2e_ptr, reinterpret_cast<const int4*>(ctx->vis_rec), pr, phi, psi, out);
  else k_sens_hex<false><<<grid, 64, 0, ctx->stream>>>(ctx->nnodes, ctx->X, ctx->n2e_ptr, reinterpret_cast<const int4*>(ctx->vis_rec), pr, phi, psi, out);
  PF_LAUNCH_CHECK();
  return PF_OK;
}

extern "C" int pf_quad_points3d(pf_ctx* ctx, double* Xq_out) {
  if (int rc = check3d(ctx, 0, "pf_quad_points3d")) return rc;
  PF_REQUIRE(Xq_out, "pf_quad_points3d: NULL output");
  k_quad_points_hex<<<(unsigned)cdiv(ctx->nelems * 8, 256), 256, 0, ctx->stream>>>(ctx->nelems, ctx->X, ctx->conn, Xq_out);
  PF_LAUNCH_CHECK();
  return PF_OK;
}

extern "C" int pf_poisson_rhs3d(pf_ctx* ctx, const double* gq, double* rhs_out) {
  if (int rc = check3d(ctx, 1, "pf_poisson_rhs3d")) return rc;
  PF_REQUIRE(gq && rhs_out, "pf_poisson_rhs3d: NULL argument");
  k_poisson_rhs_hex<<<(unsigned)cdiv(ctx->nnodes, 64), 64, 0, ctx->stream>>>(ctx->nnodes, ctx->X, ctx->n2e_ptr, reinterpret_cast<const int4*>(ctx->vis_rec), ctx->n2e, gq, rhs_out);
  PF_LAUNCH_CHECK();
  return PF_OK;
}
