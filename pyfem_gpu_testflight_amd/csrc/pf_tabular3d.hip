// Table-driven isoparametric 3-D elements (gfx950): every element shape the reference builds from a
// (Quadrature, Basis) pair other than the three hard-wired ones -- the 10-node tetrahedron with its 5-point rule and
// the 20-node brick with 3x3x3 Gauss points (pyfem.py:115-172, 380-632; tests/test_elements.py,
// examples/element_plot) -- goes through ONE kernel family that takes the basis as data:
//     N (Q,P) shape values, dN (Q,P,3) reference-space derivatives, w (Q) weights     (device arrays, fp64)
// exactly the arrays BasisBase.eval_shape_fun / eval_shape_fun_deriv and QuadratureBase.get_weight return
// (pyfem.py:222-250, 57-65).
//
// One workgroup per element.  Geometry once per quadrature point into LDS (J = dx/dxi, its inverse, the physical
// gradients G[q][a] = dN[q][a] J^-1, the scale w detJ RAMP(rho_q)); then the P x P node pairs are dealt to the
// threads, each summing its 3x3 block over the quadrature points
//     K_ab[i][j] = sum_q s_q ( lam G_a,i G_b,j + mu G_a,j G_b,i + mu delta_ij G_a.G_b )
// which is B_a^T C0 B_b for the isotropic C0 of pyfem.py:1752-1757 (the reference's einsum "iq,q,iqnj,iq,nm,iqmk->ijk",
// pyfem.py:2016-2027, contracts the same 6 x 3P strain matrices).  Blocks are added into the global matrix at the
// position found by bisection of the node row (rows hold up to 81 blocks; the byte-packed visit records of the block
// kernel stop at 64).  The scatter is COLOURED: the elements are sorted into classes none of whose members share a
// node (greedy, on the host, once per mesh), one launch per class, plain read-add-write -- no atomics, and the sum
// order is that of the classes, so results are bitwise reproducible.  (fp64 atomics, the first version, held the
// kernel at 20 G adds/s: 8.8 ms for 197 k tetrahedra.)
#include <algorithm>
#include <vector>

#include "pf_common.h"

namespace {

constexpr int TAB_THREADS = 256;

struct TabArgs {
  int P, Q;
  const double* N;    // (Q,P)
  const double* dN;   // (Q,P,3)
  const double* w;    // (Q)
  const double* X;    // (nnodes,3)
  const int32_t* conn;
  const double* rho;  // nodal density or nullptr
  double lam, mu, p;
};

// LDS layout (doubles): Xe[3P] | inv[9Q] | sc[Q] | rq[Q] | G[3PQ] | extra ; node ids as int32 after that
struct TabLds {
  double *Xe, *inv, *sc, *rq, *G, *extra;
  int32_t* en;
};

__device__ __forceinline__ TabLds tab_lds(double* sh, int P, int Q, int nextra) {
  TabLds L;
  L.Xe = sh;
  L.inv = L.Xe + 3 * P;
  L.sc = L.inv + 9 * Q;
  L.rq = L.sc + Q;
  L.G = L.rq + Q;
  L.extra = L.G + 3 * P * Q;
  L.en = reinterpret_cast<int32_t*>(L.extra + nextra);
  return L;
}

static size_t tab_lds_bytes(int P, int Q, int nextra) {
  return sizeof(double) * (size_t)(3 * P + 9 * Q + 2 * Q + 3 * P * Q + nextra) + sizeof(int32_t) * (size_t)P;
}

// geometry of element e into LDS; sc[q] = w detJ (no material factor), rq[q] = interpolated density (1 without rho)
__device__ __forceinline__ void tab_geometry(const TabArgs& A, int64_t e, const TabLds& L) {
  const int P = A.P, Q = A.Q, t = threadIdx.x;
  for (int a = t; a < P; a += TAB_THREADS) {
    const int32_t n = A.conn[e * P + a];
    L.en[a] = n;
    L.Xe[3 * a + 0] = A.X[3 * (int64_t)n + 0];
    L.Xe[3 * a + 1] = A.X[3 * (int64_t)n + 1];
    L.Xe[3 * a + 2] = A.X[3 * (int64_t)n + 2];
  }
  __syncthreads();
  for (int q = t; q < Q; q += TAB_THREADS) {
    double J[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};  // J[j][k] = dx_j / dxi_k   (utils.py:172-185)
    double r = A.rho ? 0.0 : 1.0;
    for (int a = 0; a < P; ++a) {
      const double* d = A.dN + ((int64_t)q * P + a) * 3;
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int k = 0; k < 3; ++k) J[j][k] += L.Xe[3 * a + j] * d[k];
      if (A.rho) r += A.N[(int64_t)q * P + a] * A.rho[L.en[a]];  // utils.py:204-221
    }
    const double c00 = J[1][1] * J[2][2] - J[1][2] * J[2][1];
    const double c01 = J[1][0] * J[2][2] - J[1][2] * J[2][0];
    const double c02 = J[1][0] * J[2][1] - J[1][1] * J[2][0];
    const double det = J[0][0] * c00 - J[0][1] * c01 + J[0][2] * c02;  // sign kept, like the reference
    const double id = 1.0 / det;
    double* iv = L.inv + 9 * q;  // inverse of J (utils.py:250-260)
    iv[0] = c00 * id;
    iv[1] = -(J[0][1] * J[2][2] - J[0][2] * J[2][1]) * id;
    iv[2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) * id;
    iv[3] = -c01 * id;
    iv[4] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) * id;
    iv[5] = -(J[0][0] * J[1][2] - J[0][2] * J[1][0]) * id;
    iv[6] = c02 * id;
    iv[7] = -(J[0][0] * J[2][1] - J[0][1] * J[2][0]) * id;
    iv[8] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) * id;
    L.sc[q] = A.w[q] * det;
    L.rq[q] = r;
  }
  __syncthreads();
  for (int qa = t; qa < Q * P; qa += TAB_THREADS) {  // G[q][a][l] = sum_m dN[q][a][m] inv[m][l]   (utils.py:263)
    const int q = qa / P;
    const double* d = A.dN + (int64_t)qa * 3;
    const double* iv = L.inv + 9 * q;
#pragma unroll
    for (int l = 0; l < 3; ++l) L.G[3 * qa + l] = d[0] * iv[l] + d[1] * iv[3 + l] + d[2] * iv[6 + l];
  }
  __syncthreads();
}

// MODE 0: scatter into the block-CSR values (atomics); MODE 1: element matrices Ke (nelems, 3P, 3P)
template <int MODE>
__global__ __launch_bounds__(TAB_THREADS) void k_tab_elasticity(TabArgs A, const int32_t* __restrict__ order, int64_t count,
                                                              const int32_t* __restrict__ nptr,
                                                              const int32_t* __restrict__ nidx, double* __restrict__ out) {
  extern __shared__ double sh[];
  const int P = A.P, Q = A.Q, t = threadIdx.x;
  const TabLds L = tab_lds(sh, P, Q, 0);
  for (int64_t k = blockIdx.x; k < count; k += gridDim.x) {
    const int64_t e = order ? order[k] : k;  // MODE 0: the elements of one colour; MODE 1: all elements
    tab_geometry(A, e, L);
    for (int q = t; q < Q; q += TAB_THREADS) {  // RAMP: rho / (1 + p (1 - rho))   (pyfem.py:1923-1945)
      const double r = L.rq[q];
      L.sc[q] *= r / (1.0 + A.p * (1.0 - r));
    }
    __syncthreads();
    for (int pair = t; pair < P * P; pair += TAB_THREADS) {
      const int a = pair / P, b = pair - a * P;
      double S[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};  // sum_q s G_a (x) G_b
      for (int q = 0; q < Q; ++q) {
        const double s = L.sc[q];
        const double* ga = L.G + 3 * (q * P + a);
        const double* gb = L.G + 3 * (q * P + b);
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int j = 0; j < 3; ++j) S[i][j] += s * ga[i] * gb[j];
      }
      const double tr = S[0][0] + S[1][1] + S[2][2];
      double K[3][3];
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) K[i][j] = A.lam * S[i][j] + A.mu * S[j][i] + (i == j ? A.mu * tr : 0.0);
      if (MODE == 1) {
        double* dst = out + e * (int64_t)(9 * P * P);
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int j = 0; j < 3; ++j) dst[(int64_t)(3 * a + i) * (3 * P) + 3 * b + j] = K[i][j];
      } else {
        const int32_t na = L.en[a], nb = L.en[b];
        int32_t lo = nptr[na], hi = nptr[na + 1] - 1;
        while (lo < hi) {
          const int32_t mid = (lo + hi) >> 1;
          if (nidx[mid] < nb) lo = mid + 1; else hi = mid;
        }
        double* dst = out + 9 * (int64_t)lo;  // no other element of this colour touches node na: plain adds
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int j = 0; j < 3; ++j) dst[3 * i + j] += K[i][j];
      }
    }
    __syncthreads();  // LDS is reused by the next element
  }
}

// d(phi^T K psi)/d rho_o = sum_e sum_q w detJ RAMP'(rho_q) N[q,o] eps(phi):C0:eps(psi)    (pyfem.py:1873-1920, 1948-1973)
__global__ __launch_bounds__(TAB_THREADS) void k_tab_sens(TabArgs A, const int32_t* __restrict__ order, int64_t count,
                                                         const double* __restrict__ phi, const double* __restrict__ psi,
                                                         double* __restrict__ out) {
  extern __shared__ double sh[];
  const int P = A.P, Q = A.Q, t = threadIdx.x;
  const TabLds L = tab_lds(sh, P, Q, 6 * P + Q);
  double* fe = L.extra;           // phi of the element's nodes (3P)
  double* ge = L.extra + 3 * P;   // psi
  double* tq = L.extra + 6 * P;   // per quadrature point
  for (int64_t k = blockIdx.x; k < count; k += gridDim.x) {
    const int64_t e = order[k];
    tab_geometry(A, e, L);
    for (int k = t; k < 3 * P; k += TAB_THREADS) {
      const int64_t dof = 3 * (int64_t)L.en[k / 3] + k % 3;
      fe[k] = phi[dof];
      ge[k] = psi[dof];
    }
    __syncthreads();
    for (int q = t; q < Q; q += TAB_THREADS) {
      double F[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, H[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};  // grad phi, grad psi
      for (int a = 0; a < P; ++a) {
        const double* g = L.G + 3 * (q * P + a);
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int l = 0; l < 3; ++l) {
            F[i][l] += fe[3 * a + i] * g[l];
            H[i][l] += ge[3 * a + i] * g[l];
          }
      }
      double dd = 0.0, sym = 0.0;
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int l = 0; l < 3; ++l) {
          dd += F[i][l] * H[i][l];
          sym += F[i][l] * H[l][i];
        }
      const double form = A.lam * (F[0][0] + F[1][1] + F[2][2]) * (H[0][0] + H[1][1] + H[2][2]) + A.mu * (dd + sym);
      const double den = 1.0 + A.p * (1.0 - L.rq[q]);
      tq[q] = L.sc[q] * (1.0 + A.p) / (den * den) * form;
    }
    __syncthreads();
    for (int o = t; o < P; o += TAB_THREADS) {
      double s = 0.0;
      for (int q = 0; q < Q; ++q) s += tq[q] * A.N[(int64_t)q * P + o];
      out[L.en[o]] += s;  // coloured: this element is the only one of its class at node en[o]
    }
    __syncthreads();
  }
}

int check_tab(pf_ctx* ctx, int Q, const double* N, const double* dN, const double* wq, const char* who) {
  if (!ctx) return pf_fail(nullptr, PF_E_INVALID, "ctx is NULL");
  if (ctx->ndims != 3 || ctx->d != 3)
    return pf_fail(ctx, PF_E_INVALID, std::string(who) + ": needs a pf_mesh_set3d mesh and a pattern with 3 dofs per node");
  if (Q < 1 || Q > PF_TAB_MAX_QUADS || !N || !dN || !wq)
    return pf_fail(ctx, PF_E_INVALID, std::string(who) + ": 1 to 64 quadrature points and the three tables are required");
  return PF_OK;
}

TabArgs tab_args(pf_ctx* ctx, int Q, const double* N, const double* dN, const double* wq, double E, double nu, double p,
                 const double* rho) {
  TabArgs A{};
  A.P = ctx->P; A.Q = Q; A.N = N; A.dN = dN; A.w = wq; A.X = ctx->X; A.conn = ctx->conn; A.rho = rho;
  A.lam = E * nu / ((1.0 + nu) * (1.0 - 2.0 * nu));
  A.mu = E / (2.0 * (1.0 + nu));
  A.p = p;
  return A;
}

unsigned tab_grid(int64_t count) { return (unsigned)std::min<int64_t>(count, 256 * 8); }

// Greedy element colouring, once per mesh: an element takes the lowest colour none of its nodes has seen.  A node of
// the tetrahedral grid sits in up to 24 elements, so colours are tracked in 128-bit masks.
int ensure_colours(pf_ctx* ctx) {
  if (ctx->tab_order) return PF_OK;
  const int64_t ne = ctx->nelems;
  const int P = ctx->P;
  std::vector<int32_t> conn((size_t)ne * P);
  PF_HIP(hipMemcpyAsync(conn.data(), ctx->conn, sizeof(int32_t) * conn.size(), hipMemcpyDeviceToHost, ctx->stream));
  PF_HIP(hipStreamSynchronize(ctx->stream));
  constexpr int W = 2;  // 64-bit words per node
  std::vector<uint64_t> seen((size_t)ctx->nnodes * W, 0);
  std::vector<int32_t> colour(ne);
  int ncol = 0;
  for (int64_t e = 0; e < ne; ++e) {
    uint64_t m[W] = {0, 0};
    for (int a = 0; a < P; ++a)
      for (int w = 0; w < W; ++w) m[w] |= seen[(size_t)conn[e * P + a] * W + w];
    int c = -1;
    for (int w = 0; w < W && c < 0; ++w)
      if (~m[w]) c = 64 * w + __builtin_ctzll(~m[w]);
    if (c < 0) return pf_fail(ctx, PF_E_UNSUPPORTED, "table-driven assembly: more than 128 element colours");
    colour[e] = c;
    ncol = std::max(ncol, c + 1);
    for (int a = 0; a < P; ++a) seen[(size_t)conn[e * P + a] * W + (c >> 6)] |= (uint64_t)1 << (c & 63);
  }
  ctx->tab_color_ptr.assign(ncol + 1, 0);
  for (int64_t e = 0; e < ne; ++e) ctx->tab_color_ptr[colour[e] + 1]++;
  for (int c = 0; c < ncol; ++c) ctx->tab_color_ptr[c + 1] += ctx->tab_color_ptr[c];
  std::vector<int32_t> order(ne);
  std::vector<int64_t> cur(ctx->tab_color_ptr.begin(), ctx->tab_color_ptr.end() - 1);
  for (int64_t e = 0; e < ne; ++e) order[cur[colour[e]]++] = (int32_t)e;
  PF_HIP(hipMalloc(&ctx->tab_order, sizeof(int32_t) * (size_t)ne));
  PF_HIP(hipMemcpyAsync(ctx->tab_order, order.data(), sizeof(int32_t) * (size_t)ne, hipMemcpyHostToDevice, ctx->stream));
  PF_HIP(hipStreamSynchronize(ctx->stream));  // staging vector goes out of scope
  return PF_OK;
}

}  // namespace

extern "C" int pf_assemble_elasticity_tab3d(pf_ctx* ctx, int Q, const double* N, const double* dN, const double* wq,
                                            double E, double nu, double p, const double* rho, double* Kvals) {
  if (int rc = check_tab(ctx, Q, N, dN, wq, "pf_assemble_elasticity_tab3d")) return rc;
  PF_REQUIRE(Kvals, "pf_assemble_elasticity_tab3d: Kvals is NULL");
  PF_HIP(hipSetDevice(ctx->device));
  PF_HIP(hipMemsetAsync(Kvals, 0, sizeof(double) * 9 * (size_t)ctx->nblocks, ctx->stream));
  if (int rc = ensure_colours(ctx)) return rc;
  const TabArgs A = tab_args(ctx, Q, N, dN, wq, E, nu, p, rho);
  for (size_t c = 0; c + 1 < ctx->tab_color_ptr.size(); ++c) {
    const int64_t first = ctx->tab_color_ptr[c], count = ctx->tab_color_ptr[c + 1] - first;
    k_tab_elasticity<0><<<tab_grid(count), TAB_THREADS, tab_lds_bytes(A.P, Q, 0), ctx->stream>>>(
        A, ctx->tab_order + first, count, ctx->nptr, ctx->nidx, Kvals);
  }
  PF_LAUNCH_CHECK();
  return PF_OK;
}

extern "C" int pf_element_matrices_tab3d(pf_ctx* ctx, int Q, const double* N, const double* dN, const double* wq,
                                         double E, double nu, double p, const double* rho, double* Ke_out) {
  if (int rc = check_tab(ctx, Q, N, dN, wq, "pf_element_matrices_tab3d")) return rc;
  PF_REQUIRE(Ke_out, "pf_element_matrices_tab3d: Ke_out is NULL");
  PF_HIP(hipSetDevice(ctx->device));
  const TabArgs A = tab_args(ctx, Q, N, dN, wq, E, nu, p, rho);
  k_tab_elasticity<1><<<tab_grid(ctx->nelems), TAB_THREADS, tab_lds_bytes(A.P, Q, 0), ctx->stream>>>(
      A, nullptr, ctx->nelems, ctx->nptr, ctx->nidx, Ke_out);
  PF_LAUNCH_CHECK();
  return PF_OK;
}

extern "C" int pf_sens_tab3d(pf_ctx* ctx, int Q, const double* N, const double* dN, const double* wq, double E, double nu,
                             double p, const double* rho, const double* phi, const double* psi, double* out) {
  if (int rc = check_tab(ctx, Q, N, dN, wq, "pf_sens_tab3d")) return rc;
  PF_REQUIRE(rho && phi && psi && out, "pf_sens_tab3d: NULL argument");
  PF_HIP(hipSetDevice(ctx->device));
  PF_HIP(hipMemsetAsync(out, 0, sizeof(double) * (size_t)ctx->nnodes, ctx->stream));
  if (int rc = ensure_colours(ctx)) return rc;
  const TabArgs A = tab_args(ctx, Q, N, dN, wq, E, nu, p, rho);
  for (size_t c = 0; c + 1 < ctx->tab_color_ptr.size(); ++c) {
    const int64_t first = ctx->tab_color_ptr[c], count = ctx->tab_color_ptr[c + 1] - first;
    k_tab_sens<<<tab_grid(count), TAB_THREADS, tab_lds_bytes(A.P, Q, 6 * A.P + Q), ctx->stream>>>(
        A, ctx->tab_order + first, count, phi, psi, out);
  }
  PF_LAUNCH_CHECK();
  return PF_OK;
}
