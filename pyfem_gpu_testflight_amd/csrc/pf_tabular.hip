// Table-driven isoparametric elements (gfx950): ONE kernel family that takes the element as data --
//     N (Q,P) shape values, dN (Q,P,DIM) reference-space derivatives, w (Q) weights     (device arrays, fp64)
// exactly the arrays BasisBase.eval_shape_fun / eval_shape_fun_deriv and QuadratureBase.get_weight return
// (pyfem.py:222-250, 57-65).  It serves
//   * the element shapes the reference builds from a (Quadrature, Basis) pair other than the three hard-wired ones:
//     the 10-node tetrahedron with its 5-point rule and the 20-node brick with 3x3x3 Gauss points (pyfem.py:115-172,
//     380-632; tests/test_elements.py, examples/element_plot);
//   * the imaginary part of K(rho + i drho) -- the reference's derivative tests differentiate compute_jacobian by a
//     complex step (tests/test_elasticity.py:96-98, test_linear_poisson.py:82-83, test_elements.py:50-51); to first
//     order, which is all a step of 1e-30 keeps, Im K is K assembled with the coefficient RAMP'(rho_q) (N drho)_q;
//   * a second, independent implementation of every hard-wired kernel for the parity tests.
// 2-D and 3-D, elasticity (DIM dofs per node) or the scalar LinearPoisson operator.
//
// One workgroup per element.  Geometry once per quadrature point into LDS (J = dx/dxi, its inverse, the physical
// gradients G[q][a] = dN[q][a] J^-1, the scale w detJ coef(rho_q)); then the P x P node pairs are dealt to the
// threads, each summing S = sum_q s_q G_a (x) G_b over the quadrature points and forming
//     elasticity: K_ab[i][j] = lam S_ij + mu S_ji + mu delta_ij tr S        scalar: K_ab = tr S
// which is B_a^T C0 B_b for the isotropic C0 of pyfem.py:1746-1757 (3-D: lam, mu the Lame constants; plane stress:
// lam = E nu / (1 - nu^2); the reference's einsum "iq,q,iqnj,iq,nm,iqmk->ijk", pyfem.py:2016-2027, contracts the same
// strain matrices) and grad N_a . grad N_b of pyfem.py:1176-1187.  Blocks are added into the global matrix at the
// position found by bisection of the node row (rows hold up to 81 blocks; the byte-packed visit records of the block
// kernel stop at 64).  The scatter is COLOURED: the elements are sorted into classes none of whose members share a
// node (greedy, on the host, once per mesh), one launch per class, plain read-add-write -- no atomics, and the sum
// order is that of the classes, so results are bitwise reproducible.  (fp64 atomics, the first version, held the
// kernel at 20 G adds/s: 8.8 ms for 197 k tetrahedra against 2.6 ms.)
#include <algorithm>
#include <vector>

#include "pf_common.h"

namespace {

constexpr int TAB_THREADS = 256;

struct TabArgs {
  int P, Q;
  const double* N;     // (Q,P)
  const double* dN;    // (Q,P,DIM)
  const double* w;     // (Q)
  const double* X;     // (nnodes,DIM)
  const int32_t* conn;
  const double* rho;   // nodal density or nullptr
  const double* drho;  // nullptr: coefficient RAMP(rho_q); else RAMP'(rho_q) (N drho)_q
  double lam, mu, p;
};

// LDS layout (doubles): Xe[DIM P] | inv[DIM DIM Q] | sc[Q] | rq[Q] | G[DIM P Q] | extra ; node ids as int32 after that
struct TabLds {
  double *Xe, *inv, *sc, *rq, *G, *extra;
  int32_t* en;
};

template <int DIM>
__device__ __forceinline__ TabLds tab_lds(double* sh, int P, int Q, int nextra) {
  TabLds L;
  L.Xe = sh;
  L.inv = L.Xe + DIM * P;
  L.sc = L.inv + DIM * DIM * Q;
  L.rq = L.sc + Q;
  L.G = L.rq + Q;
  L.extra = L.G + DIM * P * Q;
  L.en = reinterpret_cast<int32_t*>(L.extra + nextra);
  return L;
}

size_t tab_lds_bytes(int DIM, int P, int Q, int nextra) {
  return sizeof(double) * (size_t)(DIM * P + DIM * DIM * Q + 2 * Q + DIM * P * Q + nextra) + sizeof(int32_t) * (size_t)P;
}

// geometry of element e into LDS; sc[q] = w detJ (no material factor), rq[q] = interpolated density (1 without rho)
template <int DIM>
__device__ __forceinline__ void tab_geometry(const TabArgs& A, int64_t e, const TabLds& L) {
  const int P = A.P, Q = A.Q, t = threadIdx.x;
  for (int a = t; a < P; a += TAB_THREADS) {
    const int32_t n = A.conn[e * P + a];
    L.en[a] = n;
#pragma unroll
    for (int j = 0; j < DIM; ++j) L.Xe[DIM * a + j] = A.X[DIM * (int64_t)n + j];
  }
  __syncthreads();
  for (int q = t; q < Q; q += TAB_THREADS) {
    double J[DIM][DIM];  // J[j][k] = dx_j / dxi_k   (utils.py:172-185)
#pragma unroll
    for (int j = 0; j < DIM; ++j)
#pragma unroll
      for (int k = 0; k < DIM; ++k) J[j][k] = 0.0;
    double r = A.rho ? 0.0 : 1.0;
    for (int a = 0; a < P; ++a) {
      const double* d = A.dN + ((int64_t)q * P + a) * DIM;
#pragma unroll
      for (int j = 0; j < DIM; ++j)
#pragma unroll
        for (int k = 0; k < DIM; ++k) J[j][k] += L.Xe[DIM * a + j] * d[k];
      if (A.rho) r += A.N[(int64_t)q * P + a] * A.rho[L.en[a]];  // utils.py:204-221
    }
    double* iv = L.inv + DIM * DIM * q;  // inverse of J (utils.py:244-260); the sign of det is kept, like the reference
    double det;
    if constexpr (DIM == 2) {
      det = J[0][0] * J[1][1] - J[0][1] * J[1][0];
      const double id = 1.0 / det;
      iv[0] = J[1][1] * id;
      iv[1] = -J[0][1] * id;
      iv[2] = -J[1][0] * id;
      iv[3] = J[0][0] * id;
    } else {
      const double c00 = J[1][1] * J[2][2] - J[1][2] * J[2][1];
      const double c01 = J[1][0] * J[2][2] - J[1][2] * J[2][0];
      const double c02 = J[1][0] * J[2][1] - J[1][1] * J[2][0];
      det = J[0][0] * c00 - J[0][1] * c01 + J[0][2] * c02;
      const double id = 1.0 / det;
      iv[0] = c00 * id;
      iv[1] = -(J[0][1] * J[2][2] - J[0][2] * J[2][1]) * id;
      iv[2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) * id;
      iv[3] = -c01 * id;
      iv[4] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) * id;
      iv[5] = -(J[0][0] * J[1][2] - J[0][2] * J[1][0]) * id;
      iv[6] = c02 * id;
      iv[7] = -(J[0][0] * J[2][1] - J[0][1] * J[2][0]) * id;
      iv[8] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) * id;
    }
    L.sc[q] = A.w[q] * det;
    L.rq[q] = r;
  }
  __syncthreads();
  for (int qa = t; qa < Q * P; qa += TAB_THREADS) {  // G[q][a][l] = sum_m dN[q][a][m] inv[m][l]   (utils.py:263)
    const int q = qa / P;
    const double* d = A.dN + (int64_t)qa * DIM;
    const double* iv = L.inv + DIM * DIM * q;
#pragma unroll
    for (int l = 0; l < DIM; ++l) {
      double s = 0.0;
#pragma unroll
      for (int m = 0; m < DIM; ++m) s += d[m] * iv[DIM * m + l];
      L.G[DIM * qa + l] = s;
    }
  }
  __syncthreads();
}

// DIM space dimension, DB dofs per node (DIM: elasticity, 1: scalar).
// MODE 0: add into the block-CSR values, one colour per launch; MODE 1: element matrices Ke (nelems, DB P, DB P)
template <int DIM, int DB, int MODE>
__global__ __launch_bounds__(TAB_THREADS) void k_tab_assemble(TabArgs A, const int32_t* __restrict__ order, int64_t count,
                                                            const int32_t* __restrict__ nptr,
                                                            const int32_t* __restrict__ nidx, double* __restrict__ out) {
  extern __shared__ double sh[];
  const int P = A.P, Q = A.Q, t = threadIdx.x;
  const TabLds L = tab_lds<DIM>(sh, P, Q, 0);
  for (int64_t k = blockIdx.x; k < count; k += gridDim.x) {
    const int64_t e = order ? order[k] : k;  // MODE 0: the elements of one colour; MODE 1: all elements
    tab_geometry<DIM>(A, e, L);
    for (int q = t; q < Q; q += TAB_THREADS) {
      const double r = L.rq[q], den = 1.0 + A.p * (1.0 - r);
      if (A.drho) {  // d RAMP / d rho_q times the direction interpolated to q   (pyfem.py:1948-1973)
        double dq = 0.0;
        for (int a = 0; a < P; ++a) dq += A.N[(int64_t)q * P + a] * A.drho[L.en[a]];
        L.sc[q] *= (1.0 + A.p) / (den * den) * dq;
      } else {
        L.sc[q] *= r / den;  // RAMP: rho / (1 + p (1 - rho))   (pyfem.py:1923-1945, 1279-1301)
      }
    }
    __syncthreads();
    for (int pair = t; pair < P * P; pair += TAB_THREADS) {
      const int a = pair / P, b = pair - a * P;
      double S[DIM][DIM];  // sum_q s G_a (x) G_b
#pragma unroll
      for (int i = 0; i < DIM; ++i)
#pragma unroll
        for (int j = 0; j < DIM; ++j) S[i][j] = 0.0;
      for (int q = 0; q < Q; ++q) {
        const double s = L.sc[q];
        const double* ga = L.G + DIM * (q * P + a);
        const double* gb = L.G + DIM * (q * P + b);
#pragma unroll
        for (int i = 0; i < DIM; ++i)
#pragma unroll
          for (int j = 0; j < DIM; ++j) S[i][j] += s * ga[i] * gb[j];
      }
      double tr = 0.0;
#pragma unroll
      for (int i = 0; i < DIM; ++i) tr += S[i][i];
      double K[DB][DB];
      if constexpr (DB == 1) {
        K[0][0] = tr;
      } else {
#pragma unroll
        for (int i = 0; i < DIM; ++i)
#pragma unroll
          for (int j = 0; j < DIM; ++j) K[i][j] = A.lam * S[i][j] + A.mu * S[j][i] + (i == j ? A.mu * tr : 0.0);
      }
      double* dst;
      int64_t stride;
      if (MODE == 1) {
        dst = out + e * (int64_t)(DB * DB * P * P) + (int64_t)(DB * a) * (DB * P) + DB * b;
        stride = DB * P;
      } else {
        const int32_t na = L.en[a], nb = L.en[b];
        int32_t lo = nptr[na], hi = nptr[na + 1] - 1;
        while (lo < hi) {
          const int32_t mid = (lo + hi) >> 1;
          if (nidx[mid] < nb) lo = mid + 1; else hi = mid;
        }
        dst = out + (int64_t)(DB * DB) * lo;  // no other element of this colour touches node na: plain adds
        stride = DB;
      }
#pragma unroll
      for (int i = 0; i < DB; ++i)
#pragma unroll
        for (int j = 0; j < DB; ++j) {
          if (MODE == 1) dst[i * stride + j] = K[i][j];
          else dst[i * stride + j] += K[i][j];
        }
    }
    __syncthreads();  // LDS is reused by the next element
  }
}

// d(phi^T K psi)/d rho_o = sum_e sum_q w detJ RAMP'(rho_q) N[q,o] form(phi, psi)_q      (pyfem.py:1873-1920, 1240-1276)
//   elasticity: form = eps(phi):C0:eps(psi);  scalar: form = grad phi . grad psi
template <int DIM, int DB>
__global__ __launch_bounds__(TAB_THREADS) void k_tab_sens(TabArgs A, const int32_t* __restrict__ order, int64_t count,
                                                         const double* __restrict__ phi, const double* __restrict__ psi,
                                                         double* __restrict__ out) {
  extern __shared__ double sh[];
  const int P = A.P, Q = A.Q, t = threadIdx.x;
  const TabLds L = tab_lds<DIM>(sh, P, Q, 2 * DB * P + Q);
  double* fe = L.extra;            // phi of the element's nodes (DB P)
  double* ge = L.extra + DB * P;   // psi
  double* tq = L.extra + 2 * DB * P;  // per quadrature point
  for (int64_t k = blockIdx.x; k < count; k += gridDim.x) {
    const int64_t e = order[k];
    tab_geometry<DIM>(A, e, L);
    for (int c = t; c < DB * P; c += TAB_THREADS) {
      const int64_t dof = DB * (int64_t)L.en[c / DB] + c % DB;
      fe[c] = phi[dof];
      ge[c] = psi[dof];
    }
    __syncthreads();
    for (int q = t; q < Q; q += TAB_THREADS) {
      double F[DB][DIM], H[DB][DIM];  // grad phi, grad psi
#pragma unroll
      for (int i = 0; i < DB; ++i)
#pragma unroll
        for (int l = 0; l < DIM; ++l) F[i][l] = H[i][l] = 0.0;
      for (int a = 0; a < P; ++a) {
        const double* g = L.G + DIM * (q * P + a);
#pragma unroll
        for (int i = 0; i < DB; ++i)
#pragma unroll
          for (int l = 0; l < DIM; ++l) {
            F[i][l] += fe[DB * a + i] * g[l];
            H[i][l] += ge[DB * a + i] * g[l];
          }
      }
      double form;
      if constexpr (DB == 1) {
        form = 0.0;
#pragma unroll
        for (int l = 0; l < DIM; ++l) form += F[0][l] * H[0][l];
      } else {
        double dd = 0.0, sym = 0.0, trF = 0.0, trH = 0.0;
#pragma unroll
        for (int i = 0; i < DIM; ++i) {
          trF += F[i][i];
          trH += H[i][i];
#pragma unroll
          for (int l = 0; l < DIM; ++l) {
            dd += F[i][l] * H[i][l];
            sym += F[i][l] * H[l][i];
          }
        }
        form = A.lam * trF * trH + A.mu * (dd + sym);
      }
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

// NonlinearPoisson2D:  -div( h(x) (1 + u^2) grad u ) = g  on quads given by their tables (pyfem.py:1332-1665).
//   h = 1 + 4y(1-y) sum_k ck[k] (1-x)^(n-1-k) x^k  with ck[k] = xdv[k] C(n-1,k)   (pyfem.py:1449-1470)
//   g = 1e4 x(1-x)(1-2x) y(1-y)(1-2y)                                             (pyfem.py:1427-1447)
// MODE 0: tangent into the scalar CSR values (one colour per launch; sym_only drops the second, nonsymmetric term and
//         leaves the SPD operator the preconditioner is built from)
//           Ke[j,i] = sum_q w detJ h (1+uq^2) gradN_j.gradN_i + sum_q 2 w detJ h uq (gradN_j.grad u) N[q,i]   (:1523-1562)
// MODE 1: the element tangents (nelems, P, P)
// MODE 2: residual ("compute_rhs", :1472-1521) added to the nodal vector:
//           re[j] = sum_q w detJ h (1+uq^2) gradN_j.grad u - sum_q w detJ g N[q,j];  like _assemble_rhs (pyfem.py:861-875)
//           only local nodes j < Q are scattered
struct NlArgs {
  const double* u;   // nodal state
  const double* ck;  // (n) design coefficients times binomials
  int n;
  int sym_only;
};

template <int MODE>
__global__ __launch_bounds__(TAB_THREADS) void k_tab_nlpoisson(TabArgs A, NlArgs B, const int32_t* __restrict__ order,
                                                             int64_t count, const int32_t* __restrict__ nptr,
                                                             const int32_t* __restrict__ nidx, double* __restrict__ out) {
  extern __shared__ double sh[];
  const int P = A.P, Q = A.Q, t = threadIdx.x;
  const TabLds L = tab_lds<2>(sh, P, Q, P + 5 * Q);
  double* ue = L.extra;       // (P)
  double* hq = ue + P;        // (Q) w detJ h (1 + uq^2)
  double* fq = hq + Q;        // (Q) 2 w detJ h uq
  double* gq = fq + Q;        // (Q) w detJ g
  double* gu = gq + Q;        // (2Q) grad u
  for (int64_t k = blockIdx.x; k < count; k += gridDim.x) {
    const int64_t e = order ? order[k] : k;
    tab_geometry<2>(A, e, L);
    for (int a = t; a < P; a += TAB_THREADS) ue[a] = B.u[L.en[a]];
    __syncthreads();
    for (int q = t; q < Q; q += TAB_THREADS) {
      double x = 0.0, y = 0.0, uq = 0.0, gx = 0.0, gy = 0.0;
      for (int a = 0; a < P; ++a) {
        const double Na = A.N[(int64_t)q * P + a];
        x += Na * L.Xe[2 * a];
        y += Na * L.Xe[2 * a + 1];
        uq += Na * ue[a];
        gx += ue[a] * L.G[2 * (q * P + a)];
        gy += ue[a] * L.G[2 * (q * P + a) + 1];
      }
      double h = 0.0;
      for (int c = 0; c < B.n; ++c) h += B.ck[c] * pow(1.0 - x, (double)(B.n - 1 - c)) * pow(x, (double)c);
      h = 1.0 + h * 4.0 * y * (1.0 - y);
      const double wd = L.sc[q];
      hq[q] = wd * h * (1.0 + uq * uq);
      fq[q] = 2.0 * wd * h * uq;
      gq[q] = wd * 1e4 * x * (1.0 - x) * (1.0 - 2.0 * x) * y * (1.0 - y) * (1.0 - 2.0 * y);
      gu[2 * q] = gx;
      gu[2 * q + 1] = gy;
    }
    __syncthreads();
    if (MODE == 2) {
      for (int j = t; j < P && j < Q; j += TAB_THREADS) {
        double r = 0.0;
        for (int q = 0; q < Q; ++q) {
          const double* g = L.G + 2 * (q * P + j);
          r += hq[q] * (g[0] * gu[2 * q] + g[1] * gu[2 * q + 1]) - gq[q] * A.N[(int64_t)q * P + j];
        }
        out[L.en[j]] += r;  // coloured
      }
    } else {
      for (int pair = t; pair < P * P; pair += TAB_THREADS) {
        const int j = pair / P, i = pair - j * P;
        double v = 0.0;
        for (int q = 0; q < Q; ++q) {
          const double* gj = L.G + 2 * (q * P + j);
          const double* gi = L.G + 2 * (q * P + i);
          v += hq[q] * (gj[0] * gi[0] + gj[1] * gi[1]);
          if (!B.sym_only) v += fq[q] * (gj[0] * gu[2 * q] + gj[1] * gu[2 * q + 1]) * A.N[(int64_t)q * P + i];
        }
        if (MODE == 1) {
          out[e * (int64_t)(P * P) + pair] = v;
        } else {
          const int32_t nj = L.en[j], ni = L.en[i];
          int32_t lo = nptr[nj], hi = nptr[nj + 1] - 1;
          while (lo < hi) {
            const int32_t mid = (lo + hi) >> 1;
            if (nidx[mid] < ni) lo = mid + 1; else hi = mid;
          }
          out[lo] += v;
        }
      }
    }
    __syncthreads();
  }
}

int check_tab(pf_ctx* ctx, int Q, const double* N, const double* dN, const double* wq, int kind, const char* who) {
  if (!ctx) return pf_fail(nullptr, PF_E_INVALID, "ctx is NULL");
  if (kind != PF_TAB_ELASTICITY && kind != PF_TAB_POISSON) return pf_fail(ctx, PF_E_INVALID, std::string(who) + ": unknown kind");
  if (!ctx->d || ctx->d != (kind == PF_TAB_POISSON ? 1 : ctx->ndims))
    return pf_fail(ctx, PF_E_INVALID, std::string(who) + ": the pattern's dofs per node must be 1 (Poisson) or the space "
                                                         "dimension (elasticity)");
  if (Q < 1 || Q > PF_TAB_MAX_QUADS || ctx->P > PF_TAB_MAX_NODES || !N || !dN || !wq)
    return pf_fail(ctx, PF_E_INVALID, std::string(who) + ": at most 32 nodes, 1 to 64 quadrature points, three tables");
  return PF_OK;
}

TabArgs tab_args(pf_ctx* ctx, int Q, const double* N, const double* dN, const double* wq, double E, double nu, double p,
                 const double* rho, const double* drho) {
  TabArgs A{};
  A.P = ctx->P; A.Q = Q; A.N = N; A.dN = dN; A.w = wq; A.X = ctx->X; A.conn = ctx->conn; A.rho = rho; A.drho = drho;
  // isotropic C0 as lam, mu: 3-D (pyfem.py:1752-1757) and plane stress (pyfem.py:1746-1750: c12 = lam, c33 = mu)
  A.lam = ctx->ndims == 3 ? E * nu / ((1.0 + nu) * (1.0 - 2.0 * nu)) : E * nu / (1.0 - nu * nu);
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

template <int MODE>
int launch_assemble(pf_ctx* ctx, const TabArgs& A, int scalar, const int32_t* order, int64_t count, double* out) {
  const unsigned g = tab_grid(count);
  const size_t lds = tab_lds_bytes(ctx->ndims, A.P, A.Q, 0);
  if (ctx->ndims == 2 && scalar) k_tab_assemble<2, 1, MODE><<<g, TAB_THREADS, lds, ctx->stream>>>(A, order, count, ctx->nptr, ctx->nidx, out);
  else if (ctx->ndims == 2) k_tab_assemble<2, 2, MODE><<<g, TAB_THREADS, lds, ctx->stream>>>(A, order, count, ctx->nptr, ctx->nidx, out);
  else if (scalar) k_tab_assemble<3, 1, MODE><<<g, TAB_THREADS, lds, ctx->stream>>>(A, order, count, ctx->nptr, ctx->nidx, out);
  else k_tab_assemble<3, 3, MODE><<<g, TAB_THREADS, lds, ctx->stream>>>(A, order, count, ctx->nptr, ctx->nidx, out);
  return PF_OK;
}

}  // namespace

extern "C" int pf_assemble_tab(pf_ctx* ctx, int Q, const double* N, const double* dN, const double* wq, int kind, double E,
                               double nu, double p, const double* rho, const double* drho, double* Kvals) {
  if (int rc = check_tab(ctx, Q, N, dN, wq, kind, "pf_assemble_tab")) return rc;
  PF_REQUIRE(Kvals, "pf_assemble_tab: Kvals is NULL");
  PF_REQUIRE(!drho || rho, "pf_assemble_tab: drho needs rho");
  PF_HIP(hipSetDevice(ctx->device));
  PF_HIP(hipMemsetAsync(Kvals, 0, sizeof(double) * (size_t)ctx->d * ctx->d * (size_t)ctx->nblocks, ctx->stream));
  if (int rc = ensure_colours(ctx)) return rc;
  const TabArgs A = tab_args(ctx, Q, N, dN, wq, E, nu, p, rho, drho);
  for (size_t c = 0; c + 1 < ctx->tab_color_ptr.size(); ++c) {
    const int64_t first = ctx->tab_color_ptr[c], count = ctx->tab_color_ptr[c + 1] - first;
    launch_assemble<0>(ctx, A, kind == PF_TAB_POISSON, ctx->tab_order + first, count, Kvals);
  }
  PF_LAUNCH_CHECK();
  return PF_OK;
}

extern "C" int pf_element_matrices_tab(pf_ctx* ctx, int Q, const double* N, const double* dN, const double* wq, int kind,
                                       double E, double nu, double p, const double* rho, const double* drho,
                                       double* Ke_out) {
  if (int rc = check_tab(ctx, Q, N, dN, wq, kind, "pf_element_matrices_tab")) return rc;
  PF_REQUIRE(Ke_out, "pf_element_matrices_tab: Ke_out is NULL");
  PF_REQUIRE(!drho || rho, "pf_element_matrices_tab: drho needs rho");
  PF_HIP(hipSetDevice(ctx->device));
  const TabArgs A = tab_args(ctx, Q, N, dN, wq, E, nu, p, rho, drho);
  launch_assemble<1>(ctx, A, kind == PF_TAB_POISSON, nullptr, ctx->nelems, Ke_out);
  PF_LAUNCH_CHECK();
  return PF_OK;
}

extern "C" int pf_sens_tab(pf_ctx* ctx, int Q, const double* N, const double* dN, const double* wq, int kind, double E,
                           double nu, double p, const double* rho, const double* phi, const double* psi, double* out) {
  if (int rc = check_tab(ctx, Q, N, dN, wq, kind, "pf_sens_tab")) return rc;
  PF_REQUIRE(rho && phi && psi && out, "pf_sens_tab: NULL argument");
  PF_HIP(hipSetDevice(ctx->device));
  PF_HIP(hipMemsetAsync(out, 0, sizeof(double) * (size_t)ctx->nnodes, ctx->stream));
  if (int rc = ensure_colours(ctx)) return rc;
  const TabArgs A = tab_args(ctx, Q, N, dN, wq, E, nu, p, rho, nullptr);
  const int db = kind == PF_TAB_POISSON ? 1 : ctx->ndims;
  const size_t lds = tab_lds_bytes(ctx->ndims, A.P, Q, 2 * db * A.P + Q);
  for (size_t c = 0; c + 1 < ctx->tab_color_ptr.size(); ++c) {
    const int64_t first = ctx->tab_color_ptr[c], count = ctx->tab_color_ptr[c + 1] - first;
    const unsigned g = tab_grid(count);
    const int32_t* order = ctx->tab_order + first;
    if (ctx->ndims == 2 && db == 1) k_tab_sens<2, 1><<<g, TAB_THREADS, lds, ctx->stream>>>(A, order, count, phi, psi, out);
    else if (ctx->ndims == 2) k_tab_sens<2, 2><<<g, TAB_THREADS, lds, ctx->stream>>>(A, order, count, phi, psi, out);
    else if (db == 1) k_tab_sens<3, 1><<<g, TAB_THREADS, lds, ctx->stream>>>(A, order, count, phi, psi, out);
    else k_tab_sens<3, 3><<<g, TAB_THREADS, lds, ctx->stream>>>(A, order, count, phi, psi, out);
  }
  PF_LAUNCH_CHECK();
  return PF_OK;
}

extern "C" int pf_nlpoisson_tab(pf_ctx* ctx, int Q, const double* N, const double* dN, const double* wq, int what,
                                const double* u, const double* ck, int nck, double* out) {
  if (int rc = check_tab(ctx, Q, N, dN, wq, PF_TAB_POISSON, "pf_nlpoisson_tab")) return rc;
  PF_REQUIRE(ctx->ndims == 2, "pf_nlpoisson_tab: the nonlinear Poisson model is two-dimensional (pyfem.py:1332)");
  PF_REQUIRE(u && out && (ck || nck == 0) && nck >= 0, "pf_nlpoisson_tab: NULL argument");
  PF_REQUIRE(what >= PF_NL_TANGENT && what <= PF_NL_RESIDUAL, "pf_nlpoisson_tab: unknown output");
  PF_HIP(hipSetDevice(ctx->device));
  TabArgs A = tab_args(ctx, Q, N, dN, wq, 0.0, 0.0, 0.0, nullptr, nullptr);
  NlArgs B{u, ck, nck, what == PF_NL_TANGENT_SPD};
  const size_t lds = tab_lds_bytes(2, A.P, Q, A.P + 5 * Q);
  if (what == PF_NL_ELEMENT_TANGENT) {
    k_tab_nlpoisson<1><<<tab_grid(ctx->nelems), TAB_THREADS, lds, ctx->stream>>>(A, B, nullptr, ctx->nelems, ctx->nptr,
                                                                                 ctx->nidx, out);
    PF_LAUNCH_CHECK();
    return PF_OK;
  }
  const size_t nout = what == PF_NL_RESIDUAL ? (size_t)ctx->nnodes : (size_t)ctx->nblocks;
  PF_HIP(hipMemsetAsync(out, 0, sizeof(double) * nout, ctx->stream));
  if (int rc = ensure_colours(ctx)) return rc;
  for (size_t c = 0; c + 1 < ctx->tab_color_ptr.size(); ++c) {
    const int64_t first = ctx->tab_color_ptr[c], count = ctx->tab_color_ptr[c + 1] - first;
    if (what == PF_NL_RESIDUAL)
      k_tab_nlpoisson<2><<<tab_grid(count), TAB_THREADS, lds, ctx->stream>>>(A, B, ctx->tab_order + first, count, ctx->nptr,
                                                                            ctx->nidx, out);
    else
      k_tab_nlpoisson<0><<<tab_grid(count), TAB_THREADS, lds, ctx->stream>>>(A, B, ctx->tab_order + first, count, ctx->nptr,
                                                                            ctx->nidx, out);
  }
  PF_LAUNCH_CHECK();
  return PF_OK;
}
