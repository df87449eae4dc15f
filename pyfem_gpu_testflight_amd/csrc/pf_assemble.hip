// NUMERIC assembly for gfx950: element quadrature fused with the global scatter.
//
// Replaces (reference file:line)
//   LinearElasticity.compute_jacobian            pyfem.py:1771-1795
//     _update_material_property (RAMP)           pyfem.py:1923-1945   (utils.py:155,204)
//     _compute_element_jacobian                  pyfem.py:2030-2068   (utils.py:172,189,225)
//     _compute_element_Bmat / _einsum_...        pyfem.py:1976-2027
//     ModelBase._assemble_jacobian               pyfem.py:921-931
//   Helmholtz._compute_element_jacobian_and_rhs  pyfem.py:2127-2177
//
// Design (MI355X-first, not the reference's "materialise Ke then sort COO triplets"):
//   GATHER scatter ("owner computes rows").  One lane owns one mesh node = d rows of K.  It walks
//   the node's incident elements (node->element lists from the symbolic phase, ascending => fixed
//   summation order => bitwise reproducible), evaluates only the d rows of each Ke that belong to
//   its node, and accumulates them at the precomputed 1-byte row positions in an LDS tile laid out
//   [slot][lane] (stride 65 doubles: conflict-free both for the lane-parallel accumulate and for
//   the slot-parallel write-out).  The wave then streams its 64 nodes' rows -- one contiguous
//   range of K values -- to HBM with 32-byte-per-lane coalesced stores.  Every K value is written
//   exactly once, nothing is read-modify-written in HBM, no atomics, no Ke round trip
//   (the reference moves ~5.9 KB/element through memory; algorithmic traffic here is ~0.33 KB).
//   Geometry (J, det, grad N) is recomputed by each of the <=4 owners of an element: fp64 VALU is
//   the cheaper resource next to a second pass over HBM.
//
//   ATOMIC scatter (cross-check / evidence only): one lane per element, full Ke, fp64
//   global_atomic_add_f64 into the same value array.
#include <stdlib.h>

#include "pf_common.h"

static constexpr double GP = 0.5773502691896258;  // 1/sqrt(3), == 1.0/np.sqrt(3.0)  (pyfem.py:87-90)

// ------------------------------------------------------------------------------------------
// element geometry at one quadrature point: det J, grad N (global), N
//
// Instruction diet (first GPU profile of this round: the v1 kernel was VALU-bound, ~520 VALU instructions per
// incident element, fp64 issuing at a quarter of the fp32 rate):
//   * J from the four edge differences of the quad (2 ops per entry instead of 4),
//   * 1/det and the RAMP quotient by v_rcp_f64 + two Newton steps (5 ops) instead of the IEEE
//     division expansion (~14 ops); det J is O(h^2), far from the ranges that expansion guards,
//   * the constitutive matrix is applied once per element, not per quadrature point: the point
//     loop only accumulates the four sums  S = sum_q s_q grad N_a (x) grad N_b  per coupled node b.
// Results differ from the reference's einsum order by rounding only (parity bar 1e-10 max|K|).
// ------------------------------------------------------------------------------------------
template <int P>
struct Elem {
  double x[P], y[P];
};

__device__ __forceinline__ double fast_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  double e = fma(-x, r, 1.0);
  r = fma(r, e, r);
  e = fma(-x, r, 1.0);
  return fma(r, e, r);
}

// `own` identifies the lane's own local node a without a dynamic register index (hipcc turns a
// ?: chain over Nx[a] into a scratch round trip):
//   quad: (s0, s1) = signs of the node's corner:  N_a = (1 + s0 xi)(1 + s1 eta)/4
//   tri : grad N_a = s0*row0(invJ) + s1*row1(invJ), N_a constant
struct Own {
  double s0, s1, Na;
  double Nxa, Nya;  // filled per quadrature point
};

template <int P>
__device__ __forceinline__ Own make_own(int a) {
  Own o;
  if constexpr (P == 4) {
    o.s0 = (a == 1 || a == 2) ? 1.0 : -1.0;
    o.s1 = (a >= 2) ? 1.0 : -1.0;
    o.Na = 0.0;
  } else {
    o.s0 = (a == 0) ? 1.0 : (a == 1 ? 0.0 : -1.0);
    o.s1 = (a == 1) ? 1.0 : (a == 0 ? 0.0 : -1.0);
    o.Na = (a == 2) ? (1 - 1.0 / 3 - 1.0 / 3) : (1.0 / 3);
  }
  o.Nxa = o.Nya = 0.0;
  return o;
}

// quad: 2x2 Gauss, points ordered (-,-),(+,-),(+,+),(-,+)   (pyfem.py:83-94, 262-284)
struct QuadEdges {
  double x10, x23, x30, x21, y10, y23, y30, y21;
};

__device__ __forceinline__ QuadEdges quad_edges(const Elem<4>& el) {
  QuadEdges e;
  e.x10 = el.x[1] - el.x[0]; e.x23 = el.x[2] - el.x[3]; e.x30 = el.x[3] - el.x[0]; e.x21 = el.x[2] - el.x[1];
  e.y10 = el.y[1] - el.y[0]; e.y23 = el.y[2] - el.y[3]; e.y30 = el.y[3] - el.y[0]; e.y21 = el.y[2] - el.y[1];
  return e;
}

template <int Q>
__device__ __forceinline__ void quad_point(const QuadEdges& ed, Own& own, double& det, double Nx[4], double Ny[4],
                                           double N[4]) {
  constexpr double xi = (Q == 0 || Q == 3) ? -GP : GP;
  constexpr double et = (Q == 0 || Q == 1) ? -GP : GP;
  // dN/dxi = [-a, a, b, -b], dN/deta = [-c, -d, d, c]                       (pyfem.py:272-283)
  constexpr double a = 0.25 * (1.0 - et), b = 0.25 * (1.0 + et), c = 0.25 * (1.0 - xi), d = 0.25 * (1.0 + xi);
  N[0] = 0.25 * (1.0 - xi) * (1.0 - et);
  N[1] = 0.25 * (1.0 + xi) * (1.0 - et);
  N[2] = 0.25 * (1.0 + xi) * (1.0 + et);
  N[3] = 0.25 * (1.0 - xi) * (1.0 + et);
  // J[j][k] = sum_l dN[l][k] X[l][j]  (utils.py:172-185), regrouped by element edges
  const double J00 = fma(a, ed.x10, b * ed.x23), J01 = fma(c, ed.x30, d * ed.x21);
  const double J10 = fma(a, ed.y10, b * ed.y23), J11 = fma(c, ed.y30, d * ed.y21);
  det = fma(J00, J11, -(J01 * J10));  // utils.py:189-200 (closed form instead of LAPACK LU)
  const double r = fast_rcp(det);
  const double i00 = J11 * r, i01 = -(J01 * r), i10 = -(J10 * r), i11 = J00 * r;  // utils.py:244-247
  // Ngrad[l] = dN[l][0] invJ[0][:] + dN[l][1] invJ[1][:]                    (utils.py:263)
  const double ax = a * i00, bx = b * i00, cx = c * i10, dx = d * i10;
  Nx[0] = -ax - cx; Nx[1] = ax - dx; Nx[2] = bx + dx; Nx[3] = cx - bx;
  const double ay = a * i01, by = b * i01, cy = c * i11, dy = d * i11;
  Ny[0] = -ay - cy; Ny[1] = ay - dy; Ny[2] = by + dy; Ny[3] = cy - by;
  const double tx = fma(own.s0, xi, 1.0), ty = fma(own.s1, et, 1.0);
  const double da = (0.25 * own.s0) * ty, db = (0.25 * own.s1) * tx;
  own.Nxa = fma(da, i00, db * i10);
  own.Nya = fma(da, i01, db * i11);
  own.Na = (0.25 * tx) * ty;
}

// tri: one point (1/3,1/3); dN = [[1,0],[0,1],[-1,-1]]   (pyfem.py:68-80, 365-377)
__device__ __forceinline__ void tri_point(const Elem<3>& el, Own& own, double& det, double Nx[3], double Ny[3],
                                          double N[3]) {
  N[0] = 1.0 / 3;
  N[1] = 1.0 / 3;
  N[2] = 1 - 1.0 / 3 - 1.0 / 3;
  const double J00 = el.x[0] - el.x[2], J01 = el.x[1] - el.x[2];
  const double J10 = el.y[0] - el.y[2], J11 = el.y[1] - el.y[2];
  det = fma(J00, J11, -(J01 * J10));  // NOT |det|: the reference keeps the sign (pyfem.py:78 TODO)
  const double r = fast_rcp(det);
  const double i00 = J11 * r, i01 = -(J01 * r), i10 = -(J10 * r), i11 = J00 * r;
  Nx[0] = i00;
  Ny[0] = i01;
  Nx[1] = i10;
  Ny[1] = i11;
  Nx[2] = -i00 - i10;
  Ny[2] = -i01 - i11;
  // own.s0 / s1 are compile-time constants at every call site (local index known after inlining): pick the form the
  // compiler can fold (0.0 * x is not foldable under IEEE rules, 1.0 * x is)
  own.Nxa = (own.s1 == 0.0) ? own.s0 * i00 : (own.s0 == 0.0) ? own.s1 * i10 : fma(own.s0, i00, own.s1 * i10);
  own.Nya = (own.s1 == 0.0) ? own.s0 * i01 : (own.s0 == 0.0) ? own.s1 * i11 : fma(own.s0, i01, own.s1 * i11);
}

template <int P>
__device__ __forceinline__ void load_elem(const double* __restrict__ X, const int32_t nd[P], Elem<P>& el) {
#pragma unroll
  for (int l = 0; l < P; ++l) {
    const double2 xy = *reinterpret_cast<const double2*>(X + 2 * (int64_t)nd[l]);
    el.x[l] = xy.x;
    el.y[l] = xy.y;
  }
}

template <int P>
__device__ __forceinline__ void load_conn(const int32_t* __restrict__ conn, int64_t e, int32_t nd[P]) {
  if constexpr (P == 4) {
    const int4 c = *reinterpret_cast<const int4*>(conn + 4 * e);
    nd[0] = c.x; nd[1] = c.y; nd[2] = c.z; nd[3] = c.w;
  } else {
#pragma unroll
    for (int l = 0; l < P; ++l) nd[l] = conn[P * e + l];
  }
}

// ------------------------------------------------------------------------------------------
// physics: rows of Ke that belong to local node `a`;  V values per coupled node b.
//   point():  accumulate the quadrature sums S (NS per coupled node)
//   finish(): apply the constant material matrix once
// ------------------------------------------------------------------------------------------
// RAMP factor at one quadrature point: rho_q = sum_l N_l rho_l (utils.py:204-221), rho_q / (1 + p (1 - rho_q)) (pyfem.py:1944)
template <int P>
__device__ __forceinline__ double ramp_at(double p, const double N[P], const double rho_e[P]) {
  double rq = N[0] * rho_e[0];
#pragma unroll
  for (int l = 1; l < P; ++l) rq = fma(N[l], rho_e[l], rq);
  return rq * fast_rcp(fma(p, 1.0 - rq, 1.0));
}

struct ElasticityParams {
  double c11, c12, c33, p;     // C0 = E/(1-nu^2) [[1,nu,0],[nu,1,0],[0,0,(1-nu)/2]]  (pyfem.py:1746-1750)
  const double* rho;           // nodal density or nullptr
};

template <int P, bool RHO>
struct Elasticity {
  static constexpr int V = 4;   // 2x2 block, row-major
  static constexpr int NS = 4;  // S = [Sxx, Sxy, Syx, Syy],  S_uv = sum_q s dN_a/du dN_b/dv
  using Params = ElasticityParams;

  __device__ __forceinline__ static void point(const Params& pr, double det, double w, const double N[P],
                                               const double Nx[P], const double Ny[P], const double rho_e[P],
                                               const Own& own, double S[P][NS]) {
    double s = det * w;
    if constexpr (RHO) {
      s *= ramp_at<P>(pr.p, N, rho_e);
    }
    const double gx = s * own.Nxa, gy = s * own.Nya;
#pragma unroll
    for (int b = 0; b < P; ++b) {
      S[b][0] = fma(gx, Nx[b], S[b][0]);
      S[b][1] = fma(gx, Ny[b], S[b][1]);
      S[b][2] = fma(gy, Nx[b], S[b][2]);
      S[b][3] = fma(gy, Ny[b], S[b][3]);
    }
  }
  // B_a^T C0 B_b with B = [[Nx,0],[0,Ny],[Ny,Nx]]  (pyfem.py:1992-1997, 2016-2027)
  __device__ __forceinline__ static void finish(const Params& pr, const double S[P][NS], double v[P][V]) {
#pragma unroll
    for (int b = 0; b < P; ++b) {
      v[b][0] = fma(pr.c11, S[b][0], pr.c33 * S[b][3]);
      v[b][1] = fma(pr.c12, S[b][1], pr.c33 * S[b][2]);
      v[b][2] = fma(pr.c12, S[b][2], pr.c33 * S[b][1]);
      v[b][3] = fma(pr.c11, S[b][3], pr.c33 * S[b][0]);
    }
  }
};

struct HelmholtzParams {
  double r2;  // r0^2
  const double* rho;  // unused
};

template <int P>
struct Helmholtz {
  static constexpr int V = 2;   // [K, R]
  static constexpr int NS = 2;  // [stiffness sum, mass sum]
  using Params = HelmholtzParams;
  __device__ __forceinline__ static void point(const Params&, double det, double w, const double N[P],
                                               const double Nx[P], const double Ny[P], const double*, const Own& own,
                                               double S[P][NS]) {
    const double s = det * w;
    const double na = s * own.Na, gx = s * own.Nxa, gy = s * own.Nya;
#pragma unroll
    for (int b = 0; b < P; ++b) {
      S[b][0] = fma(gx, Nx[b], fma(gy, Ny[b], S[b][0]));  // pyfem.py:2127-2131
      S[b][1] = fma(na, N[b], S[b][1]);                   // Re, pyfem.py:2134-2136
    }
  }
  __device__ __forceinline__ static void finish(const Params& pr, const double S[P][NS], double v[P][V]) {
#pragma unroll
    for (int b = 0; b < P; ++b) {
      v[b][1] = S[b][1];
      v[b][0] = fma(pr.r2, S[b][0], S[b][1]);  // Ke = r0^2 gradN.gradN + Re, pyfem.py:2176
    }
  }
};

struct PoissonParams {
  double p;           // RAMP penalisation of the conductivity
  const double* rho;  // nodal density or nullptr
};

// LinearPoisson: Ke = sum_q kappa_q detJ w gradN gradN^T, kappa_q = RAMP(rho_q)   (pyfem.py:1176-1187, 1279-1301;
// the reference stores kappa0 but never applies it)
template <int P, bool RHO>
struct Poisson {
  static constexpr int V = 1;
  static constexpr int NS = 1;
  using Params = PoissonParams;
  __device__ __forceinline__ static void point(const Params& pr, double det, double w, const double N[P],
                                               const double Nx[P], const double Ny[P], const double rho_e[P],
                                               const Own& own, double S[P][NS]) {
    double s = det * w;
    if constexpr (RHO) {
      s *= ramp_at<P>(pr.p, N, rho_e);
    }
    const double gx = s * own.Nxa, gy = s * own.Nya;
#pragma unroll
    for (int b = 0; b < P; ++b) S[b][0] = fma(gx, Nx[b], fma(gy, Ny[b], S[b][0]));
  }
  __device__ __forceinline__ static void finish(const Params&, const double S[P][NS], double v[P][V]) {
#pragma unroll
    for (int b = 0; b < P; ++b) v[b][0] = S[b][0];
  }
};

// rows of Ke owned by local node a, from element coordinates (and nodal densities) in registers
template <int P, class Phys, bool RHO>
__device__ __forceinline__ void element_rows_core(const typename Phys::Params& pr, const Elem<P>& el,
                                                  const double rho_e[P], int a, double v[P][Phys::V]) {
  double S[P][Phys::NS];
#pragma unroll
  for (int b = 0; b < P; ++b)
#pragma unroll
    for (int k = 0; k < Phys::NS; ++k) S[b][k] = 0.0;
  double det, Nx[P], Ny[P], N[P];
  Own own = make_own<P>(a);
  if constexpr (P == 4) {
    const QuadEdges ed = quad_edges(el);
    quad_point<0>(ed, own, det, Nx, Ny, N); Phys::point(pr, det, 1.0, N, Nx, Ny, rho_e, own, S);
    quad_point<1>(ed, own, det, Nx, Ny, N); Phys::point(pr, det, 1.0, N, Nx, Ny, rho_e, own, S);
    quad_point<2>(ed, own, det, Nx, Ny, N); Phys::point(pr, det, 1.0, N, Nx, Ny, rho_e, own, S);
    quad_point<3>(ed, own, det, Nx, Ny, N); Phys::point(pr, det, 1.0, N, Nx, Ny, rho_e, own, S);
  } else {
    tri_point(el, own, det, Nx, Ny, N);
    Phys::point(pr, det, 0.5, N, Nx, Ny, rho_e, own, S);
  }
  Phys::finish(pr, S, v);
}

template <int P, class Phys, bool RHO>
__device__ __forceinline__ void element_rows(const typename Phys::Params& pr, const double* __restrict__ X,
                                             const int32_t nd[P], int a, double v[P][Phys::V]) {
  Elem<P> el;
  load_elem<P>(X, nd, el);
  double rho_e[P];
  if constexpr (RHO) {
#pragma unroll
    for (int l = 0; l < P; ++l) rho_e[l] = pr.rho[nd[l]];  // utils.py:155-168
  }
  element_rows_core<P, Phys, RHO>(pr, el, rho_e, a, v);
}

// ------------------------------------------------------------------------------------------
// GATHER assembly v4 (default): lane = incident (node, element) pair, tile = up to npt consecutive nodes.
//
// The measurements that led here (profiles/r01_*, r02_*): with lane = node the accumulator tile is
// 18.7 KB per 64 nodes, which caps a CU at 8 tiles; every tile then pays its dependent load chain,
// its quadrature and its store drain back to back and boundary/irregular nodes leave lanes idle.  Here
//   * each lane integrates ONE incident element for one node (perfect load balance for any
//     valence); a tile is the npt nodes (16 for quads) whose <= 64 incident pairs fill the wave;
//   * the tile's rows are ONE contiguous segment of the value array, and the LDS accumulator is an
//     exact image of that segment (4.6 KB for quads): contributions go in with LDS fp64 atomic adds
//     at (row offset + position byte), the write-out is a straight 16-byte-per-lane copy -- no maps,
//     no barriers;
//   * waves are persistent and software-pipelined: row pointers two tiles ahead, the visit records
//     one tile ahead, the coordinate gathers of the next tile in flight during the quadrature of the
//     current one.  Every wait targets loads that were issued BEFORE the latest stores (vmcnt retires
//     in issue order, stores included), so the store drain of tile t overlaps the work of tile t+1.
// v4 (round 2) is an INSTRUCTION diet of v3: the r02 ISA count showed ~330 wave instructions per 10-node triangle
// tile, 4 issue cycles each on a 16-lane SIMD -- more than half of the kernel's time is instruction issue, most of it
// integer bookkeeping, not fp64:
//   * the element is rotated so that the lane's node is local node 0 (pf_pattern.hip k_build_visits): no selects on
//     the local index, constant shape-function signs;
//   * every global access is a BUFFER access (raw buffer resources built with scalar instructions): one 32-bit
//     offset per gather instead of 64-bit address arithmetic, and the hardware's range check replaces every clamp --
//     lanes without a pair read zeros (= node 0, in range) and store nothing; the copy-out is `buffer_store_dwordx4`
//     with an immediate offset per piece against a resource that ends where the tile's segment ends;
//   * the image is cleared with flat 16-byte LDS stores; the copy-out reads LDS at immediate offsets from two
//     per-lane base addresses computed once per tile.
// Later in round 2 (profiles/r02_assembly_ab.md sections 7-15): the contributions to a node's diagonal block are summed
// across its lanes before they go to the LDS (v4_seg_step: DPP scan under EXEC); the records shrink to 8 bytes where
// the numbering allows it (12 for triangles otherwise); the copy-out stores are NON-TEMPORAL (K is not re-read by the
// launch: -14..-20 %); triangles walk their tiles XCD-aware, and on scattered numberings along a Morton curve.
// Summation order inside a tile is fixed by the instruction stream (same lanes, same addresses every
// run), so results are bitwise reproducible although the adds are LDS atomics.
// dynamic LDS: double img[V*plane + 32*NOUT]  (V planes of `plane` = blocks of the largest tile, made odd)
// ------------------------------------------------------------------------------------------
typedef unsigned int pf_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int pf_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int pf_u32x3 __attribute__((ext_vector_type(3)));
typedef double pf_f64x2 __attribute__((ext_vector_type(2)));
typedef __amdgpu_buffer_rsrc_t pf_rsrc;
#ifndef PF_AB_REC_AUX
#define PF_AB_REC_AUX 0  // cache-policy bits of the visit-record loads (read once per launch)
#endif
#ifndef PF_AB_STORE_AUX
#define PF_AB_STORE_AUX 2  // cache-policy bits of the copy-out stores: nt.  (tools/build_variant.sh A/B: 0 none, 1 sc0, 2 nt, 16 sc1)
#endif

// raw buffer (stride 0): offsets and num_records in bytes; out-of-range loads return 0, stores are dropped
__device__ __forceinline__ pf_rsrc make_rsrc(const void* base, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ double2 buf_ld_f64x2(pf_rsrc r, uint32_t off) {
  const pf_f64x2 v = __builtin_bit_cast(pf_f64x2, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0));
  return make_double2(v.x, v.y);
}
__device__ __forceinline__ double buf_ld_f64(pf_rsrc r, uint32_t off) {
  return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, (int)off, 0, 0));
}

// packed word of a visit record: row positions of the rotated element's nodes and the node's slot in its tile.
// REC = bytes per record (pf_pattern.hip k_build_visits / k_build_rec8 / k_build_rec_tri):
//   16 (any mesh; quads' fallback)  {o1, o2, o3, w}, w = pos0 | pos1<<6 | pos2<<12 | pos3<<18 | a<<24 | slot<<26
//   12 (triangles' fallback)        {o1, o2, w}, the same w
//    8 quads      d1 = (off3>>6)&127 | pos0<<7 | pos1<<12 | pos2<<17 | pos3<<22 | slot<<27
//    8 triangles  d1 = pos0 | pos1<<5 | pos2<<10 | slot<<15 | a<<20
template <int REC, int P>
__device__ __forceinline__ int rec_pos(uint32_t w, int k) {
  if constexpr (REC == 8) return (int)((w >> ((P == 4 ? 7 : 0) + 5 * k)) & 31u);
  else return (int)((w >> (6 * k)) & 63u);
}
template <int REC, int P>
__device__ __forceinline__ int rec_slot(uint32_t w) {
  return (int)((w >> (REC == 8 ? (P == 4 ? 27 : 15) : 26)) & 31u);
}

struct V4Rec {        // stage B: the lane's visit record + the tile's row pointers (one tile ahead)
  int4 rec;           // {o1, o2, o3, w}: rotated element, own node first (pf_pattern.hip k_build_visits)
  int32_t rp;         // lanes 0..nodes of the tile: nptr[n0 + lane]
  int32_t n0, base;   // first node / first block of the tile
  bool active;
};

template <int P, bool RHO>
struct V4Visit {      // stage C: what the quadrature needs (gathers in flight during the previous tile)
  Elem<P> el;         // (the nodal densities are NOT part of the two ping-pong sets: see v4_load_rho)
  uint32_t w;         // packed positions
  int32_t roff;       // first block of the lane's node inside the tile segment
  bool active;
};

struct V4Tile {       // wave-uniform tile scalars (one 16-byte scalar load, two tiles ahead)
  int32_t vb, nv, base, nblk, n0, nn;
  bool deep;          // a node of the tile has more than 8 incident elements (fourth step of the diagonal pre-reduction)
};

// NOTE: the pipeline stages are free __forceinline__ functions, not lambdas: hipcc keeps local arrays of
// a lambda that is inlined at two call sites in scratch memory (seen twice in this file's history).
// All prefetch loads are UNCONDITIONAL: a load under `if (active)` that is merged with a default at the join makes
// hipcc wait for it on the spot (s_waitcnt vmcnt right behind the issue), which exposes a full memory latency per
// tile.  The buffer resources make that safe without clamps.
__device__ __forceinline__ V4Tile v4_load_tile(int64_t t, int64_t t1, const int4* __restrict__ tile_desc) {
  V4Tile T;  // t is wave-uniform; look-ahead past the wave's last tile re-reads that tile's descriptor with no pairs
  const int4 dsc = tile_desc[min(t, t1 - 1)];
  T.n0 = dsc.x;
  T.vb = dsc.y;
  T.base = dsc.z;
  T.nn = dsc.w & 63;
  T.deep = (dsc.w & 128) != 0;
  T.nv = (t < t1) ? ((dsc.w >> 8) & 255) : 0;
  T.nblk = (int32_t)((uint32_t)dsc.w >> 16);
  return T;
}

// step j of a wave's walk (n steps): descriptor t0 + j tstride; look-ahead past the end re-reads the last descriptor
// with no pairs
__device__ __forceinline__ V4Tile v4_walk_tile(int64_t j, int64_t n, int64_t t0, int64_t tstride,
                                               const int4* __restrict__ tile_desc) {
  const int64_t idx = t0 + min(j, n - 1) * tstride;
  return v4_load_tile(j < n ? idx : idx + 1, idx + 1, tile_desc);
}

template <int REC>
__device__ __forceinline__ V4Rec v4_load_rec(const V4Tile& T, int lane, const int4* __restrict__ vis_rec,
                                             const int32_t* __restrict__ nptr) {
  V4Rec B;
  B.active = lane < T.nv;
  B.n0 = T.n0;
  B.base = T.base;
  // this tile's records only, and its nn + 1 row pointers
  const pf_rsrc rr = make_rsrc(reinterpret_cast<const unsigned char*>(vis_rec) + (int64_t)T.vb * REC, (uint32_t)T.nv * (uint32_t)REC);
  const pf_rsrc rn = make_rsrc(nptr + T.n0, (uint32_t)(T.nn + 1) * 4u);
  if constexpr (REC == 8) {  // rec.x / rec.y hold the two dwords
    const pf_u32x2 r = __builtin_amdgcn_raw_buffer_load_b64(rr, lane * 8, 0, PF_AB_REC_AUX);
    B.rec = make_int4((int)r.x, (int)r.y, 0, 0);
  } else if constexpr (REC == 12) {
    const pf_u32x3 r = __builtin_amdgcn_raw_buffer_load_b96(rr, lane * 12, 0, PF_AB_REC_AUX);
    B.rec = make_int4((int)r.x, (int)r.y, 0, (int)r.z);
  } else {
    const pf_u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rr, lane * 16, 0, PF_AB_REC_AUX);
    B.rec = make_int4((int)r.x, (int)r.y, (int)r.z, (int)r.w);  // triangles: .z is unused -- see v4_load_visit
  }
  B.rp = (int32_t)__builtin_amdgcn_raw_buffer_load_b32(rn, lane * 4, 0, 0);
  return B;
}

template <int P, bool RHO, int REC>
__device__ __forceinline__ void v4_load_visit(const V4Rec& B, pf_rsrc RX, pf_rsrc Rrho, V4Visit<P, RHO>& C) {
  if constexpr (REC == 8) {
    const uint32_t d0 = (uint32_t)B.rec.x, d1 = (uint32_t)B.rec.y;
    const int ln = rec_slot<8, P>(d1);
    C.w = d1;
    C.active = B.active;
    C.roff = __shfl(B.rp, ln) - B.base;
    // a lane without a pair holds an all-zero record: offsets 0, i.e. node n0 four times -- in range
    const int32_t n = B.n0 + ln;
    // quads: three 13-bit offsets; triangles: two 16-bit offsets
    const int32_t o1 = n + (P == 4 ? ((int32_t)(d0 << 19) >> 19) : ((int32_t)(d0 << 16) >> 16));
    const int32_t o2 = n + (P == 4 ? ((int32_t)(d0 << 6) >> 19) : ((int32_t)d0 >> 16));
    const int32_t o3 = n + ((int32_t)(((d0 >> 26) | ((d1 & 127u) << 6)) << 19) >> 19);
    const double2 p0 = buf_ld_f64x2(RX, (uint32_t)n * 16u);
    const double2 p1 = buf_ld_f64x2(RX, (uint32_t)o1 * 16u);
    const double2 p2 = buf_ld_f64x2(RX, (uint32_t)o2 * 16u);
    C.el.x[0] = p0.x; C.el.y[0] = p0.y;
    C.el.x[1] = p1.x; C.el.y[1] = p1.y;
    C.el.x[2] = p2.x; C.el.y[2] = p2.y;
    if constexpr (P == 4) {
      const double2 p3 = buf_ld_f64x2(RX, (uint32_t)o3 * 16u);
      C.el.x[3] = p3.x; C.el.y[3] = p3.y;
    }
    return;
  }
  // Triangles never read rec.z.  hipcc then hands that register of the in-flight 16-byte load to a temporary, and the
  // write-after-write hazard makes it wait for the record right behind its issue (r02 ISA: s_waitcnt vmcnt(1) one
  // instruction after the load -- a full memory latency per tile; 0.181 -> 0.212 ms on 4 M triangles).  Keeping the
  // component alive until the record is consumed removes the wait.
  if constexpr (P == 3 && REC == 16) asm volatile("" ::"v"(B.rec.z));
  const uint32_t w = (uint32_t)B.rec.w;
  const int ln = (int)((w >> 26) & 31);
  C.w = w;
  C.active = B.active;
  C.roff = __shfl(B.rp, ln) - B.base;
  // a lane without a pair holds an all-zero record: its gathers read node n0 and node 0, both in range
  const uint32_t n = (uint32_t)(B.n0 + ln);
  const double2 p0 = buf_ld_f64x2(RX, n * 16u);
  const double2 p1 = buf_ld_f64x2(RX, (uint32_t)B.rec.x * 16u);
  const double2 p2 = buf_ld_f64x2(RX, (uint32_t)B.rec.y * 16u);
  C.el.x[0] = p0.x; C.el.y[0] = p0.y;
  C.el.x[1] = p1.x; C.el.y[1] = p1.y;
  C.el.x[2] = p2.x; C.el.y[2] = p2.y;
  if constexpr (P == 4) {
    const double2 p3 = buf_ld_f64x2(RX, (uint32_t)B.rec.z * 16u);
    C.el.x[3] = p3.x; C.el.y[3] = p3.y;
  }
}

// The nodal densities of the lane's (rotated) element, from the record B of the tile they belong to.  ONE register set for
// them, not one per ping-pong visit set: the densities of tile j + 1 are gathered right after the quadrature of tile j has
// consumed those of tile j (v4_tile), into the same registers, and arrive during tile j's LDS adds and copy-out.  With
// them inside both visit sets the density kernel needed 136-138 VGPRs, now 132-134: still three waves per SIMD (12 per CU
// against the 16 of the density-free kernel; 0.296 ms = 0.58 of the HBM peak either way, r03).  Two ways to the last six
// registers were built and measured and are NOT kept: compiling the kernel for four waves per SIMD
// (amdgpu_waves_per_eu: 128 VGPRs + 24 bytes of scratch per lane) ran at 0.369 ms, the spills sit in the quadrature loop;
// parking the four RAMP factors in LDS so that the densities die before the loop made the scheduler hoist the reads
// back to the top (154 VGPRs; 144 with compiler barriers between the points).
template <int P, int REC>
__device__ __forceinline__ void v4_load_rho(const V4Rec& B, pf_rsrc Rrho, double rho_e[P]) {
  if constexpr (REC == 8) {
    const uint32_t d0 = (uint32_t)B.rec.x, d1 = (uint32_t)B.rec.y;
    const int32_t n = B.n0 + rec_slot<8, P>(d1);
    const int32_t o1 = n + (P == 4 ? ((int32_t)(d0 << 19) >> 19) : ((int32_t)(d0 << 16) >> 16));
    const int32_t o2 = n + (P == 4 ? ((int32_t)(d0 << 6) >> 19) : ((int32_t)d0 >> 16));
    const int32_t o3 = n + ((int32_t)(((d0 >> 26) | ((d1 & 127u) << 6)) << 19) >> 19);
    rho_e[0] = buf_ld_f64(Rrho, (uint32_t)n * 8u);
    rho_e[1] = buf_ld_f64(Rrho, (uint32_t)o1 * 8u);
    rho_e[2] = buf_ld_f64(Rrho, (uint32_t)o2 * 8u);
    if constexpr (P == 4) rho_e[3] = buf_ld_f64(Rrho, (uint32_t)o3 * 8u);
  } else {
    const uint32_t n = (uint32_t)(B.n0 + (int)(((uint32_t)B.rec.w >> 26) & 31));
    rho_e[0] = buf_ld_f64(Rrho, n * 8u);
    rho_e[1] = buf_ld_f64(Rrho, (uint32_t)B.rec.x * 8u);
    rho_e[2] = buf_ld_f64(Rrho, (uint32_t)B.rec.y * 8u);
    if constexpr (P == 4) rho_e[3] = buf_ld_f64(Rrho, (uint32_t)B.rec.z * 8u);
  }
}

// one step of the segmented inclusive scan of v4_tile: x[k] += x[k] of the lane SH places down the row if that lane
// works for the same node (keys are sorted along the lanes, so equal keys SH apart mean one run).
// The shifted values are fetched by every lane (a DPP source lane must be active), the adds run under the lane mask:
// written as `x += same ? y : 0` hipcc selects both halves of every y (2 v_cndmask per value and step -- 64 of the 240
// vector instructions of a triangle pass, r02 ISA, and the triangle kernel is bound by instruction issue); the volatile
// asm keeps the add inside the branch, i.e. under EXEC, at one instruction per value.
template <int SH, int NV>
__device__ __forceinline__ void v4_seg_step(int key, double* x) {
  const bool same = __builtin_amdgcn_update_dpp(-1, key, 0x110 + SH, 0xf, 0xf, false) == key;  // row_shr:SH
  double y[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x[k]), 0x110 + SH, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x[k]), 0x110 + SH, 0xf, 0xf, true);
    y[k] = __hiloint2double(hi, lo);
  }
  if (same) {
#pragma unroll
    for (int k = 0; k < NV; ++k) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x[k]) : "v"(y[k]));
    // the next DPP reads x: 2 wait states between a VALU write and a DPP read of one VGPR (ISA manual, manually
    // inserted wait states); hipcc's hazard recogniser does not look inside inline asm
    asm volatile("s_nop 1");
  }
}

#ifdef PF_EXPERIMENTS
// PF_ABL_WRAP_MB (experiments build only): every tile's copy-out lands inside the first N MB of the value array, so the
// written lines are overwritten in cache instead of streaming to HBM -- wrong K, for one question only: how fast is the
// kernel when its HBM write stream is (almost) taken away?
__device__ int64_t g_abl_wrap_blocks = 0;
// PF_ABL_MASK (experiments build only; wrong results, timing only): 1 no LDS adds, 2 no LDS clear, 4 no quadrature
// (the values are the gathered coordinates), 8 no copy-out
__device__ int g_abl_mask = 0;
#endif

// integrate tile Tcur from Cuse into the LDS image and copy the finished segment out.
// Image layout: V planes of `plane` doubles, value k of the tile's block B at img[k*plane + B].  With the
// blocks' 4 values adjacent (the global layout) every lane of an add instruction would land on the same
// 8 of the 32 banks (PMC counters: a third of all LDS cycles were bank conflicts); by planes, lanes that
// touch different blocks touch different banks.  The copy-out re-interleaves two planes per 16-byte piece.
template <int P, class Phys, bool RHO, int NOUT, bool COMB, int REC>
__device__ __forceinline__ void v4_tile(const typename Phys::Params& pr, const V4Tile& Tcur,
                                        const V4Visit<P, RHO>& Cuse, double* img, int plane, int lane,
                                        double* __restrict__ out0, double* __restrict__ out1, double rho_e[P],
                                        const V4Rec& Bnext, pf_rsrc Rrho) {
  constexpr int V = Phys::V;
  // clear the image (the previous tile's copy-out reads are complete: one wave, LDS executes in order)
#ifdef PF_EXPERIMENTS
  const int abl = g_abl_mask;
  if (!(abl & 2))
#endif
  {  // whole 1 KiB wave stores, a wave-uniform trip count (scalar loop): the allocation is rounded up to that
    pf_f64x2* img2 = reinterpret_cast<pf_f64x2*>(img) + lane;
    const int nzit = (V * plane + 127) >> 7;
    const pf_f64x2 z = {0.0, 0.0};
    for (int it = 0; it < nzit; ++it) img2[64 * it] = z;
  }

  if (Cuse.active) {
    double v[P][V];
#ifdef PF_EXPERIMENTS
    if (abl & 4) {
#pragma unroll
      for (int b = 0; b < P; ++b)
#pragma unroll
        for (int k = 0; k < V; ++k) v[b][k] = Cuse.el.x[b] + (double)k * Cuse.el.y[b];
    } else
#endif
    element_rows_core<P, Phys, RHO>(pr, Cuse.el, rho_e, 0, v);  // the lane's node is local node 0 (rotated record)
    double* row = img + Cuse.roff;
    bool add_diag = true;
    if constexpr (COMB) {
      // The incident pairs of a node are consecutive lanes and, with the rotated records, every one of them holds a
      // contribution to the node's DIAGONAL block in v[0]: valence-many adds to the same LDS address in one
      // instruction, which the LDS serialises (r02: this, not instruction issue, bounds the triangle kernel -- the
      // v3 kernel spread the same adds over its P instructions by accident of the local index).  Sum them across the
      // lanes first: a segmented scan over the node slot inside each row of 16 lanes (DPP row shifts: plain VALU,
      // no LDS), after which only the last lane of a node's run adds.  A run that crosses a row boundary adds twice.
      // Fixed lane order, so the result stays bitwise reproducible.
      const int key = rec_slot<REC, P>(Cuse.w);
      // 2x2 blocks: the diagonal block of a symmetric element matrix is symmetric up to the rounding of its two
      // off-diagonal entries; scan three values and use the sum of [0][1] for both
      constexpr int NV = (V == 4) ? 3 : V;
      double xs[NV];
      xs[0] = v[0][0];
      if constexpr (V == 4) { xs[1] = v[0][1]; xs[2] = v[0][3]; }
      else if constexpr (V == 2) xs[1] = v[0][1];
      v4_seg_step<1, NV>(key, xs);
      v4_seg_step<2, NV>(key, xs);
      v4_seg_step<4, NV>(key, xs);
      if (Tcur.deep) v4_seg_step<8, NV>(key, xs);  // only tiles with a node of more than 8 incident elements
      v[0][0] = xs[0];
      if constexpr (V == 4) { v[0][1] = xs[1]; v[0][2] = xs[1]; v[0][3] = xs[2]; }
      else if constexpr (V == 2) v[0][1] = xs[1];
      add_diag = __builtin_amdgcn_update_dpp(-1, key, 0x101, 0xf, 0xf, false) != key;  // row_shl:1: the next lane's node
    }
#pragma unroll
    for (int b = 0; b < P; ++b) {
      double* dst = row + rec_pos<REC, P>(Cuse.w, b);
#ifdef PF_EXPERIMENTS
      if (abl & 1) { if (v[b][0] == 1.2345e300) dst[0] = v[b][V > 1 ? 1 : 0] + v[b][V - 1]; continue; }  // keeps v alive, adds nothing
#endif
      if (b > 0 || add_diag) {
#pragma unroll
        for (int k = 0; k < V; ++k) {
          __hip_atomic_fetch_add(dst + k * plane, v[b][k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        }
      }
    }
  }

  // the NEXT tile's densities into the registers this tile's quadrature has just released (unconditional, like every
  // prefetch of this kernel; lanes without a pair read in-range nodes)
  if constexpr (RHO) v4_load_rho<P, REC>(Bnext, Rrho, rho_e);

  // copy-out: one buffer resource that covers exactly the tile's segment; pieces past its end are dropped by the
  // hardware, so the store COUNT is static (NOUT > 0: hipcc can then count what is in flight behind a prefetch and
  // keeps every s_waitcnt vmcnt partial) without clamps or branches; an empty tile (isolated nodes) stores nothing.
  const int32_t nblk = Tcur.nblk;
#ifdef PF_EXPERIMENTS
  if (abl & 8) return;
#endif
  if constexpr (V == 4) {
#ifdef PF_EXPERIMENTS
    const int64_t wb = g_abl_wrap_blocks;
    const pf_rsrc ro = make_rsrc(out0 + 4 * (wb ? (int64_t)Tcur.base % wb : (int64_t)Tcur.base), (uint32_t)nblk * 32u);
#else
    const pf_rsrc ro = make_rsrc(out0 + 4 * (int64_t)Tcur.base, (uint32_t)nblk * 32u);
#endif
    // piece q = lane + 64 u = values (2 (q & 1), 2 (q & 1) + 1) of block q >> 1:  (q >> 1) = (lane >> 1) + 32 u
    const double* a0 = img + (lane & 1) * 2 * plane + (lane >> 1);
    const double* a1 = a0 + plane;
    if constexpr (NOUT > 0) {
#pragma unroll
      for (int u = 0; u < NOUT; ++u) {
        const pf_f64x2 val = {a0[32 * u], a1[32 * u]};
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pf_u32x4, val), ro, lane * 16 + 1024 * u, 0, PF_AB_STORE_AUX);
      }
    } else {
      for (int u = 0; 32 * u < nblk; ++u) {
        const pf_f64x2 val = {a0[32 * u], a1[32 * u]};
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pf_u32x4, val), ro, lane * 16 + 1024 * u, 0, PF_AB_STORE_AUX);
      }
    }
  } else {
    // scalar matrices: 16 bytes (two consecutive entries) per lane and store, like the 2x2 path -- 1 KiB per wave store
    // instead of 512 B.  The resource of the wide stores ends at the last EVEN entry, so a piece that would straddle the
    // segment's end starts exactly at that end and is dropped whole; the odd last entry goes out in one 8-byte store.
    const uint32_t even_bytes = (uint32_t)(nblk & ~1) * 8u, all_bytes = (uint32_t)nblk * 8u;
    const pf_rsrc r0 = make_rsrc(out0 ? out0 + Tcur.base : nullptr, out0 ? even_bytes : 0u);
    const pf_rsrc r1 = make_rsrc((V == 2 && out1) ? out1 + Tcur.base : nullptr, (V == 2 && out1) ? even_bytes : 0u);
    const pf_rsrc t0 = make_rsrc(out0 ? out0 + Tcur.base : nullptr, out0 ? all_bytes : 0u);
    const pf_rsrc t1 = make_rsrc((V == 2 && out1) ? out1 + Tcur.base : nullptr, (V == 2 && out1) ? all_bytes : 0u);
    const double* a0 = img + 2 * lane;
    if constexpr (NOUT > 0) {
#pragma unroll
      for (int u = 0; u < NOUT; ++u) {
        const pf_f64x2 v0 = {a0[128 * u], a0[128 * u + 1]};
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pf_u32x4, v0), r0, lane * 16 + 1024 * u, 0, PF_AB_STORE_AUX);
        if constexpr (V == 2) {
          const pf_f64x2 v1 = {a0[plane + 128 * u], a0[plane + 128 * u + 1]};
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pf_u32x4, v1), r1, lane * 16 + 1024 * u, 0, PF_AB_STORE_AUX);
        }
      }
    } else {
      for (int u = 0; 128 * u < nblk; ++u) {
        const pf_f64x2 v0 = {a0[128 * u], a0[128 * u + 1]};
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pf_u32x4, v0), r0, lane * 16 + 1024 * u, 0, PF_AB_STORE_AUX);
        if constexpr (V == 2) {
          const pf_f64x2 v1 = {a0[plane + 128 * u], a0[plane + 128 * u + 1]};
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pf_u32x4, v1), r1, lane * 16 + 1024 * u, 0, PF_AB_STORE_AUX);
        }
      }
    }
    // the odd tail (lane 0 only: every other lane's offset lies past the resource; an even segment has no tail and
    // lane 0's offset nblk * 8 is past it too)
    const uint32_t toff = (nblk & 1) ? (uint32_t)(nblk - 1) * 8u + (uint32_t)lane * 0x100000u : all_bytes;
    const int32_t ti = (nblk & 1) ? nblk - 1 : 0;
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(pf_u32x2, img[ti]), t0, (int)toff, 0, PF_AB_STORE_AUX);
    if constexpr (V == 2)
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(pf_u32x2, img[plane + ti]), t1, (int)toff, 0, PF_AB_STORE_AUX);
  }
}

// COMB (pre-reduction of the diagonal block across the lanes of a node): on for triangles, where it took 0.191 ms back
// to 0.156 ms on 4 M elements, and its cheaper form (v4_seg_step) to 0.14 ms; quads are HBM-bound and measured the same
// with and without it (profiles/r02_assembly_ab.md)
#ifndef PF_AB_TILE_ORDER
#define PF_AB_TILE_ORDER ((P == 3 || ctx->tiles_sorted) ? 2 : 1)  // 2 XCD-aware round-robin (triangles: -20 %; Morton-sorted tiles), 1 plain round-robin (quads: -4 %), 0 contiguous run per wave
#endif
#ifndef PF_AB_COMB_MINV
#define PF_AB_COMB_MINV 2  // ... and for the two-plane (K, R) Helmholtz quads as well (4: 2x2 blocks only, the r02 state)
#endif
#ifndef PF_AB_QUAD_COMB
#define PF_AB_QUAD_COMB 1  // the diagonal pre-reduction for the 2x2-block quad kernels too (tools/build_variant.sh A/B: 0)
#endif
template <int P, class Phys, bool RHO, int NOUT, int REC, bool COMB = (P == 3 || (PF_AB_QUAD_COMB && Phys::V >= PF_AB_COMB_MINV))>
__device__ __forceinline__ void v4_kernel_body(int64_t nnodes, int64_t ntiles, int32_t plane, const double* __restrict__ X,
                                               const int32_t* __restrict__ nptr, const int4* __restrict__ tile_desc,
                                               const int4* __restrict__ vis_rec, int64_t nvis, int32_t strided,
                                               const typename Phys::Params& pr, double* __restrict__ out0,
                                               double* __restrict__ out1) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double* const img = reinterpret_cast<double*>(smem);  // V planes of `plane` doubles (odd: the planes start on different banks)
  const int lane = threadIdx.x;
  // tile walk: contiguous run per wave (tstride == 1) or round-robin over the grid (tstride == gridDim)
  // strided == 2: XCD-aware round-robin.  Workgroups are dealt to the 8 XCDs round-robin (b % 8), so with
  // the plain strided walk neighbouring tiles always sit on different XCDs.  Renumber the blocks so that one XCD's
  // blocks take a contiguous eighth of every grid-wide row of tiles.
  const int64_t tstride = strided ? gridDim.x : 1;
  const int64_t vblk = (strided == 2) ? (int64_t)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3) : blockIdx.x;
  const int64_t t0 = strided ? vblk : ntiles * blockIdx.x / gridDim.x;
  const int64_t t1 = strided ? ntiles : ntiles * (blockIdx.x + 1) / gridDim.x;
  if (t0 >= t1) return;
  const int64_t n = strided ? (t1 - t0 + tstride - 1) / tstride : t1 - t0;  // tiles of this wave
  // resources of the gathered arrays (the host checked that nnodes * 16 fits 32 bits)
  const pf_rsrc RX = make_rsrc(X, (uint32_t)nnodes * 16u);
  const pf_rsrc Rrho = make_rsrc(pr.rho, pr.rho ? (uint32_t)nnodes * 8u : 0u);

  // Pipeline state.  Two register sets (a/b) are used on alternate tiles, so nothing a load is about to
  // write is ever copied: a rotation by copy would have to wait for the loads just issued.
  V4Tile T0 = v4_walk_tile(0, n, t0, tstride, tile_desc);
  V4Tile T1 = v4_walk_tile(1, n, t0, tstride, tile_desc);
  V4Tile T2 = v4_walk_tile(2, n, t0, tstride, tile_desc);
  V4Rec Ba = v4_load_rec<REC>(T0, lane, vis_rec, nptr);
  V4Rec Bb = v4_load_rec<REC>(T1, lane, vis_rec, nptr);
  V4Visit<P, RHO> Ca, Cb;
  double rho_e[P];
#pragma unroll
  for (int b = 0; b < P; ++b) rho_e[b] = 1.0;
  v4_load_visit<P, RHO, REC>(Ba, RX, Rrho, Ca);
  if constexpr (RHO) v4_load_rho<P, REC>(Ba, Rrho, rho_e);
  // Make the prologue's loads land here.  Otherwise the loop header merges "Bb's record is the youngest
  // load" (prologue: needs vmcnt(0)) with "NOUT stores are younger than it" (back edge: a partial count is
  // enough) into one conservative vmcnt(0) that the steady state then pays on every trip.
  asm volatile("" ::"v"(Bb.rec.x), "v"(Bb.rec.y), "v"(Bb.rec.z), "v"(Bb.rec.w), "v"(Bb.rp));
  asm volatile("" ::"v"(Ca.el.x[0]), "v"(Ca.el.x[1]), "v"(Ca.el.x[2]), "v"(Ca.el.y[0]), "v"(Ca.el.y[1]), "v"(Ca.el.y[2]));
  if constexpr (P == 4) asm volatile("" ::"v"(Ca.el.x[3]), "v"(Ca.el.y[3]));
  if constexpr (RHO) asm volatile("" ::"v"(rho_e[0]), "v"(rho_e[1]), "v"(rho_e[2]), "v"(rho_e[P - 1]));

  // steady state: two tiles per trip, both always executed (a conditional second half gives the
  // back edge two different in-flight counts and hipcc falls back to vmcnt(0) at the loop top)
  int64_t j = 0;
  for (; j + 1 < n; j += 2) {
    // tile j: integrate Ca; gathers of tile j+1 -> Cb (record Bb arrived last step); record of j+2 -> Ba
    const V4Tile T3 = v4_walk_tile(j + 3, n, t0, tstride, tile_desc);
    v4_load_visit<P, RHO, REC>(Bb, RX, Rrho, Cb);
    Ba = v4_load_rec<REC>(T2, lane, vis_rec, nptr);
    v4_tile<P, Phys, RHO, NOUT, COMB, REC>(pr, T0, Ca, img, plane, lane, out0, out1, rho_e, Bb, Rrho);  // densities of j+1 from Bb
    // tile j+1: integrate Cb; gathers of tile j+2 -> Ca; record of j+3 -> Bb
    const V4Tile T4 = v4_walk_tile(j + 4, n, t0, tstride, tile_desc);
    v4_load_visit<P, RHO, REC>(Ba, RX, Rrho, Ca);
    Bb = v4_load_rec<REC>(T3, lane, vis_rec, nptr);
    v4_tile<P, Phys, RHO, NOUT, COMB, REC>(pr, T1, Cb, img, plane, lane, out0, out1, rho_e, Ba, Rrho);  // densities of j+2 from Ba
    T0 = T2; T1 = T3; T2 = T4;
  }
  if (j < n) v4_tile<P, Phys, RHO, NOUT, COMB, REC>(pr, T0, Ca, img, plane, lane, out0, out1, rho_e, Bb, Rrho);  // odd tail
}

template <int P, class Phys, bool RHO, int NOUT, int REC>
__global__ __launch_bounds__(64) void k_assemble_v4(int64_t nnodes, int64_t ntiles, int32_t plane, const double* __restrict__ X,
                                                    const int32_t* __restrict__ nptr, const int4* __restrict__ tile_desc,
                                                    const int4* __restrict__ vis_rec, int64_t nvis, int32_t strided,
                                                    typename Phys::Params pr, double* __restrict__ out0,
                                                    double* __restrict__ out1) {
  v4_kernel_body<P, Phys, RHO, NOUT, REC>(nnodes, ntiles, plane, X, nptr, tile_desc, vis_rec, nvis, strided, pr, out0, out1);
}

// ------------------------------------------------------------------------------------------
// Sensitivity pass:  out[o] = d(phi^T K(rho) psi)/d rho_o        (pyfem.py:1873-1920, 1948-1973)
// The reference materialises dKe/drho as an (E, D, D, P) tensor (2 KB per quad) and contracts it
// with phi, psi.  Here: inner[e,o] = sum_q detJ w N_o(q) r'(rho_q) (B phi_e)^T C0 (B psi_e), with
// r' = (1+p)/(1+p(1-rho_q))^2, evaluated by the owner of node o and summed over its incident
// elements in ascending order in a register: one lane per node, no scatter, no atomics, no tensor.
// ------------------------------------------------------------------------------------------
template <int P, bool SCALAR = false>
__global__ __launch_bounds__(128) void k_sens_elasticity(int64_t nnodes, const double* __restrict__ X,
                                                         const int32_t* __restrict__ n2e_ptr,
                                                         const int4* __restrict__ vis_rec, ElasticityParams pr,
                                                         const double* __restrict__ phi,
                                                         const double* __restrict__ psi, double* __restrict__ out) {
  const int64_t n = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (n >= nnodes) return;
  double acc = 0.0;
  for (int32_t jj = n2e_ptr[n]; jj < n2e_ptr[n + 1]; ++jj) {
    const int4 rec = vis_rec[jj];  // rotated element: n is its local node 0 (pf_pattern.hip k_build_visits)
    const int32_t nd[4] = {(int32_t)n, rec.x, rec.y, rec.z};
    Elem<P> el;
    double rho_e[P], fx[P], fy[P], gx[P], gy[P];
#pragma unroll
    for (int b = 0; b < P; ++b) {
      const double2 xy = *reinterpret_cast<const double2*>(X + 2 * (int64_t)nd[b]);
      el.x[b] = xy.x;
      el.y[b] = xy.y;
      rho_e[b] = pr.rho[nd[b]];
      if constexpr (SCALAR) {  // LinearPoisson: scalar fields (pyfem.py:1240-1276)
        fx[b] = phi[nd[b]]; gx[b] = psi[nd[b]]; fy[b] = 0.0; gy[b] = 0.0;
      } else {
        const double2 f = *reinterpret_cast<const double2*>(phi + 2 * (int64_t)nd[b]);
        const double2 g = *reinterpret_cast<const double2*>(psi + 2 * (int64_t)nd[b]);
        fx[b] = f.x; fy[b] = f.y; gx[b] = g.x; gy[b] = g.y;
      }
    }
    Own own = make_own<P>(0);
    auto point = [&](double det, double w, const double* N, const double* Nx, const double* Ny) {
      double rq = 0.0, e1 = 0.0, e2 = 0.0, e3 = 0.0, h1 = 0.0, h2 = 0.0, h3 = 0.0;
#pragma unroll
      for (int b = 0; b < P; ++b) {
        rq = fma(N[b], rho_e[b], rq);
        e1 = fma(Nx[b], fx[b], e1);                       // strains of phi: B phi_e  (pyfem.py:1992-1997)
        e2 = fma(Ny[b], fy[b], e2);
        e3 = fma(Ny[b], fx[b], fma(Nx[b], fy[b], e3));
        h1 = fma(Nx[b], gx[b], h1);                       // strains of psi
        h2 = fma(Ny[b], gy[b], h2);
        h3 = fma(Ny[b], gx[b], fma(Nx[b], gy[b], h3));
      }
      const double t = fma(pr.p, 1.0 - rq, 1.0);
      const double dr = (1.0 + pr.p) / (t * t);           // RAMP derivative, pyfem.py:1968
      double eCe;
      if constexpr (SCALAR) {
        double px = 0.0, py = 0.0, qx = 0.0, qy = 0.0;  // grad phi . grad psi
#pragma unroll
        for (int b = 0; b < P; ++b) {
          px = fma(Nx[b], fx[b], px); py = fma(Ny[b], fx[b], py);
          qx = fma(Nx[b], gx[b], qx); qy = fma(Ny[b], gx[b], qy);
        }
        eCe = px * qx + py * qy;
      } else {
        eCe = pr.c11 * (e1 * h1 + e2 * h2) + pr.c12 * (e1 * h2 + e2 * h1) + pr.c33 * e3 * h3;
      }
      acc = fma(det * w * dr * own.Na, eCe, acc);
    };
    double det, Nx[P], Ny[P], N[P];
    if constexpr (P == 4) {
      const QuadEdges ed = quad_edges(el);
      quad_point<0>(ed, own, det, Nx, Ny, N); point(det, 1.0, N, Nx, Ny);
      quad_point<1>(ed, own, det, Nx, Ny, N); point(det, 1.0, N, Nx, Ny);
      quad_point<2>(ed, own, det, Nx, Ny, N); point(det, 1.0, N, Nx, Ny);
      quad_point<3>(ed, own, det, Nx, Ny, N); point(det, 1.0, N, Nx, Ny);
    } else {
      tri_point(el, own, det, Nx, Ny, N);
      point(det, 0.5, N, Nx, Ny);
    }
  }
  out[n] = acc;
}

// ------------------------------------------------------------------------------------------
// LinearPoisson right-hand side (pyfem.py:997-1004, 1131-1173, 861-875).
//   k_quad_points:  Xq[e,q,:] = sum_l N[q,l] X[conn[e,l],:]   (the points the Python gfunc is evaluated at)
//   k_poisson_rhs:  rhs[n] = sum over incident (e,a) of rhs_e[e,a] with
//                   rhs_e[e,a] = sum_q detJ_q w_q N[q][a] g[e,q]       -- quads (N symmetric: pyfem.py:1132)
//                   rhs_e[e,0] = detJ * 0.5 * g[e,0], other columns dropped -- triangles: the reference's
//                   _assemble_rhs loops over range(nquads) columns (pyfem.py:873), reproduced.
// ------------------------------------------------------------------------------------------
template <int P>
__global__ void k_quad_points(int64_t nelems, const double* __restrict__ X, const int32_t* __restrict__ conn,
                              double* __restrict__ Xq) {
  const int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (e >= nelems) return;
  int32_t nd[P];
  load_conn<P>(conn, e, nd);
  Elem<P> el;
  load_elem<P>(X, nd, el);
  constexpr int Q = (P == 4) ? 4 : 1;
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    double N[P];
    if constexpr (P == 4) {
      const double xi = (q == 0 || q == 3) ? -GP : GP, et = (q == 0 || q == 1) ? -GP : GP;
      N[0] = 0.25 * (1.0 - xi) * (1.0 - et); N[1] = 0.25 * (1.0 + xi) * (1.0 - et);
      N[2] = 0.25 * (1.0 + xi) * (1.0 + et); N[3] = 0.25 * (1.0 - xi) * (1.0 + et);
    } else {
      N[0] = 1.0 / 3; N[1] = 1.0 / 3; N[2] = 1 - 1.0 / 3 - 1.0 / 3;
    }
    double xq = 0.0, yq = 0.0;
#pragma unroll
    for (int l = 0; l < P; ++l) {
      xq += N[l] * el.x[l];
      yq += N[l] * el.y[l];
    }
    Xq[(e * Q + q) * 2 + 0] = xq;
    Xq[(e * Q + q) * 2 + 1] = yq;
  }
}

template <int P>
__global__ __launch_bounds__(128) void k_poisson_rhs(int64_t nnodes, const double* __restrict__ X,
                                                     const int32_t* __restrict__ conn,
                                                     const int32_t* __restrict__ n2e_ptr,
                                                     const int32_t* __restrict__ n2e, const double* __restrict__ gq,
                                                     double* __restrict__ rhs) {
  const int64_t n = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (n >= nnodes) return;
  double acc = 0.0;
  for (int32_t jj = n2e_ptr[n]; jj < n2e_ptr[n + 1]; ++jj) {
    const int32_t ea = n2e[jj];
    const int32_t e = ea / P;
    const int a = ea - e * P;
    int32_t nd[P];
    load_conn<P>(conn, e, nd);
    Elem<P> el;
    load_elem<P>(X, nd, el);
    Own own = make_own<P>(a);
    double det, Nx[P], Ny[P], N[P];
    if constexpr (P == 4) {
      const QuadEdges ed = quad_edges(el);
      quad_point<0>(ed, own, det, Nx, Ny, N); acc = fma(det * own.Na, gq[4 * (int64_t)e + 0], acc);
      quad_point<1>(ed, own, det, Nx, Ny, N); acc = fma(det * own.Na, gq[4 * (int64_t)e + 1], acc);
      quad_point<2>(ed, own, det, Nx, Ny, N); acc = fma(det * own.Na, gq[4 * (int64_t)e + 2], acc);
      quad_point<3>(ed, own, det, Nx, Ny, N); acc = fma(det * own.Na, gq[4 * (int64_t)e + 3], acc);
    } else {
      tri_point(el, own, det, Nx, Ny, N);
      if (a == 0) acc = fma(det * 0.5, gq[e], acc);
    }
  }
  rhs[n] = acc;
}

// ------------------------------------------------------------------------------------------
// ATOMIC scatter (elasticity only): one lane per element
// ------------------------------------------------------------------------------------------
template <int P, bool RHO>
__global__ __launch_bounds__(256) void k_assemble_atomic(int64_t nelems, const double* __restrict__ X,
                                                         const int32_t* __restrict__ conn,
                                                         const int32_t* __restrict__ nptr,
                                                         const uint8_t* __restrict__ e2pos, ElasticityParams pr,
                                                         double* __restrict__ out) {
  const int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (e >= nelems) return;
  int32_t nd[P];
  load_conn<P>(conn, e, nd);
#pragma unroll
  for (int a = 0; a < P; ++a) {
    double v[P][4];
    element_rows<P, Elasticity<P, RHO>, RHO>(pr, X, nd, a, v);
    const int64_t rb = nptr[nd[a]];
#pragma unroll
    for (int b = 0; b < P; ++b) {
      double* dst = out + 4 * (rb + e2pos[(e * P + a) * P + b]);
#pragma unroll
      for (int k = 0; k < 4; ++k) unsafeAtomicAdd(dst + k, v[b][k]);
    }
  }
}

// element matrices to memory, (E, P*d, P*d) row-major
template <int P, class Phys, bool RHO>
__global__ __launch_bounds__(256) void k_element_matrices(int64_t nelems, const double* __restrict__ X,
                                                          const int32_t* __restrict__ conn,
                                                          typename Phys::Params pr, double* __restrict__ out0,
                                                          double* __restrict__ out1) {
  const int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (e >= nelems) return;
  int32_t nd[P];
  load_conn<P>(conn, e, nd);
#pragma unroll
  for (int a = 0; a < P; ++a) {
    double v[P][Phys::V];
    element_rows<P, Phys, RHO>(pr, X, nd, a, v);
    if constexpr (Phys::V == 4) {
      constexpr int D = 2 * P;
      double* base = out0 + e * D * D;
#pragma unroll
      for (int b = 0; b < P; ++b) {
        base[(2 * a) * D + 2 * b] = v[b][0];
        base[(2 * a) * D + 2 * b + 1] = v[b][1];
        base[(2 * a + 1) * D + 2 * b] = v[b][2];
        base[(2 * a + 1) * D + 2 * b + 1] = v[b][3];
      }
    } else {
#pragma unroll
      for (int b = 0; b < P; ++b) {
        if (out0) out0[e * P * P + a * P + b] = v[b][0];
        if constexpr (Phys::V == 2)
          if (out1) out1[e * P * P + a * P + b] = v[b][1];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
static inline ElasticityParams make_elasticity(double E, double nu, double p, const double* rho) {
  ElasticityParams pr;
  const double f = E * (1.0 / (1.0 - nu * nu));  // same operation order as pyfem.py:1747-1750
  pr.c11 = f;
  pr.c12 = (E * nu) * (1.0 / (1.0 - nu * nu));
  pr.c33 = (E * (0.5 * (1.0 - nu))) * (1.0 / (1.0 - nu * nu));
  pr.p = p;
  pr.rho = rho;
  return pr;
}

// copy-out variant for `per_lane` 16-byte pieces per lane: a static store count (3, 5, 10) or the counted loop (0)
template <int P, class Phys, bool RHO, int REC>
static auto v4_variant(int per_lane, int& nout) -> decltype(&k_assemble_v4<P, Phys, RHO, 0, REC>) {
  // quads with a density field: the counted loop.  The static store count costs that kernel six registers (134 against
  // 128: 12 against 16 resident waves per CU); with the loop it runs 0.302 -> 0.296 ms on 4 M quads (r03, tools/asm_sweep.py).
  // The sixteen waves are NOT the 0.64 of peak VERDICT r2 hoped for: the kernel's extra 50 us over the plain one are the
  // density gathers and the fp64 divisions of RAMP at every quadrature point, not occupancy.
  if (RHO && P == 4) { nout = per_lane; return k_assemble_v4<P, Phys, RHO, 0, REC>; }
  if (per_lane <= 3) { nout = 3; return k_assemble_v4<P, Phys, RHO, 3, REC>; }
  if (per_lane <= 5) { nout = 5; return k_assemble_v4<P, Phys, RHO, 5, REC>; }
  if (per_lane <= 10) { nout = 10; return k_assemble_v4<P, Phys, RHO, 10, REC>; }
  nout = per_lane;
  return k_assemble_v4<P, Phys, RHO, 0, REC>;
}

// Launch of the fused assembly kernel.  Experiment knobs (builds with -DPF_EXPERIMENTS only; the product library never
// reads them): PF_GATHER_WAVES_PER_CU, PF_TILE_ORDER=c|r (default: XCD-aware round-robin).
template <int P, class Phys, bool RHO>
static int launch_gather(pf_ctx* ctx, typename Phys::Params pr, double* out0, double* out1) {
  if (ctx->npt == 0 || !ctx->vis_rec || !ctx->tile_desc)
    return pf_fail(ctx, PF_E_UNSUPPORTED, "assemble: this pattern has no assembly tiles (node valence above 64)");
  if (ctx->nnodes >= ((int64_t)1 << 28))
    return pf_fail(ctx, PF_E_UNSUPPORTED, "assemble: more than 2^28 nodes do not fit the 32-bit buffer offsets of the gathers");
  // stores per lane of the copy-out: from the largest tile the greedy pass actually made (tiles stop at 64 incident
  // pairs, so on triangles they hold 10 of the npt = 16 nodes the valence rule allows: 3 stores, not 5)
  const int64_t tile_blocks = ctx->max_tile_blocks > 0 ? ctx->max_tile_blocks : (int64_t)ctx->npt * ctx->max_row_blocks;
  const int per_lane = Phys::V == 4 ? (int)cdiv(tile_blocks * 2, 64) : (int)cdiv(tile_blocks, 128);  // 16-byte pieces
  // compact records whenever the symbolic phase built them (quads: r02 A/B on 4M elements -- triangles ran 2.5 % slower
  // with them, the quad Helmholtz kernel 2.6 % faster, quad elasticity the same with 8 % less traffic)
  // compact records whenever the symbolic phase built them: 8 bytes when the numbering allows it, else 12 (triangles) or
  // 16 (quads).  Both kernels are bound by the memory system (r02: quads 8 % less traffic, the same time; triangles
  // after the cheaper pre-reduction: see profiles/r02_assembly_ab.md)
  constexpr int RECF = (P == 3) ? 12 : 16;  // fallback record
  const void* recs = ctx->vis_rec8 ? ctx->vis_rec8 : (P == 3 && ctx->vis_rec12) ? ctx->vis_rec12 : ctx->vis_rec;
  const int rec = ctx->vis_rec8 ? 8 : (P == 3 && ctx->vis_rec12) ? 12 : 16;
  int nout = 0;
  decltype(&k_assemble_v4<P, Phys, RHO, 0, 8>) k4 = nullptr;
  if (rec == 8) k4 = v4_variant<P, Phys, RHO, 8>(per_lane, nout);
  else if (rec == RECF) k4 = v4_variant<P, Phys, RHO, RECF>(per_lane, nout);
#if defined(PF_EXPERIMENTS) || defined(PF_AB_TRI_REC)  // triangles on the 16-byte records: A/B builds only
  else if constexpr (P == 3) k4 = v4_variant<P, Phys, RHO, 16>(per_lane, nout);
#endif
  if (!k4) return pf_fail(ctx, PF_E_UNSUPPORTED, "assemble: no kernel for this record layout");
  // image + the run-over of the copy-out's LDS reads (pieces past the segment end are read and then dropped)
  const size_t plane = (size_t)tile_blocks | 1;
  // (2x2 blocks: piece u of a lane reads 32 u doubles past its base; scalar matrices: 128 u.  Keep it tight: with 2.5 KB
  // more than needed the quad elasticity kernel dropped from 16 to 15 resident waves per CU and ran 25 % slower, r02.)
  const size_t lds4 = (((plane * Phys::V + 127) & ~(size_t)127) + (size_t)(Phys::V == 4 ? 32 : 128) * nout + 2) * 8;
  if (lds4 > 160 * 1024)
    return pf_fail(ctx, PF_E_UNSUPPORTED, "assemble: node rows too long for the LDS image of a tile");
  if (lds4 > 48 * 1024)
    PF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k4), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4));
#ifdef PF_EXPERIMENTS
  {
    const char* wm = pf_exp_env("PF_ABL_WRAP_MB");
    const int64_t wrap_blocks = (wm && atoi(wm) > 0) ? (int64_t)atoi(wm) * 1024 * 1024 / 32 : 0;
    PF_HIP(hipMemcpyToSymbolAsync(HIP_SYMBOL(g_abl_wrap_blocks), &wrap_blocks, sizeof(wrap_blocks), 0, hipMemcpyHostToDevice, ctx->stream));
    const char* am = pf_exp_env("PF_ABL_MASK");
    const int mask = am ? atoi(am) : 0;
    PF_HIP(hipMemcpyToSymbolAsync(HIP_SYMBOL(g_abl_mask), &mask, sizeof(mask), 0, hipMemcpyHostToDevice, ctx->stream));
  }
#endif
  const int64_t nt = ctx->ntiles;
  // persistent waves: exactly as many as are resident at once (registers and LDS decide: 16 per CU for the quad
  // elasticity kernel, 24-32 for the others).  More than that and the surplus waves only start when the first ones
  // have finished their share -- a second, half-empty round (r02: 24 launched / 16 resident cost the quad kernel a
  // quarter of its time); fewer and latency is not covered.  Capped at 32 per CU (r01 sweep: no gain beyond).
  static thread_local struct { size_t lds; int wpc; } occ[12] = {};  // per copy-out variant and record width
  auto& oc = occ[(nout == 3 ? 1 : nout == 5 ? 2 : nout == 10 ? 3 : 0) + (rec == 8 ? 4 : rec == 12 ? 8 : 0)];
  if (oc.wpc == 0 || oc.lds != lds4) {
    int q = 0;
    PF_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&q, reinterpret_cast<const void*>(k4), 64, lds4));
    oc.lds = lds4;
    oc.wpc = std::max(1, std::min(q, 32));
  }
  int wpc = oc.wpc;
  const char* g3 = pf_exp_env("PF_GATHER_WAVES_PER_CU");
  if (g3 && atoi(g3) > 0) wpc = atoi(g3);
  const int ncu = ctx->ncu > 0 ? ctx->ncu : 256;
  const unsigned grid = (unsigned)std::min<int64_t>(nt, (int64_t)ncu * wpc);
  // tile walk: 0 contiguous run per wave, 1 round-robin over the grid, 2 XCD-aware round-robin (grid % 8 == 0)
  const char* to = pf_exp_env("PF_TILE_ORDER");
  int32_t tile_order = (to && to[0] == 'c') ? 0 : (to && to[0] == 'r') ? 1 : PF_AB_TILE_ORDER;
  if (tile_order == 2 && (grid & 7)) tile_order = 1;
  hipLaunchKernelGGL(k4, dim3(grid), dim3(64), lds4, ctx->stream, ctx->nnodes, ctx->ntiles, (int32_t)plane,
                     ctx->X, ctx->nptr, reinterpret_cast<const int4*>(ctx->tile_desc),
                     reinterpret_cast<const int4*>(recs), (int64_t)(ctx->nelems * ctx->P), tile_order,
                     pr, out0, out1);
  PF_LAUNCH_CHECK();
  return PF_OK;
}

int pf_launch_assemble_elasticity(pf_ctx* ctx, double E, double nu, double p, const double* rho, int mode,
                                  double* Kvals) {
  ElasticityParams pr = make_elasticity(E, nu, p, rho);
  if (mode == PF_SCATTER_GATHER) {
    if (ctx->P == 4)
      return rho ? launch_gather<4, Elasticity<4, true>, true>(ctx, pr, Kvals, nullptr)
                 : launch_gather<4, Elasticity<4, false>, false>(ctx, pr, Kvals, nullptr);
    return rho ? launch_gather<3, Elasticity<3, true>, true>(ctx, pr, Kvals, nullptr)
               : launch_gather<3, Elasticity<3, false>, false>(ctx, pr, Kvals, nullptr);
  }
  int rc = pf_ensure_e2pos(ctx);
  if (rc) return rc;
  PF_HIP(hipMemsetAsync(Kvals, 0, sizeof(double) * 4 * ctx->nblocks, ctx->stream));
  const unsigned grid = (unsigned)cdiv(ctx->nelems, 256);
  if (ctx->P == 4) {
    if (rho) k_assemble_atomic<4, true><<<grid, 256, 0, ctx->stream>>>(ctx->nelems, ctx->X, ctx->conn, ctx->nptr, ctx->e2pos, pr, Kvals);
    else k_assemble_atomic<4, false><<<grid, 256, 0, ctx->stream>>>(ctx->nelems, ctx->X, ctx->conn, ctx->nptr, ctx->e2pos, pr, Kvals);
  } else {
    if (rho) k_assemble_atomic<3, true><<<grid, 256, 0, ctx->stream>>>(ctx->nelems, ctx->X, ctx->conn, ctx->nptr, ctx->e2pos, pr, Kvals);
    else k_assemble_atomic<3, false><<<grid, 256, 0, ctx->stream>>>(ctx->nelems, ctx->X, ctx->conn, ctx->nptr, ctx->e2pos, pr, Kvals);
  }
  PF_LAUNCH_CHECK();
  return PF_OK;
}

int pf_launch_assemble_helmholtz(pf_ctx* ctx, double r0, double* Kvals, double* Rvals) {
  HelmholtzParams pr{r0 * r0, nullptr};
  if (ctx->P == 4) return launch_gather<4, Helmholtz<4>, false>(ctx, pr, Kvals, Rvals);
  return launch_gather<3, Helmholtz<3>, false>(ctx, pr, Kvals, Rvals);
}

extern "C" int pf_assemble_elasticity2d(pf_ctx* ctx, double E, double nu, double p, const double* rho, int mode,
                                        double* Kvals) {
  if (!ctx) return pf_fail(nullptr, PF_E_INVALID, "ctx is NULL");
  PF_REQUIRE(ctx->ndims == 2, "this entry point is for 2-D meshes (pf_mesh_set); 3-D blocks have their own *3d entry points");
  PF_REQUIRE(ctx->d == 2, "pf_assemble_elasticity2d: pattern must be built with ndof_per_node = 2");
  PF_REQUIRE(Kvals, "pf_assemble_elasticity2d: Kvals is NULL");
  PF_REQUIRE(mode == PF_SCATTER_GATHER || mode == PF_SCATTER_ATOMIC, "pf_assemble_elasticity2d: unknown mode");
  return pf_launch_assemble_elasticity(ctx, E, nu, p, rho, mode, Kvals);
}

extern "C" int pf_assemble_helmholtz2d(pf_ctx* ctx, double r0, double* Kvals, double* Rvals) {
  if (!ctx) return pf_fail(nullptr, PF_E_INVALID, "ctx is NULL");
  PF_REQUIRE(ctx->ndims == 2, "this entry point is for 2-D meshes (pf_mesh_set); 3-D blocks have their own *3d entry points");
  PF_REQUIRE(ctx->d == 1, "pf_assemble_helmholtz2d: pattern must be built with ndof_per_node = 1");
  PF_REQUIRE(Kvals || Rvals, "pf_assemble_helmholtz2d: both outputs NULL");
  return pf_launch_assemble_helmholtz(ctx, r0, Kvals, Rvals);
}

extern "C" int pf_element_matrices_elasticity2d(pf_ctx* ctx, double E, double nu, double p, const double* rho,
                                                double* Ke_out) {
  if (!ctx) return pf_fail(nullptr, PF_E_INVALID, "ctx is NULL");
  PF_REQUIRE(ctx->ndims == 2, "this entry point is for 2-D meshes (pf_mesh_set); 3-D blocks have their own *3d entry points");
  PF_REQUIRE(ctx->conn && Ke_out, "pf_element_matrices_elasticity2d: no mesh / NULL output");
  ElasticityParams pr = make_elasticity(E, nu, p, rho);
  const unsigned grid = (unsigned)cdiv(ctx->nelems, 256);
  if (ctx->P == 4) {
    if (rho) k_element_matrices<4, Elasticity<4, true>, true><<<grid, 256, 0, ctx->stream>>>(ctx->nelems, ctx->X, ctx->conn, pr, Ke_out, nullptr);
    else k_element_matrices<4, Elasticity<4, false>, false><<<grid, 256, 0, ctx->stream>>>(ctx->nelems, ctx->X, ctx->conn, pr, Ke_out, nullptr);
  } else {
    if (rho) k_element_matrices<3, Elasticity<3, true>, true><<<grid, 256, 0, ctx->stream>>>(ctx->nelems, ctx->X, ctx->conn, pr, Ke_out, nullptr);
    else k_element_matrices<3, Elasticity<3, false>, false><<<grid, 256, 0, ctx->stream>>>(ctx->nelems, ctx->X, ctx->conn, pr, Ke_out, nullptr);
  }
  PF_LAUNCH_CHECK();
  return PF_OK;
}

extern "C" int pf_element_matrices_helmholtz2d(pf_ctx* ctx, double r0, double* Ke_out, double* Re_out) {
  if (!ctx) return pf_fail(nullptr, PF_E_INVALID, "ctx is NULL");
  PF_REQUIRE(ctx->ndims == 2, "this entry point is for 2-D meshes (pf_mesh_set); 3-D blocks have their own *3d entry points");
  PF_REQUIRE(ctx->conn && (Ke_out || Re_out), "pf_element_matrices_helmholtz2d: no mesh / NULL outputs");
  HelmholtzParams pr{r0 * r0, nullptr};
  const unsigned grid = (unsigned)cdiv(ctx->nelems, 256);
  if (ctx->P == 4)
    k_element_matrices<4, Helmholtz<4>, false><<<grid, 256, 0, ctx->stream>>>(ctx->nelems, ctx->X, ctx->conn, pr, Ke_out, Re_out);
  else
    k_element_matrices<3, Helmholtz<3>, false><<<grid, 256, 0, ctx->stream>>>(ctx->nelems, ctx->X, ctx->conn, pr, Ke_out, Re_out);
  PF_LAUNCH_CHECK();
  return PF_OK;
}

extern "C" int pf_sens_elasticity2d(pf_ctx* ctx, double E, double nu, double p, const double* rho, const double* phi,
                                    const double* psi, double* out) {
  if (!ctx) return pf_fail(nullptr, PF_E_INVALID, "ctx is NULL");
  PF_REQUIRE(ctx->ndims == 2, "this entry point is for 2-D meshes (pf_mesh_set); 3-D blocks have their own *3d entry points");
  PF_REQUIRE(ctx->d == 2, "pf_sens_elasticity2d: pattern must be built with ndof_per_node = 2");
  PF_REQUIRE(rho && phi && psi && out, "pf_sens_elasticity2d: NULL argument");
  PF_REQUIRE(ctx->vis_rec, "pf_sens_elasticity2d: node valence above 64 is not supported");
  ElasticityParams pr = make_elasticity(E, nu, p, rho);
  const unsigned grid = (unsigned)cdiv(ctx->nnodes, 128);
  if (ctx->P == 4)
    k_sens_elasticity<4><<<grid, 128, 0, ctx->stream>>>(ctx->nnodes, ctx->X, ctx->n2e_ptr, reinterpret_cast<const int4*>(ctx->vis_rec), pr, phi, psi, out);
  else
    k_sens_elasticity<3><<<grid, 128, 0, ctx->stream>>>(ctx->nnodes, ctx->X, ctx->n2e_ptr, reinterpret_cast<const int4*>(ctx->vis_rec), pr, phi, psi, out);
  PF_LAUNCH_CHECK();
  return PF_OK;
}

// ---- LinearPoisson (row f4, 2-D part) ---------------------------------------------------------------
extern "C" int pf_assemble_poisson2d(pf_ctx* ctx, double p, const double* rho, double* Kvals) {
  if (!ctx) return pf_fail(nullptr, PF_E_INVALID, "ctx is NULL");
  PF_REQUIRE(ctx->ndims == 2, "this entry point is for 2-D meshes (pf_mesh_set); 3-D blocks have their own *3d entry points");
  PF_REQUIRE(ctx->d == 1, "pf_assemble_poisson2d: pattern must be built with ndof_per_node = 1");
  PF_REQUIRE(Kvals, "pf_assemble_poisson2d: Kvals is NULL");
  PoissonParams pr{p, rho};
  if (ctx->P == 4)
    return rho ? launch_gather<4, Poisson<4, true>, true>(ctx, pr, Kvals, nullptr)
               : launch_gather<4, Poisson<4, false>, false>(ctx, pr, Kvals, nullptr);
  return rho ? launch_gather<3, Poisson<3, true>, true>(ctx, pr, Kvals, nullptr)
             : launch_gather<3, Poisson<3, false>, false>(ctx, pr, Kvals, nullptr);
}

extern "C" int pf_element_matrices_poisson2d(pf_ctx* ctx, double p, const double* rho, double* Ke_out) {
  if (!ctx) return pf_fail(nullptr, PF_E_INVALID, "ctx is NULL");
  PF_REQUIRE(ctx->ndims == 2, "this entry point is for 2-D meshes (pf_mesh_set); 3-D blocks have their own *3d entry points");
  PF_REQUIRE(ctx->conn && Ke_out, "pf_element_matrices_poisson2d: no mesh / NULL output");
  PoissonParams pr{p, rho};
  const unsigned grid = (unsigned)cdiv(ctx->nelems, 256);
  if (ctx->P == 4) {
    if (rho) k_element_matrices<4, Poisson<4, true>, true><<<grid, 256, 0, ctx->stream>>>(ctx->nelems, ctx->X, ctx->conn, pr, Ke_out, nullptr);
    else k_element_matrices<4, Poisson<4, false>, false><<<grid, 256, 0, ctx->stream>>>(ctx->nelems, ctx->X, ctx->conn, pr, Ke_out, nullptr);
  } else {
    if (rho) k_element_matrices<3, Poisson<3, true>, true><<<grid, 256, 0, ctx->stream>>>(ctx->nelems, ctx->X, ctx->conn, pr, Ke_out, nullptr);
    else k_element_matrices<3, Poisson<3, false>, false><<<grid, 256, 0, ctx->stream>>>(ctx->nelems, ctx->X, ctx->conn, pr, Ke_out, nullptr);
  }
  PF_LAUNCH_CHECK();
  return PF_OK;
}

extern "C" int pf_quad_points(pf_ctx* ctx, double* Xq_out) {
  if (!ctx) return pf_fail(nullptr, PF_E_INVALID, "ctx is NULL");
  PF_REQUIRE(ctx->ndims == 2, "this entry point is for 2-D meshes (pf_mesh_set); 3-D blocks have their own *3d entry points");
  PF_REQUIRE(ctx->conn && Xq_out, "pf_quad_points: no mesh / NULL output");
  const unsigned grid = (unsigned)cdiv(ctx->nelems, 256);
  if (ctx->P == 4) k_quad_points<4><<<grid, 256, 0, ctx->stream>>>(ctx->nelems, ctx->X, ctx->conn, Xq_out);
  else k_quad_points<3><<<grid, 256, 0, ctx->stream>>>(ctx->nelems, ctx->X, ctx->conn, Xq_out);
  PF_LAUNCH_CHECK();
  return PF_OK;
}

extern "C" int pf_poisson_rhs(pf_ctx* ctx, const double* gq, double* rhs_out) {
  if (!ctx) return pf_fail(nullptr, PF_E_INVALID, "ctx is NULL");
  PF_REQUIRE(ctx->ndims == 2, "this entry point is for 2-D meshes (pf_mesh_set); 3-D blocks have their own *3d entry points");
  PF_REQUIRE(ctx->d == 1 && gq && rhs_out, "pf_poisson_rhs: needs a scalar pattern / NULL argument");
  const unsigned grid = (unsigned)cdiv(ctx->nnodes, 128);
  if (ctx->P == 4) k_poisson_rhs<4><<<grid, 128, 0, ctx->stream>>>(ctx->nnodes, ctx->X, ctx->conn, ctx->n2e_ptr, ctx->n2e, gq, rhs_out);
  else k_poisson_rhs<3><<<grid, 128, 0, ctx->stream>>>(ctx->nnodes, ctx->X, ctx->conn, ctx->n2e_ptr, ctx->n2e, gq, rhs_out);
  PF_LAUNCH_CHECK();
  return PF_OK;
}

extern "C" int pf_sens_poisson2d(pf_ctx* ctx, double p, const double* rho, const double* phi, const double* psi,
                                 double* out) {
  if (!ctx) return pf_fail(nullptr, PF_E_INVALID, "ctx is NULL");
  PF_REQUIRE(ctx->ndims == 2, "this entry point is for 2-D meshes (pf_mesh_set); 3-D blocks have their own *3d entry points");
  PF_REQUIRE(ctx->d == 1, "pf_sens_poisson2d: pattern must be built with ndof_per_node = 1");
  PF_REQUIRE(rho && phi && psi && out, "pf_sens_poisson2d: NULL argument");
  PF_REQUIRE(ctx->vis_rec, "pf_sens_poisson2d: node valence above 64 is not supported");
  ElasticityParams pr{0.0, 0.0, 0.0, p, rho};
  const unsigned grid = (unsigned)cdiv(ctx->nnodes, 128);
  if (ctx->P == 4)
    k_sens_elasticity<4, true><<<grid, 128, 0, ctx->stream>>>(ctx->nnodes, ctx->X, ctx->n2e_ptr, reinterpret_cast<const int4*>(ctx->vis_rec), pr, phi, psi, out);
  else
    k_sens_elasticity<3, true><<<grid, 128, 0, ctx->stream>>>(ctx->nnodes, ctx->X, ctx->n2e_ptr, reinterpret_cast<const int4*>(ctx->vis_rec), pr, phi, psi, out);
  PF_LAUNCH_CHECK();
  return PF_OK;
}
