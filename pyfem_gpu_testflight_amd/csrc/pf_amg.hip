// Aggregation AMG V-cycle preconditioner for the PCG of pf_solve.hip (gfx950).
//
// Counterpart of the reference's `pyamg.smoothed_aggregation_solver(K).aspreconditioner()`
// (pyfem.py:2393-2400) and `pyamg.ruge_stuben_solver(K)` (pyfem.py:2098) -- SURVEY.md 8f row f1.
// pyamg is a host library; this is a from-scratch device implementation on the same data layout as
// the rest of the path (node graph + dense d x d blocks), not a port:
//   * aggregates depend on the node GRAPH only, so they are built once per mesh: the greedy root/neighbourhood
//     aggregation on the host (serial O(nnz), a few ms per million nodes), the patterns of the smoothed
//     prolongator, of its transpose and of the coarse operators on the device;
//   * per matrix, on the device: the prolongator P = (I - w D^-1 A) P_tent (smoothed aggregation), where the
//     tentative prolongator carries the near-null space: the constant for scalar problems, the three
//     rigid-body modes (two translations and the in-plane rotation about the aggregate centroid) for plane
//     elasticity -> 3 x 3 coarse blocks, the six rigid-body modes for 3-D elasticity -> 6 x 6 blocks; dofs whose
//     row is decoupled (Dirichlet rows after apply_dirichlet_bcs) are kept out of P and are solved exactly by
//     the smoother; Galerkin coarse operators A_c = P^T A P (one wave per coarse row, no global atomics);
//   * V(2,2) cycle with damped Jacobi, omega = 4/(3*1.1*lambda_max(D^-1 A)) with lambda_max from ten
//     device power iterations per level (invariant under the scaling of individual dofs, unlike a
//     Gershgorin row sum: translations and rotations have different units on the coarse levels), dense
//     inverse on the coarsest level (<= 160 dofs).
// The cycle is a fixed symmetric positive definite operator, so it is a valid CG preconditioner.
// PF_AMG_SA=0 selects plain aggregation (tentative P formed on the fly from the coordinates, coarse correction
// scaled by 2 because plain aggregation under-corrects): kept as the cheap-setup variant and for comparison.
#include <hipcub/hipcub.hpp>
#include <stdlib.h>

#include <algorithm>
#include <chrono>
#include <thread>
#include <vector>

#include "pf_common.h"

static constexpr int AMG_MAX_LEVELS = 12;
static constexpr int AMG_POW_PARTS = 4096;     // waves of one product launch at most (k_amg_spmv grids, SPMV_MAXGRID)
static constexpr int AMG_COARSEST_DOFS = 160;  // dense inverse: cost ~ m^3 on one workgroup per numeric setup

struct AmgLevel {
  int64_t n = 0;             // nodes
  int dr = 0;                // dofs per node on this level
  int32_t* ptr = nullptr;    // node graph (device); level 0 aliases the ctx pattern
  int32_t* idx = nullptr;
  bool owns_graph = false;
  int64_t nblk = 0;
  int max_row = 0;           // longest row of the node graph
  double* vals = nullptr;    // block values (level 0: the caller's matrix, not owned)
  float* vals32 = nullptr;   // fp32 copy of vals for the products of the cycle (owned; nullptr with PF_AMG_FP32=0)
  double* dinv = nullptr;    // omega_1 / diag per dof (raw diagonal during the numeric setup)
  double* dinv2 = nullptr;   // omega_2 / diag: second factor of the two-step smoother
  uint8_t* iso = nullptr;    // 1 = decoupled dof (row has no off-diagonal entries)
  double* xy = nullptr;      // node coordinates (n,dim) (level 0 aliases ctx->X)
  bool owns_xy = false;
  // transfer to the next level
  int64_t nc = 0;            // number of aggregates
  int32_t* agg = nullptr;    // (n) aggregate of each node
  int32_t* aptr = nullptr;   // (nc+1) aggregate -> members
  int32_t* amem = nullptr;   // (n)
  // smoothed prolongator P = (I - w D^-1 A) P_tent, stored: row n couples to the aggregates of n's neighbours
  int32_t* pptr = nullptr;   // (n+1)
  int32_t* pidx = nullptr;   // (np) aggregate ids, ascending per row
  double* pvals = nullptr;   // (np) blocks of dr x dc(next level), row-major
  int32_t* tptr = nullptr;   // (nc+1) transpose pattern: entries of each coarse column, ascending node
  int32_t* tnode = nullptr;  // (np) fine node of the entry
  int32_t* tslot = nullptr;  // (np) its index into pidx / pvals
  double* ptvals = nullptr;  // (np) the same blocks in transpose order (streamed by the restriction); PF_AMG_FP32=0 only
  float* pvals32 = nullptr;  // fp32 copies for the CYCLE's transfers (default): prolongation reads pvals32, restriction streams
  float* ptvals32 = nullptr; //   ptvals32 = its transpose copy -- the same rounded values, so the cycle stays symmetric.  The
                             //   Galerkin products and the distributed gathers read the fp64 pvals.
  int64_t np = 0;
  // two-step Galerkin product (A P first): pattern of A P, row n = the aggregates in the rows of P of n's neighbours
  int32_t* apptr = nullptr;  // (n+1)
  int32_t* apidx = nullptr;  // (nap) aggregate ids, ascending per row
  int32_t* aprow = nullptr;  // (nap) the row of every entry
  double* apvals = nullptr;  // (nap) blocks of dr x dc(next level)
  int64_t nap = 0;
  int32_t* aplptr = nullptr; // (nap + 1) plan of A P: for every entry (n, J) the (block of A, block of P) pairs it sums --
  int2* aplpair = nullptr;   //   which neighbour's row of P holds J, found once per pattern instead of once per matrix
  int32_t* gptr = nullptr;   // (next level's nblk + 1) gather plan of P^T (A P): the (P block, A P block) pairs of every coarse
  int2* gpair = nullptr;     //   entry (I, J >= I), in ascending fine-node order; entries below the diagonal have none
  int64_t ngpair = 0;
  int32_t max_crow = 0;      // longest row of the next level's graph
  bool sa = false;           // this level's transfer uses the stored smoothed P (else the tentative P on the fly)
  // work vectors
  double *x = nullptr, *b = nullptr, *t = nullptr;
  double omega = 0.6;
};

struct pf_amg {
  int nlevels = 0;
  int dim = 2;                  // stride of the coordinate arrays; 3 only for 3-D elasticity (scalar problems never
                                // read coordinates, so they run the dim = 2 code on any mesh)
  AmgLevel L[AMG_MAX_LEVELS];
  double* dense_inv = nullptr;  // coarsest level, (m x m) row-major
  // The inversion (one workgroup, ~0.47 ms for 114 dofs) runs on a SIDE stream: the numeric setup does not wait for it,
  // and the first cycle's work on the finer levels overlaps it; the stream that runs the cycle waits for ev_inv once,
  // right before the first dense apply.
  hipStream_t side = nullptr;
  hipEvent_t ev_coarse = nullptr, ev_inv = nullptr;
  bool inv_pending = false;
  int64_t dense_m = 0;
  double* gersh = nullptr;      // scratch vector for the power iteration
  double* powpart = nullptr;    // 2 x AMG_POW_PARTS per-wave partials of the last two iterates' norms
  double* lam = nullptr;        // device: per-level |D^-1 A v|^2, the not-SPD flag, the finest diagonal scale
  bool numeric_ok = false;      // the numeric setup below matches L[0].vals
  int sa_mode = 1;              // PF_AMG_SA: 0 tentative P on the fly everywhere (plain aggregation), 2 smoothed
                                // aggregation (stored P) on every level, 1 (default) smoothed except on the finest
                                // level of 3-D elasticity, where the triple product with 125-node columns of 3x6
                                // blocks costs more than the iterations it saves (96^3: 61 ms against 45 ms of solve)
  int nsmooth = 2;              // damped-Jacobi sweeps before and after the coarse correction
  double over = 2.0;            // scaling of the coarse-grid correction: plain aggregation under-corrects.
                                // 2048^2 plane stress, rtol 1e-8 (tools/amg_sweep.py): V(1,1)/1.0 180 its 572 ms,
                                // V(2,2)/2.0 60 its 231 ms, Jacobi-PCG 16 380 its 7.3 s.  Any value > 0 keeps M SPD.
  double cheb = 4.4;            // PF_AMG_CHEB: the two sweeps target [lambda_max / cheb, lambda_max]; <= 1: equal factors
  double over_sa = 1.0;         // the same for levels with a smoothed prolongator: no over-correction needed
  double omega_scale = 1.0;
  double psmooth = 1.0;         // scale of the prolongator-smoothing weight 4 / (3 * 1.1 lambda_max); PF_AMG_PSMOOTH (experiments)
  int pow_iters = 10;           // power-iteration steps of the lambda_max(D^-1 A) estimates; PF_AMG_POW (experiments)
  int nv_cap = 1;               // right-hand sides the level vectors are allocated for (amg_cycle grows them once)
  bool two_step = true;         // Galerkin products as P^T (A P) (k_sa_ap, k_sa_ptap); PF_AMG_TWOSTEP=0 (experiments): one sweep
};

// host-side graph logic (greedy aggregation, coarse graphs, host fallback of the prolongator patterns): pf_hostlogic.h --
// plain C++, so that tests/test_host_logic.py can build it with g++ -fsanitize=address,undefined and run it without a GPU
#include "pf_hostlogic.h"

// ------------------------------------------------------------------------------------------
// device kernels
// ------------------------------------------------------------------------------------------
// decoupled dofs + raw diagonal per dof; over the coupled dofs bad[0] is raised by a non-positive diagonal entry and
// scale[1] by a positive one: both raised = indefinite (no AMG), bad alone = negative definite (the cycle is
// sign-agnostic: D^-1 A, the Galerkin products and the coarsest inverse only change sign together)
// AMG_ROW_LANES lanes share a node: consecutive lanes read consecutive blocks of the row (one thread per row left the
// 600 MB sweep over the fine matrix latency-bound: 1.7 ms at 2 M nodes against 0.2 ms of traffic)
static constexpr int AMG_ROW_LANES = 8;
template <int D>
__global__ __launch_bounds__(256) void k_amg_rows(int64_t n, const int32_t* __restrict__ ptr, const int32_t* __restrict__ idx,
                                                  const double* __restrict__ vals, uint8_t* __restrict__ iso,
                                                  double* __restrict__ diag, double* __restrict__ bad,
                                                  double* __restrict__ scale, int finest, float* __restrict__ v32) {
  // v32 != NULL: the fp32 copy of the operator the cycle reads is written in the same pass (it was a pass of its own over
  // the 600 MB of the fine matrix, and one more launch on every coarse level)
  const int64_t i = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) / AMG_ROW_LANES;
  const int sub = threadIdx.x & (AMG_ROW_LANES - 1);
  const bool valid = i < n;  // the lanes of a group agree, and a group never straddles a wave
  double off[D], dg[D];
#pragma unroll
  for (int k = 0; k < D; ++k) { off[k] = 0.0; dg[k] = 0.0; }
  const int32_t gb = valid ? ptr[i] : 0, ge = valid ? ptr[i + 1] : 0;
  for (int32_t g = gb + sub; g < ge; g += AMG_ROW_LANES) {
    const int64_t m = idx[g];
    const double* a = vals + (int64_t)g * D * D;
    if (v32) {
#pragma unroll
      for (int q = 0; q < D * D; ++q) v32[(int64_t)g * D * D + q] = (float)a[q];
    }
#pragma unroll
    for (int k = 0; k < D; ++k)
#pragma unroll
      for (int l = 0; l < D; ++l) {
        if (m == i && k == l) dg[k] = a[k * D + l];
        else off[k] += fabs(a[k * D + l]);
      }
  }
#pragma unroll
  for (int k = 0; k < D; ++k)
#pragma unroll
    for (int o = 1; o < AMG_ROW_LANES; o <<= 1) {
      off[k] += __shfl_xor(off[k], o);
      dg[k] += __shfl_xor(dg[k], o);  // one lane holds the diagonal entry, the others 0
    }
  if (!valid || sub != 0) return;
  double rmax = 0.0;
#pragma unroll
  for (int k = 0; k < D; ++k) rmax = fmax(rmax, fabs(dg[k]) + off[k]);
#pragma unroll
  for (int k = 0; k < D; ++k) {
    // An empty coarse dof -- e.g. the rotation of a one-node aggregate, whose column of P is zero -- comes out of the
    // triple product as rounding noise of either sign: treat a row that is negligible against its node's other rows
    // as decoupled (identity), not as evidence of an indefinite matrix.
    const bool empty = fabs(dg[k]) + off[k] <= 1e-12 * (finest ? rmax : fmax(rmax, scale[0]));
    const bool decoupled = off[k] == 0.0 || empty;
    iso[D * i + k] = decoupled ? 1 : 0;
    if (dg[k] == 0.0 || empty) dg[k] = 1.0;
    if (!decoupled) {  // benign races: every writer stores the same value; read first -- millions of stores to one
      double* flag = dg[k] > 0.0 ? scale + 1 : bad;  // address serialise (11 ms at 2 M nodes)
      if (*reinterpret_cast<volatile double*>(flag) == 0.0) *flag = 1.0;
    }
    diag[D * i + k] = dg[k];
  }
  // the finest level leaves the size of its largest diagonal entry for the coarse levels' emptiness test
  // (positive doubles order like their bit patterns)
  // the value only grows, so a stale read can only cost a redundant atomic
  if (finest && rmax > *reinterpret_cast<volatile double*>(scale))
    atomicMax(reinterpret_cast<unsigned long long*>(scale), (unsigned long long)__double_as_longlong(rmax));
}

// Damping factors of the smoother from lam2[0] = |D^-1 A v|^2 of the last power iteration (b = 1.1 lambda_max):
//   cheb == 0: plain damped Jacobi, both factors 4 / (3 b);
//   cheb > 1:  the two sweeps of a V(2,2) side are the degree-2 Chebyshev polynomial on [b / cheb, b]: factors
//              1 / (c +- h cos(pi/4)), c = (a + b)/2, h = (b - a)/2.  Same cost as two equal sweeps; the worst
//              damping on that interval is 1/T_2((b+a)/(b-a)) (0.25 for cheb = 4.4) instead of 0.49.
__global__ void k_amg_scale_dinv(int64_t n, const double* __restrict__ lam2, double scale, double cheb,
                                 double* __restrict__ d, double* __restrict__ d2, const uint8_t* __restrict__ iso) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const double bb = 1.1 * sqrt(lam2[0]);
  double w1 = scale * 4.0 / (3.0 * bb), w2 = w1;
  if (cheb > 1.0) {
    const double aa = bb / cheb, c = 0.5 * (aa + bb), h = 0.5 * (bb - aa) * 0.70710678118654752;
    w1 = scale / (c + h);
    w2 = scale / (c - h);
  }
  if (i < n) {  // decoupled dofs are solved exactly
    const double dg = d[i];
    d[i] = (iso[i] ? 1.0 : w1) / dg;
    d2[i] = (iso[i] ? 1.0 : w2) / dg;
  }
}

// tentative prolongator row block P_n (DR x DC) for node n at offset r from its aggregate's centroid
template <int DIM, int DR, int DC>
__device__ __forceinline__ void amg_P(const double* __restrict__ xn, const double* __restrict__ xc, const uint8_t* iso_n,
                                      double Pn[DR][DC]) {
  if constexpr (DR == 1) {
    Pn[0][0] = iso_n[0] ? 0.0 : 1.0;
  } else if constexpr (DIM == 2) {
    const double dx = xn[0] - xc[0], dy = xn[1] - xc[1];
    // DR == 2: [I2 | rotation about the centroid];  DR == 3: rigid-body modes of a (ux, uy, theta) node
    Pn[0][0] = 1.0; Pn[0][1] = 0.0; Pn[0][2] = -dy;
    Pn[1][0] = 0.0; Pn[1][1] = 1.0; Pn[1][2] = dx;
    if constexpr (DR == 3) { Pn[2][0] = 0.0; Pn[2][1] = 0.0; Pn[2][2] = 1.0; }
#pragma unroll
    for (int k = 0; k < DR; ++k)
      if (iso_n[k]) { Pn[k][0] = Pn[k][1] = Pn[k][2] = 0.0; }
  } else {
    // 3-D: u = t + w x r.  DR == 3: [I3 | R(r)];  DR == 6 (t, w nodes): [[I3, R(r)], [0, I3]]
    const double rx = xn[0] - xc[0], ry = xn[1] - xc[1], rz = xn[2] - xc[2];
#pragma unroll
    for (int r = 0; r < DR; ++r)
#pragma unroll
      for (int c = 0; c < DC; ++c) Pn[r][c] = (r == c) ? 1.0 : 0.0;
    Pn[0][4] = rz;  Pn[0][5] = -ry;
    Pn[1][3] = -rz; Pn[1][5] = rx;
    Pn[2][3] = ry;  Pn[2][4] = -rx;
#pragma unroll
    for (int k = 0; k < DR; ++k)
      if (iso_n[k]) {
#pragma unroll
        for (int c = 0; c < DC; ++c) Pn[k][c] = 0.0;
      }
  }
}

// coarse coordinates = centroid of the members
__global__ void k_amg_centroids(int64_t nc, int dim, const int32_t* __restrict__ aptr, const int32_t* __restrict__ amem,
                                const double* __restrict__ xy, double* __restrict__ cxy) {
  const int64_t a = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (a >= nc) return;
  double s[3] = {0.0, 0.0, 0.0};
  const int32_t b = aptr[a], e = aptr[a + 1];
  for (int32_t k = b; k < e; ++k)
    for (int c = 0; c < dim; ++c) s[c] += xy[dim * (int64_t)amem[k] + c];
  const double inv = 1.0 / (double)max(e - b, 1);
  for (int c = 0; c < dim; ++c) cxy[dim * a + c] = s[c] * inv;
}

// Galerkin product, one lane per coarse row I:  C[I,J] = sum_{n in I} sum_{m in J} P_n^T A[n,m] P_m
template <int DIM, int DR, int DC>
__global__ void k_amg_galerkin(int64_t nc, const int32_t* __restrict__ aptr, const int32_t* __restrict__ amem,
                               const int32_t* __restrict__ agg, const int32_t* __restrict__ ptr,
                               const int32_t* __restrict__ idx, const double* __restrict__ vals,
                               const uint8_t* __restrict__ iso, const double* __restrict__ xy,
                               const double* __restrict__ cxy, const int32_t* __restrict__ cptr,
                               const int32_t* __restrict__ cidx, double* __restrict__ cvals) {
  const int64_t I = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (I >= nc) return;
  const int32_t cb = cptr[I], ce = cptr[I + 1];
  for (int64_t q = (int64_t)cb * DC * DC; q < (int64_t)ce * DC * DC; ++q) cvals[q] = 0.0;
  for (int32_t k = aptr[I]; k < aptr[I + 1]; ++k) {
    const int64_t n = amem[k];
    double Pn[DR][DC];
    amg_P<DIM, DR, DC>(xy + DIM * n, cxy + DIM * I, iso + DR * n, Pn);
    for (int32_t g = ptr[n]; g < ptr[n + 1]; ++g) {
      const int64_t m = idx[g];
      const int32_t J = agg[m];
      double Pm[DR][DC];
      amg_P<DIM, DR, DC>(xy + DIM * m, cxy + DIM * (int64_t)J, iso + DR * m, Pm);
      const double* a = vals + (int64_t)g * DR * DR;
      double AP[DR][DC];  // A[n,m] P_m
#pragma unroll
      for (int r = 0; r < DR; ++r)
#pragma unroll
        for (int c = 0; c < DC; ++c) {
          double s = 0.0;
#pragma unroll
          for (int t = 0; t < DR; ++t) s += a[r * DR + t] * Pm[t][c];
          AP[r][c] = s;
        }
      int32_t pos = cb;
      while (pos < ce - 1 && cidx[pos] != J) ++pos;  // J is in the row by construction; bounded anyway
      double* dst = cvals + (int64_t)pos * DC * DC;
#pragma unroll
      for (int r = 0; r < DC; ++r)
#pragma unroll
        for (int c = 0; c < DC; ++c) {
          double s = 0.0;
#pragma unroll
          for (int t = 0; t < DR; ++t) s += Pn[t][r] * AP[t][c];
          dst[r * DC + c] += s;
        }
    }
  }
}

// The same product with one WAVE per coarse row (rows of at most 64 couplings), upper triangle only
// (k_sa_mirror_lower fills the rest): two members of the aggregate per pass, 32 lanes over each member's row, the
// DC x DC contributions added into an LDS image of the coarse row.  The lane-per-row kernel above leaves the 36 k
// aggregates of 96^3 blocks on 560 waves with 27 x 27 block products each: 9.3 ms; this one takes 1.
template <int DIM, int DR, int DC>
__global__ __launch_bounds__(64) void k_amg_galerkin_wave(int64_t nc, const int32_t* __restrict__ aptr,
                                                          const int32_t* __restrict__ amem,
                                                          const int32_t* __restrict__ agg,
                                                          const int32_t* __restrict__ ptr,
                                                          const int32_t* __restrict__ idx,
                                                          const double* __restrict__ vals,
                                                          const uint8_t* __restrict__ iso,
                                                          const double* __restrict__ xy, const double* __restrict__ cxy,
                                                          const int32_t* __restrict__ cptr,
                                                          const int32_t* __restrict__ cidx, double* __restrict__ cvals) {
  __shared__ int32_t crow[64];
  __shared__ double cacc[64 * DC * DC];
  const int lane = threadIdx.x, half = lane >> 5, sub = lane & 31;
  for (int64_t I = blockIdx.x; I < nc; I += gridDim.x) {
    const int32_t cb = cptr[I], clen = cptr[I + 1] - cb;
    if (lane < clen) crow[lane] = cidx[cb + lane];
    for (int q = lane; q < clen * DC * DC; q += 64) cacc[q] = 0.0;
    __syncthreads();
    const int32_t kb = aptr[I], ke = aptr[I + 1];
    for (int32_t k0 = kb; k0 < ke; k0 += 2) {
      const int32_t k = k0 + half;
      const bool ok = k < ke;
      const int64_t n = ok ? amem[k] : 0;
      const int32_t rb = ok ? ptr[n] : 0, rl = ok ? ptr[n + 1] - rb : 0;
      double Pn[DR][DC];
      amg_P<DIM, DR, DC>(xy + DIM * n, cxy + DIM * I, iso + DR * n, Pn);
      const int32_t rlmax = max(__shfl(rl, 0), __shfl(rl, 32));
      for (int32_t g0 = 0; g0 < rlmax; g0 += 32) {
        const int32_t go = g0 + sub;
        if (!ok || go >= rl) continue;
        const int32_t g = rb + go;
        const int64_t m = idx[g];
        const int32_t J = agg[m];
        if (J < (int32_t)I) continue;  // symmetric: mirrored afterwards
        double Pm[DR][DC];
        amg_P<DIM, DR, DC>(xy + DIM * m, cxy + DIM * (int64_t)J, iso + DR * m, Pm);
        const double* a = vals + (int64_t)g * DR * DR;
        double AP[DR][DC];  // A[n,m] P_m
#pragma unroll
        for (int r = 0; r < DR; ++r)
#pragma unroll
          for (int c = 0; c < DC; ++c) {
            double t2 = 0.0;
#pragma unroll
            for (int t = 0; t < DR; ++t) t2 += a[r * DR + t] * Pm[t][c];
            AP[r][c] = t2;
          }
        int pos = 0;
        while (pos < clen - 1 && crow[pos] != J) ++pos;  // J is in the row by construction; bounded anyway
#pragma unroll
        for (int r = 0; r < DC; ++r)
#pragma unroll
          for (int c = 0; c < DC; ++c) {
            double t2 = 0.0;
#pragma unroll
            for (int t = 0; t < DR; ++t) t2 += Pn[t][r] * AP[t][c];
            __hip_atomic_fetch_add(cacc + (pos * DC + r) * DC + c, t2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
          }
      }
    }
    __syncthreads();
    double* dst = cvals + (int64_t)cb * DC * DC;
    for (int q = lane; q < clen * DC * DC; q += 64) dst[q] = cacc[q];
    __syncthreads();
  }
}

// generic block SpMV for the coarse levels: 64-node tiles, blocks streamed lane-per-block through LDS.
// MODE fuses the smoother into the product (the coarse levels are launch-bound, every fused pass counts):
//   AMG_AX   y = A x                         AMG_RESID  y = b - A x
//   AMG_SMOOTH y = x + dinv .* (b - A x)     (one damped-Jacobi sweep, y != x)
//   AMG_SMOOTH0 two sweeps from a zero guess: x1 = dinv .* b (formed on the fly), y = x1 + dinv2 .* (b - A x1)
//   AMG_POW  y = (A x) ./ dinv  (one power-iteration step of D^-1 A during the numeric setup, when dinv still holds the RAW
//            diagonal) and, if dinv2 != NULL, dinv2[wave] = this wave's share of |y|^2 (fixed order: reproducible)
enum { AMG_AX = 0, AMG_SMOOTH = 1, AMG_RESID = 2, AMG_SMOOTH0 = 3, AMG_POW = 4 };
#ifndef PF_AB_AS_U
#define PF_AB_AS_U 4  // blocks per lane in flight in k_amg_spmv
#endif
#ifndef PF_AB_RS_U
#define PF_AB_RS_U 4  // entries per lane in flight in k_sa_restrict_stream
#endif
// TR = rows per wave tile.  64 on the large levels (bandwidth); 16 on the small ones (n <= AMG_SMALL_LEVEL), where a tile's
// ~25 x 64 entries took two to four dependent chunks and the kernel 8 - 9 us of pure latency against ~5.5 us with a
// single chunk per wave and four times as many waves (r03).
#ifndef PF_AB_SMALL_LEVEL
#define PF_AB_SMALL_LEVEL 65536
#endif
static constexpr int64_t AMG_SMALL_LEVEL = PF_AB_SMALL_LEVEL;
// NV > 1: NV vectors at once, interleaved per dof (entry (dof q, vector v) at q * NV + v); dinv is per dof.  The matrix
// stream and the launch are shared by the NV products (the two gradient filters of a topology iteration, pf_cg2).
template <int D, int MODE, typename VT, int TR = 64, int NV = 1>
__global__ __launch_bounds__(64) void k_amg_spmv(int64_t n, const int32_t* __restrict__ ptr,
                                                 const int32_t* __restrict__ idx, const VT* __restrict__ vals,
                                                 const double* __restrict__ x, double* __restrict__ y,
                                                 const double* __restrict__ dinv, const double* __restrict__ dinv2,
                                                 const double* __restrict__ b) {
  constexpr int CH = 448;
  __shared__ double prod[CH * D * NV];
  const int lane = threadIdx.x;
  double yy = 0.0;  // AMG_POW
  const int64_t ntiles = (n + TR - 1) / TR;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t i = tile * TR + lane;
    const bool valid = lane < TR && i < n;
    const int32_t rb = valid ? ptr[i] : 0, re = valid ? ptr[i + 1] : 0;
    const int32_t base = __shfl(rb, 0);
    const int32_t end = ptr[min(n, (tile + 1) * TR)];
    double acc[D * NV];
#pragma unroll
    for (int k = 0; k < D * NV; ++k) acc[k] = 0.0;
    for (int32_t c0 = base; c0 < end; c0 += CH) {
      const int32_t c1 = min(end, c0 + CH);
      // U blocks per lane in flight (column ids first, then blocks and gathers together): these levels wait for memory
      constexpr int U = (D <= 3) ? PF_AB_AS_U : 1;
      for (int32_t g0 = c0 + lane; g0 < c1; g0 += 64 * U) {
        int32_t gg[U];
        int64_t mm[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          gg[u] = min(g0 + 64 * u, c1 - 1);  // clamped: loads stay in range, the result is discarded
          mm[u] = idx[gg[u]];
        }
        VT av[U][D * D];
        double xv[U][D * NV];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const VT* a = vals + (int64_t)gg[u] * D * D;  // VT = float: the fp32 copy of the coarse operator (cycle only)
#pragma unroll
          for (int q = 0; q < D * D; ++q) av[u][q] = a[q];
#pragma unroll
          for (int l = 0; l < D; ++l)
#pragma unroll
            for (int v = 0; v < NV; ++v)
              xv[u][l * NV + v] = (MODE == AMG_SMOOTH0) ? dinv[D * mm[u] + l] * b[(D * mm[u] + l) * NV + v] : x[(D * mm[u] + l) * NV + v];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (g0 + 64 * u >= c1) continue;
#pragma unroll
          for (int k = 0; k < D; ++k)
#pragma unroll
            for (int v = 0; v < NV; ++v) {
              double s = 0.0;
#pragma unroll
              for (int l = 0; l < D; ++l) s += (double)av[u][k * D + l] * xv[u][l * NV + v];
              prod[(gg[u] - c0) * D * NV + k * NV + v] = s;
            }
        }
      }
      __syncthreads();
      for (int32_t g = max(rb, c0); g < min(re, c1); ++g)
#pragma unroll
        for (int k = 0; k < D * NV; ++k) acc[k] += prod[(g - c0) * D * NV + k];
      __syncthreads();
    }
    if (valid)
#pragma unroll
      for (int k = 0; k < D; ++k)
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          const int64_t q = D * i + k, qv = q * NV + v;
          double out = acc[k * NV + v];
          if constexpr (MODE == AMG_RESID) out = b[qv] - out;
          if constexpr (MODE == AMG_SMOOTH) out = x[qv] + dinv[q] * (b[qv] - out);
          if constexpr (MODE == AMG_SMOOTH0) out = dinv[q] * b[qv] + dinv2[q] * (b[qv] - out);  // x1 + D2 (b - A x1)
          if constexpr (MODE == AMG_POW) { out = out / dinv[q]; yy += out * out; }
          y[qv] = out;
        }
  }
  if constexpr (MODE == AMG_POW) {
    if (dinv2) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) yy += __shfl_xor(yy, o);
      if (lane == 0) const_cast<double*>(dinv2)[blockIdx.x] = yy;
    }
  }
}

// x = dinv .* b   (one damped-Jacobi sweep from a zero guess; dinv already holds omega/diag)
template <int NV = 1>
__global__ void k_amg_jacobi0(int64_t n, const double* __restrict__ dinv, const double* __restrict__ b,
                              double* __restrict__ x) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n * NV; i += (int64_t)gridDim.x * blockDim.x)
    x[i] = dinv[i / NV] * b[i];
}
// coarse rhs: bc_I = sum_{n in I} P_n^T res_n
template <int DIM, int DR, int DC>
__global__ void k_amg_restrict(int64_t nc, const int32_t* __restrict__ aptr, const int32_t* __restrict__ amem,
                               const uint8_t* __restrict__ iso, const double* __restrict__ xy,
                               const double* __restrict__ cxy, const double* __restrict__ resid,
                               double* __restrict__ bc) {
  const int64_t I = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (I >= nc) return;
  double s[DC];
#pragma unroll
  for (int c = 0; c < DC; ++c) s[c] = 0.0;
  for (int32_t k = aptr[I]; k < aptr[I + 1]; ++k) {
    const int64_t n = amem[k];
    double Pn[DR][DC];
    amg_P<DIM, DR, DC>(xy + DIM * n, cxy + DIM * I, iso + DR * n, Pn);
#pragma unroll
    for (int r = 0; r < DR; ++r) {
      const double res = resid[DR * n + r];
#pragma unroll
      for (int c = 0; c < DC; ++c) s[c] += Pn[r][c] * res;
    }
  }
#pragma unroll
  for (int c = 0; c < DC; ++c) bc[DC * I + c] = s[c];
}

// x_n += P_n e_{agg[n]}
template <int DIM, int DR, int DC>
__global__ void k_amg_prolong(int64_t n, const int32_t* __restrict__ agg, const uint8_t* __restrict__ iso,
                              const double* __restrict__ xy, const double* __restrict__ cxy,
                              const double* __restrict__ ec, double* __restrict__ x, double over) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t I = agg[i];
  double Pn[DR][DC];
  amg_P<DIM, DR, DC>(xy + DIM * i, cxy + DIM * I, iso + DR * i, Pn);
#pragma unroll
  for (int r = 0; r < DR; ++r) {
    double s = 0.0;
#pragma unroll
    for (int c = 0; c < DC; ++c) s += Pn[r][c] * ec[DC * I + c];
    x[DR * i + r] += over * s;
  }
}

// ---- smoothed aggregation: stored prolongator ------------------------------------------------------------
// P = (I - w D^-1 A) P_tent, w = 4 / (3 * 1.1 * lambda_max(D^-1 A)); one lane per fine node owns its row of P.
// diag holds the raw diagonal, lam2 the squared estimate of lambda_max from the power iteration.
#ifndef PF_AB_SMOOTHP_UNROLL
#define PF_AB_SMOOTHP_UNROLL 3
#endif
template <int DIM, int DR, int DC>
__global__ void k_sa_smooth_P(int64_t n, const int32_t* __restrict__ ptr, const int32_t* __restrict__ idx,
                              const double* __restrict__ vals, const int32_t* __restrict__ agg,
                              const uint8_t* __restrict__ iso, const double* __restrict__ xy,
                              const double* __restrict__ cxy, const double* __restrict__ diag,
                              const double* __restrict__ lam2, const int32_t* __restrict__ pptr,
                              const int32_t* __restrict__ pidx, double* __restrict__ pvals, double wscale) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double w = wscale * 4.0 / (3.3 * sqrt(lam2[0]));
  const int32_t pb = pptr[i], pe = pptr[i + 1];
  double wr[DR];
#pragma unroll
  for (int r = 0; r < DR; ++r) wr[r] = iso[DR * i + r] ? 0.0 : w / diag[DR * i + r];
  const int32_t I = agg[i];
  if (DR * DC <= 9 && pe - pb <= 4) {  // (3-D blocks: four slots of 18 or 36 doubles would spill)
    // Rows of at most four entries (every row of a structured mesh, most rows of any 2-D mesh): the row is accumulated in
    // registers and written once.  The general path below reads and writes its entries in global memory once per
    // neighbour -- r02 counters: the address unit was 100 % busy with 9.5 GB of L2 requests for 0.85 GB of data.
    int32_t col[4];
    double acc[4][DR * DC];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      col[k] = (pb + k < pe) ? pidx[pb + k] : -1;
#pragma unroll
      for (int q = 0; q < DR * DC; ++q) acc[k][q] = 0.0;
    }
    {
      double Pn[DR][DC];
      amg_P<DIM, DR, DC>(xy + DIM * i, cxy + DIM * (int64_t)I, iso + DR * i, Pn);
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (col[k] == I) {
#pragma unroll
          for (int r = 0; r < DR; ++r)
#pragma unroll
            for (int c = 0; c < DC; ++c) acc[k][r * DC + c] = Pn[r][c];
        }
    }
    // the neighbours three at a time -- ids first, then their aggregates, then blocks and coordinates together: a row's
    // nine neighbours are nine chains of four dependent loads otherwise, and the kernel waits for them
    constexpr int UN = PF_AB_SMOOTHP_UNROLL;
    const int32_t ge = ptr[i + 1];
    for (int32_t g0 = ptr[i]; g0 < ge; g0 += UN) {
      int32_t gg[UN], JJ[UN];
      int64_t mm[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) { gg[u] = min(g0 + u, ge - 1); mm[u] = idx[gg[u]]; }
#pragma unroll
      for (int u = 0; u < UN; ++u) JJ[u] = agg[mm[u]];
      double Pm[UN][DR][DC], a[UN][DR * DR];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        amg_P<DIM, DR, DC>(xy + DIM * mm[u], cxy + DIM * (int64_t)JJ[u], iso + DR * mm[u], Pm[u]);
        const double* ap = vals + (int64_t)gg[u] * DR * DR;
#pragma unroll
        for (int q = 0; q < DR * DR; ++q) a[u][q] = ap[q];
      }
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        if (g0 + u >= ge) continue;
        double d[DR * DC];
#pragma unroll
        for (int r = 0; r < DR; ++r)
#pragma unroll
          for (int c = 0; c < DC; ++c) {
            double t2 = 0.0;
#pragma unroll
            for (int t = 0; t < DR; ++t) t2 += a[u][r * DR + t] * Pm[u][t][c];
            d[r * DC + c] = wr[r] * t2;
          }
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (col[k] == JJ[u]) {
#pragma unroll
            for (int q = 0; q < DR * DC; ++q) acc[k][q] -= d[q];
          }
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (pb + k < pe) {
        double* dst = pvals + (int64_t)(pb + k) * DR * DC;
#pragma unroll
        for (int q = 0; q < DR * DC; ++q) dst[q] = acc[k][q];
      }
    return;
  }
  for (int64_t q = (int64_t)pb * DR * DC; q < (int64_t)pe * DR * DC; ++q) pvals[q] = 0.0;
  {
    double Pn[DR][DC];
    amg_P<DIM, DR, DC>(xy + DIM * i, cxy + DIM * (int64_t)I, iso + DR * i, Pn);
    int32_t s = pb;
    while (s < pe - 1 && pidx[s] != I) ++s;
    double* dst = pvals + (int64_t)s * DR * DC;
#pragma unroll
    for (int r = 0; r < DR; ++r)
#pragma unroll
      for (int c = 0; c < DC; ++c) dst[r * DC + c] = Pn[r][c];
  }
  for (int32_t g = ptr[i]; g < ptr[i + 1]; ++g) {
    const int64_t m = idx[g];
    const int32_t J = agg[m];
    double Pm[DR][DC];
    amg_P<DIM, DR, DC>(xy + DIM * m, cxy + DIM * (int64_t)J, iso + DR * m, Pm);
    const double* a = vals + (int64_t)g * DR * DR;
    int32_t s = pb;
    while (s < pe - 1 && pidx[s] != J) ++s;
    double* dst = pvals + (int64_t)s * DR * DC;
#pragma unroll
    for (int r = 0; r < DR; ++r)
#pragma unroll
      for (int c = 0; c < DC; ++c) {
        double acc = 0.0;
#pragma unroll
        for (int t = 0; t < DR; ++t) acc += a[r * DR + t] * Pm[t][c];
        dst[r * DC + c] -= wr[r] * acc;
      }
  }
}

// Galerkin product with the stored P, one lane per coarse row I:
//   C[I,J] = sum_{n in col I} sum_{m in N(n)} P[n,I]^T A[n,m] P[m,J]
template <int DR, int DC>
__global__ void k_sa_galerkin(int64_t nc, const int32_t* __restrict__ tptr, const int32_t* __restrict__ tnode,
                              const int32_t* __restrict__ tslot, const int32_t* __restrict__ ptr,
                              const int32_t* __restrict__ idx, const double* __restrict__ vals,
                              const int32_t* __restrict__ pptr, const int32_t* __restrict__ pidx,
                              const double* __restrict__ pvals, const int32_t* __restrict__ cptr,
                              const int32_t* __restrict__ cidx, double* __restrict__ cvals) {
  const int64_t I = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (I >= nc) return;
  const int32_t cb = cptr[I], ce = cptr[I + 1];
  for (int64_t q = (int64_t)cb * DC * DC; q < (int64_t)ce * DC * DC; ++q) cvals[q] = 0.0;
  for (int32_t e = tptr[I]; e < tptr[I + 1]; ++e) {
    const int64_t n = tnode[e];
    const double* Pn = pvals + (int64_t)tslot[e] * DR * DC;
    double pn[DR][DC];
#pragma unroll
    for (int r = 0; r < DR; ++r)
#pragma unroll
      for (int c = 0; c < DC; ++c) pn[r][c] = Pn[r * DC + c];
    for (int32_t g = ptr[n]; g < ptr[n + 1]; ++g) {
      const int64_t m = idx[g];
      const double* a = vals + (int64_t)g * DR * DR;
      double T[DC][DR];  // P[n,I]^T A[n,m]
#pragma unroll
      for (int c = 0; c < DC; ++c)
#pragma unroll
        for (int t = 0; t < DR; ++t) {
          double s = 0.0;
#pragma unroll
          for (int r = 0; r < DR; ++r) s += pn[r][c] * a[r * DR + t];
          T[c][t] = s;
        }
      int32_t pos = cb;
      for (int32_t s2 = pptr[m]; s2 < pptr[m + 1]; ++s2) {
        const int32_t J = pidx[s2];  // ascending per row, like cidx: the search resumes where it stopped
        while (pos < ce - 1 && cidx[pos] != J) ++pos;
        const double* Pm = pvals + (int64_t)s2 * DR * DC;
        double* dst = cvals + (int64_t)pos * DC * DC;
#pragma unroll
        for (int c = 0; c < DC; ++c)
#pragma unroll
          for (int c2 = 0; c2 < DC; ++c2) {
            double s = 0.0;
#pragma unroll
            for (int t = 0; t < DR; ++t) s += T[c][t] * Pm[t * DC + c2];
            dst[c * DC + c2] += s;
          }
      }
    }
  }
}

// The same product with one WAVE per coarse row (rows of at most 64 couplings): lanes split the (n, m) pairs --
// E entries of the column of I times G neighbour slots per pass -- and add their DC x DC contributions
// into an LDS image of the row with LDS atomics; the row is then written out once.  NW waves share a row (its
// column entries dealt round-robin) on levels with few, long columns: 1331 coarse rows of ~200 entries each on the
// second level of 96^3 blocks kept one wave per row busy for 7 ms.  The lane-per-row kernel above
// takes 15 ms for the 233 k coarse rows of the 2048x1024 plane-stress matrix; this one well under one.
template <int DR, int DC, int NW>
__global__ __launch_bounds__(64 * NW) void k_sa_galerkin_wave(int64_t nc, const int32_t* __restrict__ tptr,
                                                         const int32_t* __restrict__ tnode,
                                                         const int32_t* __restrict__ tslot,
                                                         const int32_t* __restrict__ ptr, const int32_t* __restrict__ idx,
                                                         const double* __restrict__ vals,
                                                         const int32_t* __restrict__ pptr,
                                                         const int32_t* __restrict__ pidx,
                                                         const double* __restrict__ pvals,
                                                         const int32_t* __restrict__ cptr,
                                                         const int32_t* __restrict__ cidx, double* __restrict__ cvals,
                                                         int G) {
  // G neighbour slots per column entry (the fine level's longest row, at most 32), E = 64 / G entries per pass: 9 x 7
  // on quads instead of a fixed 16 x 4 takes the 25 entries of a column in 4 passes instead of 7
  __shared__ int32_t crow[64];
  __shared__ double cacc[64 * DC * DC];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int E = 64 / G, es = lane / G, gs = lane - es * G;
  for (int64_t I = blockIdx.x; I < nc; I += gridDim.x) {
    const int32_t cb = cptr[I], clen = cptr[I + 1] - cb;
    if ((int)threadIdx.x < clen) crow[threadIdx.x] = cidx[cb + threadIdx.x];
    for (int q = threadIdx.x; q < clen * DC * DC; q += 64 * NW) cacc[q] = 0.0;
    __syncthreads();
    const int32_t tb = tptr[I], te = tptr[I + 1];
    // Contributions to the DIAGONAL block (J == I: every fine node of the column couples back into its own aggregate) would
    // all land on the same DC x DC addresses -- r02 counters: 97 % of this kernel's issue stalls wait for the LDS, 39 % of
    // its LDS cycles are conflicts.  They are summed in registers per lane and reduced across the wave once per row.
    constexpr bool DIAGREG = DC == 3;  // 2-D elasticity levels: -15 % on the finest one; 6 x 6 blocks (3-D) would cost 72 registers, scalar rows do not pay for the butterfly
    double dg[DIAGREG ? DC : 1][DIAGREG ? DC : 1];
    if constexpr (DIAGREG) {
#pragma unroll
      for (int c = 0; c < DC; ++c)
#pragma unroll
        for (int c2 = 0; c2 < DC; ++c2) dg[c][c2] = 0.0;
    }
    for (int32_t e0 = tb + E * wave; e0 < te; e0 += E * NW) {
      const int32_t e = e0 + es;
      const bool eok = es < E && e < te;
      const int64_t n = eok ? tnode[e] : 0;
      const int32_t rb = eok ? ptr[n] : 0, rl = eok ? ptr[n + 1] - rb : 0;
      double pn[DR][DC];
      if (eok) {
        const double* Pn = pvals + (int64_t)tslot[e] * DR * DC;
#pragma unroll
        for (int r = 0; r < DR; ++r)
#pragma unroll
          for (int c = 0; c < DC; ++c) pn[r][c] = Pn[r * DC + c];
      }
      for (int32_t g0 = 0; g0 < rl; g0 += G) {  // rl = 0 for idle lanes
        const int32_t go = g0 + gs;
        if (go < rl) {
          const int32_t g = rb + go;
          const int64_t m = idx[g];
          const double* av = vals + (int64_t)g * DR * DR;
          double T[DC][DR];
#pragma unroll
          for (int c = 0; c < DC; ++c)
#pragma unroll
            for (int t = 0; t < DR; ++t) {
              double s = 0.0;
#pragma unroll
              for (int r = 0; r < DR; ++r) s += pn[r][c] * av[r * DR + t];
              T[c][t] = s;
            }
          int pos = 0;
          for (int32_t s2 = pptr[m]; s2 < pptr[m + 1]; ++s2) {
            const int32_t J = pidx[s2];
            if (J < (int32_t)I) continue;  // A_c is symmetric: the lower triangle is mirrored by k_sa_mirror_lower
            const double* Pm = pvals + (int64_t)s2 * DR * DC;
            if (DIAGREG && J == (int32_t)I) {
#pragma unroll
              for (int c = 0; c < DC; ++c)
#pragma unroll
                for (int c2 = 0; c2 < DC; ++c2) {
#pragma unroll
                  for (int t = 0; t < DR; ++t) dg[DIAGREG ? c : 0][DIAGREG ? c2 : 0] += T[c][t] * Pm[t * DC + c2];
                }
              continue;
            }
            while (pos < clen - 1 && crow[pos] != J) ++pos;
#pragma unroll
            for (int c = 0; c < DC; ++c)
#pragma unroll
              for (int c2 = 0; c2 < DC; ++c2) {
                double s = 0.0;
#pragma unroll
                for (int t = 0; t < DR; ++t) s += T[c][t] * Pm[t * DC + c2];
                __hip_atomic_fetch_add(cacc + (pos * DC + c) * DC + c2, s, __ATOMIC_RELAXED,
                                       NW == 1 ? __HIP_MEMORY_SCOPE_WAVEFRONT : __HIP_MEMORY_SCOPE_WORKGROUP);
              }
          }
        }
      }
    }
    if constexpr (DIAGREG) {  // the diagonal block: butterfly over the wave (every lane ends with the same bits), one lane adds
      int dpos = 0;
      while (dpos < clen - 1 && crow[dpos] != (int32_t)I) ++dpos;
#pragma unroll
      for (int c = 0; c < DC; ++c)
#pragma unroll
        for (int c2 = 0; c2 < DC; ++c2) {
          double v = dg[c][c2];
#pragma unroll
          for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
          if (lane == 0)
            __hip_atomic_fetch_add(cacc + (dpos * DC + c) * DC + c2, v, __ATOMIC_RELAXED,
                                   NW == 1 ? __HIP_MEMORY_SCOPE_WAVEFRONT : __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    __syncthreads();
    double* dst = cvals + (int64_t)cb * DC * DC;
    for (int q = threadIdx.x; q < clen * DC * DC; q += 64 * NW) dst[q] = cacc[q];
    __syncthreads();
  }
}

// C[I,J] = C[J,I]^T for J < I, after k_sa_galerkin_wave has filled the upper triangle (halves its LDS adds)
template <int DC>
__global__ void k_sa_mirror_lower(int64_t nc, const int32_t* __restrict__ cptr, const int32_t* __restrict__ cidx,
                                  double* __restrict__ cvals) {
  const int64_t I = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (I >= nc) return;
  for (int32_t pos = cptr[I]; pos < cptr[I + 1]; ++pos) {
    const int32_t J = cidx[pos];
    if (J >= (int32_t)I) break;  // ascending columns: the rest is the upper triangle
    int32_t q = cptr[J];
    const int32_t qe = cptr[J + 1];
    while (q < qe - 1 && cidx[q] != (int32_t)I) ++q;
    const double* src = cvals + (int64_t)q * DC * DC;
    double* dst = cvals + (int64_t)pos * DC * DC;
#pragma unroll
    for (int c = 0; c < DC; ++c)
#pragma unroll
      for (int c2 = 0; c2 < DC; ++c2) dst[c * DC + c2] = src[c2 * DC + c];
  }
}

// ptvals[e] = pvals[tslot[e]]: P in transpose order, so that the restriction streams it
template <typename PT>
__global__ void k_sa_transpose_vals(int64_t np, int bs, const int32_t* __restrict__ tslot,
                                    const double* __restrict__ pvals, PT* __restrict__ ptvals) {
  const int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (q >= np * bs) return;
  const int64_t e = q / bs;
  ptvals[q] = (PT)pvals[(int64_t)tslot[e] * bs + (q - e * bs)];
}

// restriction as a streamed block product with P^T (same scheme as k_amg_spmv: 64 coarse rows per wave tile,
// lanes stream the tile's contiguous entries, products parked in LDS, lane-per-row sums)
template <int DR, int DC, typename PT = double, int TR = 64, int NV = 1>
__global__ __launch_bounds__(64) void k_sa_restrict_stream(int64_t nc, const int32_t* __restrict__ tptr,
                                                           const int32_t* __restrict__ tnode,
                                                           const PT* __restrict__ ptvals,
                                                           const double* __restrict__ resid, double* __restrict__ bc) {
  constexpr int CH = 512;
  __shared__ double prod[CH * DC * NV];
  const int lane = threadIdx.x;
  const int64_t ntiles = (nc + TR - 1) / TR;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t I = tile * TR + lane;
    const bool valid = lane < TR && I < nc;
    const int32_t rb = valid ? tptr[I] : 0, re = valid ? tptr[I + 1] : 0;
    const int32_t base = __shfl(rb, 0);
    const int32_t end = tptr[min(nc, (tile + 1) * TR)];
    double acc[DC * NV];
#pragma unroll
    for (int c = 0; c < DC * NV; ++c) acc[c] = 0.0;
    for (int32_t c0 = base; c0 < end; c0 += CH) {
      const int32_t c1 = min(end, c0 + CH);
      // U entries per lane in flight (the node ids first, then blocks and residual gathers together): the kernel waits
      // for memory 93 % of its time at 13 resident waves per CU (r02 counters)
      constexpr int U = (DR * DC <= 9) ? PF_AB_RS_U : 1;
      for (int32_t e0 = c0 + lane; e0 < c1; e0 += 64 * U) {
        int32_t ee[U];
        int64_t nn[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          ee[u] = min(e0 + 64 * u, c1 - 1);  // clamped: loads stay in range, the result is discarded
          nn[u] = tnode[ee[u]];
        }
        double Pv[U][DR * DC], rv[U][DR * NV];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const PT* Pn = ptvals + (int64_t)ee[u] * DR * DC;
#pragma unroll
          for (int q = 0; q < DR * DC; ++q) Pv[u][q] = (double)Pn[q];
#pragma unroll
          for (int r = 0; r < DR * NV; ++r) rv[u][r] = resid[DR * NV * nn[u] + r];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (e0 + 64 * u >= c1) continue;
#pragma unroll
          for (int c = 0; c < DC; ++c)
#pragma unroll
            for (int v = 0; v < NV; ++v) {
              double s = 0.0;
#pragma unroll
              for (int r = 0; r < DR; ++r) s += Pv[u][r * DC + c] * rv[u][r * NV + v];
              prod[(c * NV + v) * CH + (ee[u] - c0)] = s;
            }
        }
      }
      __syncthreads();
      for (int32_t e = max(rb, c0); e < min(re, c1); ++e)
#pragma unroll
        for (int c = 0; c < DC * NV; ++c) acc[c] += prod[c * CH + (e - c0)];
      __syncthreads();
    }
    if (valid)
#pragma unroll
      for (int c = 0; c < DC * NV; ++c) bc[DC * NV * I + c] = acc[c];
  }
}

// The same restriction with one wave per coarse row, for levels whose columns of P are long and few (3-D: ~200
// entries per coarse node, 1331 coarse nodes on the second level of 96^3 blocks -- 21 tiles of 64 rows left the
// stream kernel on 21 waves, 300 us per call).  Lanes stride over the column, fixed-order wave sum: deterministic.
template <int DR, int DC, typename PT = double, int NV = 1>
__global__ __launch_bounds__(256) void k_sa_restrict_wave(int64_t nc, const int32_t* __restrict__ tptr,
                                                          const int32_t* __restrict__ tnode,
                                                          const PT* __restrict__ ptvals,
                                                          const double* __restrict__ resid, double* __restrict__ bc) {
  const int lane = threadIdx.x & 63;
  const int64_t I = blockIdx.x * (int64_t)(blockDim.x >> 6) + (threadIdx.x >> 6);
  if (I >= nc) return;  // wave-uniform
  double acc[DC * NV];
#pragma unroll
  for (int c = 0; c < DC * NV; ++c) acc[c] = 0.0;
  for (int32_t e = tptr[I] + lane; e < tptr[I + 1]; e += 64) {
    const int64_t n = tnode[e];
    const PT* Pn = ptvals + (int64_t)e * DR * DC;
    double rv[DR * NV];
#pragma unroll
    for (int r = 0; r < DR * NV; ++r) rv[r] = resid[DR * NV * n + r];
#pragma unroll
    for (int r = 0; r < DR; ++r)
#pragma unroll
      for (int c = 0; c < DC; ++c)
#pragma unroll
        for (int v = 0; v < NV; ++v) acc[c * NV + v] += (double)Pn[r * DC + c] * rv[r * NV + v];
  }
#pragma unroll
  for (int c = 0; c < DC * NV; ++c)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc[c] += __shfl_xor(acc[c], o);
  if (lane == 0)
#pragma unroll
    for (int c = 0; c < DC * NV; ++c) bc[DC * NV * I + c] = acc[c];
}

// x_n += over * sum_J P[n,J] e_J
template <int DR, int DC, typename PT = double, int NV = 1>
__global__ void k_sa_prolong(int64_t n, const int32_t* __restrict__ pptr, const int32_t* __restrict__ pidx,
                             const PT* __restrict__ pvals, const double* __restrict__ ec, double* __restrict__ x,
                             double over) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  double s[DR * NV];
#pragma unroll
  for (int r = 0; r < DR * NV; ++r) s[r] = 0.0;
  const int32_t qb = pptr[i], qe = pptr[i + 1];
  if constexpr (DR * DC <= 9) {
    // four entries at a time: their aggregate ids first, then the blocks and the coarse values together (independent
    // loads in flight instead of one dependent chain per entry)
    for (int32_t q0 = qb; q0 < qe; q0 += 4) {
      int32_t qq[4];
      int64_t JJ[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        qq[u] = min(q0 + u, qe - 1);
        JJ[u] = pidx[qq[u]];
      }
      double Pv[4][DR * DC], ev[4][DC * NV];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const PT* Pn = pvals + (int64_t)qq[u] * DR * DC;
#pragma unroll
        for (int k = 0; k < DR * DC; ++k) Pv[u][k] = (double)Pn[k];
#pragma unroll
        for (int c = 0; c < DC * NV; ++c) ev[u][c] = ec[DC * NV * JJ[u] + c];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (q0 + u >= qe) continue;
#pragma unroll
        for (int r = 0; r < DR; ++r)
#pragma unroll
          for (int c = 0; c < DC; ++c)
#pragma unroll
            for (int v = 0; v < NV; ++v) s[r * NV + v] += Pv[u][r * DC + c] * ev[u][c * NV + v];
      }
    }
  } else {
    for (int32_t q = qb; q < qe; ++q) {
      const PT* Pn = pvals + (int64_t)q * DR * DC;
      const double* e = ec + DC * NV * (int64_t)pidx[q];
#pragma unroll
      for (int r = 0; r < DR; ++r)
#pragma unroll
        for (int c = 0; c < DC; ++c)
#pragma unroll
          for (int v = 0; v < NV; ++v) s[r * NV + v] += (double)Pn[r * DC + c] * e[c * NV + v];
    }
  }
#pragma unroll
  for (int r = 0; r < DR * NV; ++r) x[DR * NV * i + r] += over * s[r];
}

// coarsest level: scatter the block matrix into a dense m x m array, invert it in place (Gauss-Jordan,
// SPD so no pivoting) with one workgroup, then x = Ainv b per cycle
template <int D>
__global__ void k_amg_to_dense(int64_t n, const int32_t* __restrict__ ptr, const int32_t* __restrict__ idx,
                               const double* __restrict__ vals, double* __restrict__ A, int64_t m) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  for (int32_t g = ptr[i]; g < ptr[i + 1]; ++g)
    for (int k = 0; k < D; ++k)
      for (int l = 0; l < D; ++l) A[(D * i + k) * m + D * (int64_t)idx[g] + l] = vals[(int64_t)g * D * D + k * D + l];
}

template <int T>  // T x T entries per thread: T = 4 covers m <= 128 (the 108 dofs of a 36-node plane coarsest level), 5 the cap
__global__ __launch_bounds__(1024) void k_amg_dense_invert(double* __restrict__ A, int m) {
  // in-place Gauss-Jordan without pivoting (definite matrix); rows/cols with a zero pivot are treated as identity.
  // The matrix lives in registers: thread (ty, tx) owns the entries (ty + 32 a, tx + 32 b); each step the owners of
  // the pivot row and column publish them through LDS (double-buffered: one barrier per step).
  static_assert(T * 32 >= 128 && (AMG_COARSEST_DOFS + 31) / 32 <= 5, "dense inverse tiles");
  __shared__ double piv_row[2][T * 32];
  __shared__ double piv_col[2][T * 32];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  double a[T][T];
#pragma unroll
  for (int ia = 0; ia < T; ++ia)
#pragma unroll
    for (int jb = 0; jb < T; ++jb) {
      const int i = ty + 32 * ia, j = tx + 32 * jb;
      a[ia][jb] = (i < m && j < m) ? A[i * m + j] : 0.0;
    }
  for (int p = 0; p < m; ++p) {
    const int s = p & 1;
#pragma unroll
    for (int ia = 0; ia < T; ++ia)
#pragma unroll
      for (int jb = 0; jb < T; ++jb) {
        const int i = ty + 32 * ia, j = tx + 32 * jb;
        if (i == p) piv_row[s][j] = a[ia][jb];
        if (j == p) piv_col[s][i] = a[ia][jb];
      }
    __syncthreads();
    double pv = piv_row[s][p];
    if (pv == 0.0) pv = 1.0;
    const double ip = 1.0 / pv;
#pragma unroll
    for (int ia = 0; ia < T; ++ia) {
      const int i = ty + 32 * ia;
      const double ci = piv_col[s][i];
#pragma unroll
      for (int jb = 0; jb < T; ++jb) {
        const int j = tx + 32 * jb;
        double v;
        if (i == p && j == p) v = ip;
        else if (i == p) v = piv_row[s][j] * ip;
        else if (j == p) v = -ci * ip;
        else v = a[ia][jb] - ci * piv_row[s][j] * ip;
        a[ia][jb] = v;
      }
    }
  }
#pragma unroll
  for (int ia = 0; ia < T; ++ia)
#pragma unroll
    for (int jb = 0; jb < T; ++jb) {
      const int i = ty + 32 * ia, j = tx + 32 * jb;
      if (i < m && j < m) A[i * m + j] = a[ia][jb];
    }
}

// x = Ainv b; Ainv is symmetric, so lane i walks column i (= row i) with coalesced reads
// 4 x 192 threads: thread (part, i) sums the rows j = part (mod 4) of column i, the four partial sums meet in LDS -- a
// single thread per entry walked all m rows in one dependent chain (13.8 us for 114 dofs, three launch floors)
template <int NV = 1>
__global__ __launch_bounds__(768) void k_amg_dense_apply(int m, const double* __restrict__ Ainv, const double* __restrict__ b,
                                                         double* __restrict__ x) {
  __shared__ double sb[AMG_COARSEST_DOFS * NV];
  __shared__ double ps[4][192 * NV];
  static_assert(AMG_COARSEST_DOFS <= 192, "k_amg_dense_apply: 192 columns per part");
  for (int j = threadIdx.x; j < m * NV; j += blockDim.x) sb[j] = b[j];
  __syncthreads();
  const int i = threadIdx.x % 192, part = threadIdx.x / 192;
  double s[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) s[v] = 0.0;
  if (i < m) {
#pragma unroll 4
    for (int j = part; j < m; j += 4) {
      const double a = Ainv[j * m + i];
#pragma unroll
      for (int v = 0; v < NV; ++v) s[v] += a * sb[j * NV + v];
    }
  }
#pragma unroll
  for (int v = 0; v < NV; ++v) ps[part][i * NV + v] = s[v];
  __syncthreads();
  if (part == 0 && i < m) {
#pragma unroll
    for (int v = 0; v < NV; ++v) x[i * NV + v] = (ps[0][i * NV + v] + ps[1][i * NV + v]) + (ps[2][i * NV + v] + ps[3][i * NV + v]);
  }
}

// ------------------------------------------------------------------------------------------
// The same patterns on the device (sa_pattern_host is the fallback when a row exceeds the local-array caps):
// rows of P, its transpose (entries of each coarse column, ascending node) and the rows of P^T A P, each as a
// count pass, a scan and a fill pass with per-thread sorted-unique arrays.  Everything is sorted, so the result is
// identical to the host version.
// ------------------------------------------------------------------------------------------
static constexpr int SA_PCAP = 32;    // aggregates one row of P may touch
static constexpr int SA_CCAP = 160;   // couplings of one coarse row

__device__ __forceinline__ bool sa_insert(int32_t* arr, int& cnt, int cap, int32_t v) {
  int k = 0;
  while (k < cnt && arr[k] < v) ++k;
  if (k < cnt && arr[k] == v) return true;
  if (cnt == cap) return false;
  for (int t = cnt; t > k; --t) arr[t] = arr[t - 1];
  arr[k] = v;
  ++cnt;
  return true;
}

__global__ void k_sa_prow(int64_t n, const int32_t* __restrict__ ptr, const int32_t* __restrict__ idx,
                          const int32_t* __restrict__ agg, int32_t* __restrict__ count,
                          const int32_t* __restrict__ pptr, int32_t* __restrict__ pidx, int32_t* __restrict__ flag) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  int32_t arr[SA_PCAP];
  int cnt = 0;
  bool ok = sa_insert(arr, cnt, SA_PCAP, agg[i]);
  for (int32_t g = ptr[i]; g < ptr[i + 1]; ++g) ok = sa_insert(arr, cnt, SA_PCAP, agg[idx[g]]) && ok;
  if (!ok) flag[0] = 1;
  if (count) count[i] = cnt;
  else
    for (int k = 0; k < cnt; ++k) pidx[pptr[i] + k] = arr[k];
}

__global__ void k_sa_tcount(int64_t n, const int32_t* __restrict__ pptr, const int32_t* __restrict__ pidx,
                            int32_t* __restrict__ tcount) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  for (int32_t s = pptr[i]; s < pptr[i + 1]; ++s) atomicAdd(&tcount[pidx[s]], 1);
}

__global__ void k_sa_tfill(int64_t n, const int32_t* __restrict__ pptr, const int32_t* __restrict__ pidx,
                           const int32_t* __restrict__ tptr, int32_t* __restrict__ cursor,
                           int32_t* __restrict__ tnode, int32_t* __restrict__ tslot) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  for (int32_t s = pptr[i]; s < pptr[i + 1]; ++s) {
    const int32_t J = pidx[s];
    const int32_t at = tptr[J] + atomicAdd(&cursor[J], 1);
    tnode[at] = (int32_t)i;
    tslot[at] = s;
  }
}

// order each coarse column by node (the atomics above fill it in arrival order)
__global__ void k_sa_tsort(int64_t nc, const int32_t* __restrict__ tptr, int32_t* __restrict__ tnode,
                           int32_t* __restrict__ tslot) {
  const int64_t I = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (I >= nc) return;
  const int32_t b = tptr[I], e = tptr[I + 1];
  for (int32_t q = b + 1; q < e; ++q) {
    const int32_t vn = tnode[q], vs = tslot[q];
    int32_t j = q - 1;
    while (j >= b && tnode[j] > vn) {
      tnode[j + 1] = tnode[j];
      tslot[j + 1] = tslot[j];
      --j;
    }
    tnode[j + 1] = vn;
    tslot[j + 1] = vs;
  }
}

__global__ void k_sa_crow(int64_t nc, const int32_t* __restrict__ tptr, const int32_t* __restrict__ tnode,
                          const int32_t* __restrict__ ptr, const int32_t* __restrict__ idx,
                          const int32_t* __restrict__ pptr, const int32_t* __restrict__ pidx,
                          int32_t* __restrict__ count, const int32_t* __restrict__ cptr, int32_t* __restrict__ cidx,
                          int32_t* __restrict__ flag) {
  const int64_t I = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (I >= nc) return;
  int32_t arr[SA_CCAP];
  int cnt = 0;
  bool ok = sa_insert(arr, cnt, SA_CCAP, (int32_t)I);
  for (int32_t q = tptr[I]; q < tptr[I + 1]; ++q) {
    const int32_t i = tnode[q];
    for (int32_t g = ptr[i]; g < ptr[i + 1]; ++g) {
      const int32_t m = idx[g];
      for (int32_t s = pptr[m]; s < pptr[m + 1]; ++s) ok = sa_insert(arr, cnt, SA_CCAP, pidx[s]) && ok;
    }
  }
  if (!ok) flag[0] = 1;
  if (count) count[I] = cnt;
  else
    for (int k = 0; k < cnt; ++k) cidx[cptr[I] + k] = arr[k];
}

// rows of A P: the aggregates in the rows of P of node n's neighbours (count pass, then fill pass)
__global__ void k_sa_aprow(int64_t n, const int32_t* __restrict__ ptr, const int32_t* __restrict__ idx,
                           const int32_t* __restrict__ pptr, const int32_t* __restrict__ pidx, int32_t* __restrict__ count,
                           const int32_t* __restrict__ apptr, int32_t* __restrict__ apidx, int32_t* __restrict__ flag) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  int32_t arr[SA_PCAP];
  int cnt = 0;
  bool ok = true;
  for (int32_t g = ptr[i]; g < ptr[i + 1]; ++g) {
    const int32_t m = idx[g];
    for (int32_t s2 = pptr[m]; s2 < pptr[m + 1]; ++s2) ok = sa_insert(arr, cnt, SA_PCAP, pidx[s2]) && ok;
  }
  if (!ok) flag[0] = 1;
  if (count) count[i] = cnt;
  else
    for (int k = 0; k < cnt; ++k) apidx[apptr[i] + k] = arr[k];
}

// ---- Galerkin product in two steps (round 3) ----------------------------------------------------------------------
// k_sa_galerkin_wave forms P^T A P in one sweep: every (fine node of the column, neighbour, entry of the neighbour's row
// of P) triple adds a block into an LDS image of the coarse row -- 630 triples per row on quads, most of them onto a
// handful of addresses: 97 % of its issue stalls waited for the LDS, 39 % of the LDS cycles were same-address conflicts
// (profiles/r02_pmc_amg_setup.txt), 2.56 ms on configuration 5's finest level.  Two steps with NO atomics instead:
//   k_sa_ap     (A P)[n, J] = sum_m A[n, m] P[m, J]: one lane per (n, J) entry of the pattern of A P; it walks the row of A
//               and looks J up in the (<= 4-entry, sorted) row of P of each neighbour
//   k_sa_ptap   C[I, J] = sum_{n in column I of P} P[n, I]^T (A P)[n, J], J >= I: one lane per (I, J) entry of the coarse
//               graph; which (P block, A P block) pairs it sums is a GATHER PLAN built once per pattern (k_sa_gplan: 113
//               pairs per coarse row on quads, 8 bytes each) -- looking J up in the rows of A P inside the numeric kernel
//               cost a dozen dependent loads per fine node (first version: 1.8 ms).  Every output block is accumulated in
//               registers by its own lane in ascending fine-node order (deterministic) and written once;
//               k_sa_mirror_lower fills the lower triangle as before.
template <int DR, int DC>
__global__ __launch_bounds__(256) void k_sa_ap(int64_t nap, const int32_t* __restrict__ aprow, const int32_t* __restrict__ apidx,
                                              const int32_t* __restrict__ ptr, const int32_t* __restrict__ idx,
                                              const double* __restrict__ vals, const int32_t* __restrict__ pptr,
                                              const int32_t* __restrict__ pidx, const double* __restrict__ pvals,
                                              double* __restrict__ apvals) {
  // one lane per entry (n, J) of the pattern of A P; the entries of a node are consecutive, so the ~10 lanes of a node read
  // the same row of A and the same rows of P (broadcast loads).  aprow[q] = n (a bisection of apptr cost 21 dependent loads;
  // 16 lanes per node with a loop over the row's entries was twice as slow again: r03)
  const int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (q >= nap) return;
  const int64_t i = aprow[q];
  const int32_t J = apidx[q];
  double acc[DR][DC];
#pragma unroll
  for (int r = 0; r < DR; ++r)
#pragma unroll
    for (int c = 0; c < DC; ++c) acc[r][c] = 0.0;
  for (int32_t g = ptr[i]; g < ptr[i + 1]; ++g) {
    const int32_t m = idx[g];
    int32_t s2 = pptr[m];
    const int32_t se = pptr[m + 1];
    while (s2 < se && pidx[s2] < J) ++s2;
    if (s2 == se || pidx[s2] != J) continue;
    const double* a = vals + (int64_t)g * DR * DR;
    const double* Pm = pvals + (int64_t)s2 * DR * DC;
#pragma unroll
    for (int r = 0; r < DR; ++r)
#pragma unroll
      for (int c = 0; c < DC; ++c) {
        double sacc = 0.0;
#pragma unroll
        for (int tt = 0; tt < DR; ++tt) sacc += a[r * DR + tt] * Pm[tt * DC + c];
        acc[r][c] += sacc;
      }
  }
  double* dst = apvals + q * DR * DC;
#pragma unroll
  for (int r = 0; r < DR; ++r)
#pragma unroll
    for (int c = 0; c < DC; ++c) dst[r * DC + c] = acc[r][c];
}

// Plan of the first step, once per pattern: entry q = (n, J) of A P sums A[n, m] P[m, J] over the neighbours m of n whose
// row of P holds J.  Finding them is four dependent loads per neighbour (column id, row bounds of P, its aggregate ids,
// then the blocks) -- k_sa_ap spent its 0.94 ms on configuration 5's finest level waiting for those chains.  The plan
// stores the (block of A, block of P) index pairs, so the numeric kernel issues independent loads.
// count != NULL: count pass; else fill pass.
__global__ __launch_bounds__(256) void k_sa_applan(int64_t nap, const int32_t* __restrict__ aprow, const int32_t* __restrict__ apidx,
                                                  const int32_t* __restrict__ ptr, const int32_t* __restrict__ idx,
                                                  const int32_t* __restrict__ pptr, const int32_t* __restrict__ pidx,
                                                  int32_t* __restrict__ count, const int32_t* __restrict__ aplptr,
                                                  int2* __restrict__ aplpair) {
  const int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (q >= nap) return;
  const int64_t i = aprow[q];
  const int32_t J = apidx[q];
  int32_t cnt = 0;
  const int32_t base = count ? 0 : aplptr[q];
  for (int32_t g = ptr[i]; g < ptr[i + 1]; ++g) {
    const int32_t m = idx[g];
    int32_t s2 = pptr[m];
    const int32_t se = pptr[m + 1];
    while (s2 < se && pidx[s2] < J) ++s2;
    if (s2 == se || pidx[s2] != J) continue;
    if (!count) aplpair[base + cnt] = make_int2(g, s2);
    ++cnt;
  }
  if (count) count[q] = cnt;
}

#ifndef PF_AB_AP_U
#define PF_AB_AP_U 5
#endif
template <int DR, int DC>
__global__ __launch_bounds__(256) void k_sa_ap_planned(int64_t nap, const int32_t* __restrict__ aplptr, const int2* __restrict__ aplpair,
                                                      const double* __restrict__ vals, const double* __restrict__ pvals,
                                                      double* __restrict__ apvals) {
  const int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (q >= nap) return;
  const int32_t kb = aplptr[q], ke = aplptr[q + 1];
  double acc[DR][DC];
#pragma unroll
  for (int r = 0; r < DR; ++r)
#pragma unroll
    for (int c = 0; c < DC; ++c) acc[r][c] = 0.0;
  constexpr int U = (DR * DC <= 9) ? PF_AB_AP_U : 1;  // pairs in flight
  for (int32_t k0 = kb; k0 < ke; k0 += U) {
    int2 pr[U];
    double a[U][DR * DR], Pm[U][DR * DC];
#pragma unroll
    for (int u = 0; u < U; ++u) pr[u] = aplpair[min(k0 + u, ke - 1)];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const double* p1 = vals + (int64_t)pr[u].x * DR * DR;
      const double* p2 = pvals + (int64_t)pr[u].y * DR * DC;
#pragma unroll
      for (int k = 0; k < DR * DR; ++k) a[u][k] = p1[k];
#pragma unroll
      for (int k = 0; k < DR * DC; ++k) Pm[u][k] = p2[k];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (k0 + u >= ke) continue;
#pragma unroll
      for (int r = 0; r < DR; ++r)
#pragma unroll
        for (int c = 0; c < DC; ++c) {
          double sacc = 0.0;
#pragma unroll
          for (int tt = 0; tt < DR; ++tt) sacc += a[u][r * DR + tt] * Pm[u][tt * DC + c];
          acc[r][c] += sacc;  // neighbours in ascending order, as in k_sa_ap: the same bits
        }
    }
  }
  double* dst = apvals + q * DR * DC;
#pragma unroll
  for (int r = 0; r < DR; ++r)
#pragma unroll
    for (int c = 0; c < DC; ++c) dst[r * DC + c] = acc[r][c];
}

__global__ void k_sa_aprow_of(int64_t n, const int32_t* __restrict__ apptr, int32_t* __restrict__ aprow) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  for (int32_t k = apptr[i]; k < apptr[i + 1]; ++k) aprow[k] = (int32_t)i;
}

// Gather plan of the second step, once per pattern: one wave per coarse row I, lane k = the row's k-th entry J_k; the
// column of I is walked in ascending fine-node order and every lane with J_k >= I notes, for each fine node whose row of
// A P holds J_k, the pair (slot of P[n, I], slot of (A P)[n, J_k]).  count != NULL: count pass, else fill pass.
__global__ __launch_bounds__(64) void k_sa_gplan(int64_t nc, const int32_t* __restrict__ tptr, const int32_t* __restrict__ tnode,
                                                 const int32_t* __restrict__ tslot, const int32_t* __restrict__ apptr,
                                                 const int32_t* __restrict__ apidx, const int32_t* __restrict__ cptr,
                                                 const int32_t* __restrict__ cidx, int32_t* __restrict__ count,
                                                 const int32_t* __restrict__ gptr, int2* __restrict__ gpair) {
  const int lane = threadIdx.x;
  for (int64_t I = blockIdx.x; I < nc; I += gridDim.x) {
    const int32_t cb = cptr[I], clen = cptr[I + 1] - cb;
    for (int32_t k0 = 0; k0 < clen; k0 += 64) {  // rows longer than a wave: 64 entries at a time
      const int32_t k = k0 + lane;
      const int32_t J = k < clen ? cidx[cb + k] : -1;
      const bool mine = J >= (int32_t)I;
      int32_t cnt = 0;
      const int32_t gb = (!count && mine) ? gptr[cb + k] : 0;
      for (int32_t e = tptr[I]; e < tptr[I + 1]; ++e) {
        const int64_t nd = tnode[e];
        const int32_t ab = apptr[nd], ae = apptr[nd + 1];
        if (!mine) continue;
        int32_t s2 = ab;
        while (s2 < ae && apidx[s2] < J) ++s2;
        if (s2 == ae || apidx[s2] != J) continue;
        if (!count) gpair[gb + cnt] = make_int2(tslot[e], s2);
        ++cnt;
      }
      if (count && k < clen) count[cb + k] = cnt;
    }
  }
}

#ifndef PF_AB_PTAP_U
#define PF_AB_PTAP_U 4
#endif
template <int DR, int DC>
__global__ __launch_bounds__(256) void k_sa_ptap(int64_t nq, const int32_t* __restrict__ gptr, const int2* __restrict__ gpair,
                                                const double* __restrict__ pvals, const double* __restrict__ apvals,
                                                double* __restrict__ cvals) {
  const int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (q >= nq) return;
  const int32_t gb = gptr[q], ge = gptr[q + 1];
  if (gb == ge) return;  // below the diagonal (k_sa_mirror_lower) -- an entry on or above it always has its own aggregate's nodes
  double acc[DC][DC];
#pragma unroll
  for (int c = 0; c < DC; ++c)
#pragma unroll
    for (int c2 = 0; c2 < DC; ++c2) acc[c][c2] = 0.0;
  constexpr int UP = PF_AB_PTAP_U;  // pairs in flight
  for (int32_t g0 = gb; g0 < ge; g0 += UP) {
    int2 pr[UP];
    double Pn[UP][DR * DC], AP[UP][DR * DC];
#pragma unroll
    for (int u = 0; u < UP; ++u) pr[u] = gpair[min(g0 + u, ge - 1)];
#pragma unroll
    for (int u = 0; u < UP; ++u) {
      const double* p1 = pvals + (int64_t)pr[u].x * DR * DC;
      const double* p2 = apvals + (int64_t)pr[u].y * DR * DC;
#pragma unroll
      for (int k = 0; k < DR * DC; ++k) { Pn[u][k] = p1[k]; AP[u][k] = p2[k]; }
    }
#pragma unroll
    for (int u = 0; u < UP; ++u) {
      if (g0 + u >= ge) continue;
#pragma unroll
      for (int c = 0; c < DC; ++c)
#pragma unroll
        for (int c2 = 0; c2 < DC; ++c2) {
          double sacc = 0.0;
#pragma unroll
          for (int r = 0; r < DR; ++r) sacc += Pn[u][r * DC + c] * AP[u][r * DC + c2];
          acc[c][c2] += sacc;
        }
    }
  }
  double* dst = cvals + q * DC * DC;
#pragma unroll
  for (int c = 0; c < DC; ++c)
#pragma unroll
    for (int c2 = 0; c2 < DC; ++c2) dst[c * DC + c2] = acc[c][c2];
}

static int sa_scan(pf_ctx* ctx, int32_t* in, int32_t* out, int64_t n) {
  size_t need = 0;
  PF_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, need, in, out, (int)n, ctx->stream));
  if (need > ctx->scan_tmp_cap) {
    if (ctx->scan_tmp) (void)hipFree(ctx->scan_tmp);
    ctx->scan_tmp = nullptr;
    PF_HIP(hipMalloc(&ctx->scan_tmp, need));
    ctx->scan_tmp_cap = need;
  }
  PF_HIP(hipcub::DeviceScan::ExclusiveSum(ctx->scan_tmp, need, in, out, (int)n, ctx->stream));
  return PF_OK;
}

// Builds Lv.pptr/pidx/tptr/tnode/tslot and the next level's graph (dptr_c/didx_c, also copied to the host vectors for
// the next aggregation).  d_ptr/d_idx: this level's graph on the device, Lv.agg already uploaded.
// Returns PF_E_UNSUPPORTED (nothing allocated) when a row exceeds the caps: the caller then uses sa_pattern_host.
static int sa_pattern_device(pf_ctx* ctx, AmgLevel& Lv, int64_t n, const int32_t* d_ptr, const int32_t* d_idx, int64_t nc,
                             int32_t*& dptr_c, int32_t*& didx_c, std::vector<int32_t>& cptr, std::vector<int32_t>& cidx) {
  hipStream_t st = ctx->stream;
  int32_t *cnt = nullptr, *flag = nullptr, *cur = nullptr;
  const int64_t nmax = std::max(n, nc) + 1;
  PF_HIP(hipMalloc(&cnt, sizeof(int32_t) * nmax));
  PF_HIP(hipMalloc(&cur, sizeof(int32_t) * (nc + 1)));
  PF_HIP(hipMalloc(&flag, sizeof(int32_t)));
  PF_HIP(hipMemsetAsync(flag, 0, sizeof(int32_t), st));
  PF_HIP(hipMemsetAsync(cnt, 0, sizeof(int32_t) * nmax, st));
  int32_t *pptr = nullptr, *pidx = nullptr, *tptr = nullptr, *tnode = nullptr, *tslot = nullptr;
  int rc = PF_OK;
  int32_t total = 0, f = 0;
  auto fail = [&](int code) {
    pf_free(pptr); pf_free(pidx); pf_free(tptr); pf_free(tnode); pf_free(tslot); pf_free(dptr_c); pf_free(didx_c);
    (void)hipFree(cnt); (void)hipFree(cur); (void)hipFree(flag);
    return code;
  };
  const unsigned gn = (unsigned)cdiv(n, 128), gc = (unsigned)cdiv(nc, 128);
  // rows of P
  PF_HIP(hipMalloc(&pptr, sizeof(int32_t) * (n + 1)));
  k_sa_prow<<<gn, 128, 0, st>>>(n, d_ptr, d_idx, Lv.agg, cnt, nullptr, nullptr, flag);
  if ((rc = sa_scan(ctx, cnt, pptr, n + 1))) return fail(rc);
  PF_HIP(hipMemcpyAsync(&total, pptr + n, sizeof(total), hipMemcpyDeviceToHost, st));
  PF_HIP(hipMemcpyAsync(&f, flag, sizeof(f), hipMemcpyDeviceToHost, st));
  PF_HIP(hipStreamSynchronize(st));
  if (f) return fail(PF_E_UNSUPPORTED);
  const int64_t np = total;
  PF_HIP(hipMalloc(&pidx, sizeof(int32_t) * std::max<int64_t>(1, np)));
  k_sa_prow<<<gn, 128, 0, st>>>(n, d_ptr, d_idx, Lv.agg, nullptr, pptr, pidx, flag);
  // transpose
  PF_HIP(hipMalloc(&tptr, sizeof(int32_t) * (nc + 1)));
  PF_HIP(hipMalloc(&tnode, sizeof(int32_t) * std::max<int64_t>(1, np)));
  PF_HIP(hipMalloc(&tslot, sizeof(int32_t) * std::max<int64_t>(1, np)));
  PF_HIP(hipMemsetAsync(cnt, 0, sizeof(int32_t) * nmax, st));
  PF_HIP(hipMemsetAsync(cur, 0, sizeof(int32_t) * (nc + 1), st));
  k_sa_tcount<<<gn, 128, 0, st>>>(n, pptr, pidx, cnt);
  if ((rc = sa_scan(ctx, cnt, tptr, nc + 1))) return fail(rc);
  k_sa_tfill<<<gn, 128, 0, st>>>(n, pptr, pidx, tptr, cur, tnode, tslot);
  k_sa_tsort<<<gc, 128, 0, st>>>(nc, tptr, tnode, tslot);
  // rows of P^T A P
  PF_HIP(hipMalloc(&dptr_c, sizeof(int32_t) * (nc + 1)));
  PF_HIP(hipMemsetAsync(cnt, 0, sizeof(int32_t) * nmax, st));
  k_sa_crow<<<gc, 128, 0, st>>>(nc, tptr, tnode, d_ptr, d_idx, pptr, pidx, cnt, nullptr, nullptr, flag);
  if ((rc = sa_scan(ctx, cnt, dptr_c, nc + 1))) return fail(rc);
  PF_HIP(hipMemcpyAsync(&total, dptr_c + nc, sizeof(total), hipMemcpyDeviceToHost, st));
  PF_HIP(hipMemcpyAsync(&f, flag, sizeof(f), hipMemcpyDeviceToHost, st));
  PF_HIP(hipStreamSynchronize(st));
  if (f) return fail(PF_E_UNSUPPORTED);
  PF_HIP(hipMalloc(&didx_c, sizeof(int32_t) * std::max<int32_t>(1, total)));
  k_sa_crow<<<gc, 128, 0, st>>>(nc, tptr, tnode, d_ptr, d_idx, pptr, pidx, nullptr, dptr_c, didx_c, flag);
  PF_HIP(hipGetLastError());
  cptr.resize(nc + 1);
  cidx.resize(total);
  PF_HIP(hipMemcpyAsync(cptr.data(), dptr_c, sizeof(int32_t) * (nc + 1), hipMemcpyDeviceToHost, st));
  if (total) PF_HIP(hipMemcpyAsync(cidx.data(), didx_c, sizeof(int32_t) * total, hipMemcpyDeviceToHost, st));
  PF_HIP(hipStreamSynchronize(st));
  // rows of A P for the two-step Galerkin product (k_sa_ap / k_sa_ptap); rows beyond the cap leave the one-step kernel in use
  {
    int32_t* apptr = nullptr;
    int32_t* apidx = nullptr;
    PF_HIP(hipMalloc(&apptr, sizeof(int32_t) * (n + 1)));
    PF_HIP(hipMemsetAsync(cnt, 0, sizeof(int32_t) * nmax, st));
    PF_HIP(hipMemsetAsync(flag, 0, sizeof(int32_t), st));
    k_sa_aprow<<<gn, 128, 0, st>>>(n, d_ptr, d_idx, pptr, pidx, cnt, nullptr, nullptr, flag);
    int rc2 = sa_scan(ctx, cnt, apptr, n + 1);
    int32_t tot2 = 0, f2 = 0;
    if (rc2 == PF_OK) {
      PF_HIP(hipMemcpyAsync(&tot2, apptr + n, sizeof(tot2), hipMemcpyDeviceToHost, st));
      PF_HIP(hipMemcpyAsync(&f2, flag, sizeof(f2), hipMemcpyDeviceToHost, st));
      PF_HIP(hipStreamSynchronize(st));
    }
    if (rc2 == PF_OK && !f2 && tot2 > 0) {
      PF_HIP(hipMalloc(&apidx, sizeof(int32_t) * (size_t)tot2));
      k_sa_aprow<<<gn, 128, 0, st>>>(n, d_ptr, d_idx, pptr, pidx, nullptr, apptr, apidx, flag);
      PF_HIP(hipGetLastError());
      Lv.apptr = apptr; Lv.apidx = apidx; Lv.nap = tot2;
      PF_HIP(hipMalloc(&Lv.aprow, sizeof(int32_t) * (size_t)tot2));
      k_sa_aprow_of<<<gn, 128, 0, st>>>(n, apptr, Lv.aprow);
      {  // plan of A P (count, scan, fill); without it k_sa_ap searches the rows of P itself
        int32_t *acnt = nullptr, *aplptr = nullptr;
        const unsigned ga = (unsigned)cdiv(tot2, 256);
        PF_HIP(hipMalloc(&acnt, sizeof(int32_t) * (size_t)(tot2 + 1)));
        PF_HIP(hipMalloc(&aplptr, sizeof(int32_t) * (size_t)(tot2 + 1)));
        PF_HIP(hipMemsetAsync(acnt, 0, sizeof(int32_t) * (size_t)(tot2 + 1), st));
        k_sa_applan<<<ga, 256, 0, st>>>(tot2, Lv.aprow, apidx, d_ptr, d_idx, pptr, pidx, acnt, nullptr, nullptr);
        int rc4 = sa_scan(ctx, acnt, aplptr, (int64_t)tot2 + 1);
        int32_t tot4 = 0;
        if (rc4 == PF_OK) {
          PF_HIP(hipMemcpyAsync(&tot4, aplptr + tot2, sizeof(tot4), hipMemcpyDeviceToHost, st));
          PF_HIP(hipStreamSynchronize(st));
        }
        (void)hipFree(acnt);
        int2* aplpair = nullptr;
        if (rc4 == PF_OK && tot4 > 0 && hipMalloc(&aplpair, sizeof(int2) * (size_t)tot4) == hipSuccess) {
          k_sa_applan<<<ga, 256, 0, st>>>(tot2, Lv.aprow, apidx, d_ptr, d_idx, pptr, pidx, nullptr, aplptr, aplpair);
          PF_HIP(hipGetLastError());
          Lv.aplptr = aplptr; Lv.aplpair = aplpair;
        } else {
          (void)hipGetLastError();
          (void)hipFree(aplptr);
        }
      }
      // gather plan of P^T (A P) over the coarse graph just built (count, scan, fill)
      int32_t *gcnt = nullptr, *gptr = nullptr;
      int2* gpair = nullptr;
      const int64_t nq = total;  // entries of the coarse graph
      PF_HIP(hipMalloc(&gcnt, sizeof(int32_t) * (size_t)(nq + 1)));
      PF_HIP(hipMalloc(&gptr, sizeof(int32_t) * (size_t)(nq + 1)));
      PF_HIP(hipMemsetAsync(gcnt, 0, sizeof(int32_t) * (size_t)(nq + 1), st));
      const unsigned gw = (unsigned)std::min<int64_t>(nc, 1 << 20);
      k_sa_gplan<<<gw, 64, 0, st>>>(nc, tptr, tnode, tslot, apptr, apidx, dptr_c, didx_c, gcnt, nullptr, nullptr);
      int rc3 = sa_scan(ctx, gcnt, gptr, nq + 1);
      int32_t tot3 = 0;
      if (rc3 == PF_OK) {
        PF_HIP(hipMemcpyAsync(&tot3, gptr + nq, sizeof(tot3), hipMemcpyDeviceToHost, st));
        PF_HIP(hipStreamSynchronize(st));
      }
      (void)hipFree(gcnt);
      if (rc3 == PF_OK && tot3 > 0) {
        PF_HIP(hipMalloc(&gpair, sizeof(int2) * (size_t)tot3));
        k_sa_gplan<<<gw, 64, 0, st>>>(nc, tptr, tnode, tslot, apptr, apidx, dptr_c, didx_c, nullptr, gptr, gpair);
        PF_HIP(hipGetLastError());
        Lv.gptr = gptr; Lv.gpair = gpair; Lv.ngpair = tot3;
      } else {
        (void)hipFree(gptr);
      }
    } else {
      (void)hipFree(apptr);
    }
  }
  (void)hipFree(cnt); (void)hipFree(cur); (void)hipFree(flag);
  Lv.pptr = pptr; Lv.pidx = pidx; Lv.tptr = tptr; Lv.tnode = tnode; Lv.tslot = tslot;
  Lv.np = np;
  return PF_OK;
}

// ------------------------------------------------------------------------------------------
// hierarchy management
// ------------------------------------------------------------------------------------------
void pf_amg_free(pf_ctx* ctx) {
  pf_amg* H = ctx->amg;
  if (!H) return;
  for (int l = 0; l < H->nlevels; ++l) {
    AmgLevel& Lv = H->L[l];
    if (Lv.owns_graph) { pf_free(Lv.ptr); pf_free(Lv.idx); }
    if (l > 0) pf_free(Lv.vals);
    pf_free(Lv.vals32);
    if (Lv.owns_xy) pf_free(Lv.xy);
    pf_free(Lv.dinv); pf_free(Lv.dinv2); pf_free(Lv.iso); pf_free(Lv.agg); pf_free(Lv.aptr); pf_free(Lv.amem);
    pf_free(Lv.pptr); pf_free(Lv.pidx); pf_free(Lv.pvals); pf_free(Lv.tptr); pf_free(Lv.tnode); pf_free(Lv.tslot);
    pf_free(Lv.ptvals); pf_free(Lv.pvals32); pf_free(Lv.ptvals32); pf_free(Lv.apptr); pf_free(Lv.apidx); pf_free(Lv.aprow); pf_free(Lv.apvals); pf_free(Lv.gptr); pf_free(Lv.gpair); pf_free(Lv.aplptr); pf_free(Lv.aplpair);
    pf_free(Lv.x); pf_free(Lv.b); pf_free(Lv.t);
  }
  if (H->side) {
    (void)hipStreamSynchronize(H->side);
    (void)hipStreamDestroy(H->side);
    (void)hipEventDestroy(H->ev_coarse);
    (void)hipEventDestroy(H->ev_inv);
  }
  pf_free(H->dense_inv);
  pf_free(H->gersh);
  pf_free(H->powpart);
  pf_free(H->lam);
  delete H;
  ctx->amg = nullptr;
}

template <typename T>
static int to_device(pf_ctx* ctx, const std::vector<T>& h, T*& d) {
  PF_HIP(hipMalloc(&d, std::max<size_t>(1, h.size()) * sizeof(T)));
  if (!h.empty()) PF_HIP(hipMemcpyAsync(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
  return PF_OK;
}

// graph-only part, once per pattern
static int amg_build_hierarchy(pf_ctx* ctx) {
  pf_amg* H = new pf_amg();
  ctx->amg = H;
  t_host_threads = ctx->host_threads;
  const int d0 = ctx->d;
  H->dim = (ctx->ndims == 3 && d0 == 3) ? 3 : 2;
  const int dc = (d0 == 1) ? 1 : (H->dim == 3) ? 6 : 3;
  if (const char* e = pf_exp_env("PF_AMG_SA")) H->sa_mode = atoi(e);
  std::vector<int32_t> ptr(ctx->nnodes + 1), idx(ctx->nblocks);
  PF_HIP(hipMemcpyAsync(ptr.data(), ctx->nptr, ptr.size() * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  PF_HIP(hipMemcpyAsync(idx.data(), ctx->nidx, idx.size() * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  PF_HIP(hipStreamSynchronize(ctx->stream));
  int64_t n = ctx->nnodes;
  const int xdim = ctx->ndims;  // coordinates per node on every level (the rigid-body kernels read them when dr > 1: dim == xdim there)
  std::vector<double> xy((size_t)n * xdim);
  PF_HIP(hipMemcpyAsync(xy.data(), ctx->X, xy.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  PF_HIP(hipStreamSynchronize(ctx->stream));
  H->L[0].n = n; H->L[0].dr = d0; H->L[0].ptr = ctx->nptr; H->L[0].idx = ctx->nidx; H->L[0].nblk = ctx->nblocks;
  H->L[0].xy = ctx->X;
  H->L[0].max_row = ctx->max_row_blocks;
  int l = 0;
  while (true) {
    AmgLevel& Lv = H->L[l];
    const int64_t ndof = Lv.n * Lv.dr;
    PF_HIP(hipMalloc(&Lv.dinv, sizeof(double) * ndof));
    PF_HIP(hipMalloc(&Lv.dinv2, sizeof(double) * ndof));
    PF_HIP(hipMalloc(&Lv.iso, (size_t)ndof));
    PF_HIP(hipMalloc(&Lv.t, sizeof(double) * ndof));
    if (l > 0) {
      PF_HIP(hipMalloc(&Lv.x, sizeof(double) * ndof));
      PF_HIP(hipMalloc(&Lv.b, sizeof(double) * ndof));
      PF_HIP(hipMalloc(&Lv.vals, sizeof(double) * Lv.nblk * Lv.dr * Lv.dr));
    }
    const bool last = (l > 0 && ndof <= AMG_COARSEST_DOFS) || l + 1 >= AMG_MAX_LEVELS;
    if (last) break;
    std::vector<int32_t> agg, aptr, amem, cptr, cidx;
    int64_t nc = 0;
    const bool verbose = ctx->amg_verbose;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_a = now();
    aggregate_host(n, ptr, idx, xy, xdim, agg, nc);
    if (nc >= n || nc == 0) break;  // no coarsening possible
    const double t_b = now();
    const bool sa_level = H->sa_mode == 2 || (H->sa_mode == 1 && !(H->dim == 3 && l == 0));
    coarse_graph_host(n, ptr, idx, agg, nc, aptr, amem, cptr, cidx, sa_level);
    if (verbose) fprintf(stderr, "[pf_amg] level %d: %lld nodes, aggregate %.3f s, tentative graph %.3f s\n", l, (long long)n, t_b - t_a, now() - t_b);
    Lv.nc = nc;
    int rc = PF_OK;
    Lv.sa = sa_level;
    int32_t *dptr_c = nullptr, *didx_c = nullptr;  // the next level's graph when it was built on the device
    if (Lv.sa) {  // the coarse graph is that of P^T A P with the smoothed P
      const double t_c = now();
      rc = to_device(ctx, agg, Lv.agg);
      if (rc) return rc;
      rc = pf_exp_env("PF_AMG_HOST_PATTERNS") ? PF_E_UNSUPPORTED
                                          : sa_pattern_device(ctx, Lv, n, Lv.ptr, Lv.idx, nc, dptr_c, didx_c, cptr, cidx);
      if (rc == PF_E_UNSUPPORTED) {  // a row beyond the device kernels' local arrays: same patterns on host threads
        std::vector<int32_t> pptr, pidx, tptr, tnode, tslot;
        sa_pattern_host(n, ptr, idx, agg, nc, pptr, pidx, tptr, tnode, tslot, cptr, cidx);
        Lv.np = (int64_t)pidx.size();
        rc = to_device(ctx, pptr, Lv.pptr);
        if (rc == PF_OK) rc = to_device(ctx, pidx, Lv.pidx);
        if (rc == PF_OK) rc = to_device(ctx, tptr, Lv.tptr);
        if (rc == PF_OK) rc = to_device(ctx, tnode, Lv.tnode);
        if (rc == PF_OK) rc = to_device(ctx, tslot, Lv.tslot);
      }
      if (rc) return rc;
      if (verbose) fprintf(stderr, "[pf_amg] level %d: smoothed-prolongator and coarse patterns %.3f s (%s)\n", l, now() - t_c, dptr_c ? "device" : "host");
      PF_HIP(hipMalloc(&Lv.pvals, sizeof(double) * Lv.np * Lv.dr * dc));
      if (Lv.apptr) PF_HIP(hipMalloc(&Lv.apvals, sizeof(double) * (size_t)Lv.nap * Lv.dr * dc));
      PF_HIP(hipStreamSynchronize(ctx->stream));  // the staging vectors go out of scope
    }
    Lv.max_crow = 0;
    for (int64_t I = 0; I < nc; ++I) Lv.max_crow = std::max(Lv.max_crow, cptr[I + 1] - cptr[I]);
    H->L[l + 1].max_row = Lv.max_crow;
    if (rc == PF_OK && !Lv.agg) rc = to_device(ctx, agg, Lv.agg);
    if (rc == PF_OK) rc = to_device(ctx, aptr, Lv.aptr);
    if (rc == PF_OK) rc = to_device(ctx, amem, Lv.amem);
    AmgLevel& Nx = H->L[l + 1];
    Nx.n = nc; Nx.dr = dc; Nx.nblk = (int64_t)cidx.size(); Nx.owns_graph = true; Nx.owns_xy = true;
    if (dptr_c) { Nx.ptr = dptr_c; Nx.idx = didx_c; }
    else {
      if (rc == PF_OK) rc = to_device(ctx, cptr, Nx.ptr);
      if (rc == PF_OK) rc = to_device(ctx, cidx, Nx.idx);
    }
    if (rc) return rc;
    PF_HIP(hipMalloc(&Nx.xy, sizeof(double) * xdim * nc));
    k_amg_centroids<<<(unsigned)cdiv(nc, 256), 256, 0, ctx->stream>>>(nc, xdim, Lv.aptr, Lv.amem, Lv.xy, Nx.xy);
    PF_LAUNCH_CHECK();
    xy.resize((size_t)nc * xdim);
    PF_HIP(hipMemcpyAsync(xy.data(), Nx.xy, xy.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    PF_HIP(hipStreamSynchronize(ctx->stream));  // host vectors go out of scope
    ptr.swap(cptr);
    idx.swap(cidx);
    n = nc;
    ++l;
  }
  H->nlevels = l + 1;
  AmgLevel& C = H->L[l];
  H->dense_m = C.n * C.dr;
  if (H->nlevels > 1 && H->dense_m <= AMG_COARSEST_DOFS)
    PF_HIP(hipMalloc(&H->dense_inv, sizeof(double) * H->dense_m * H->dense_m));
  int64_t maxdof = 1;
  for (int k = 0; k < H->nlevels; ++k) maxdof = std::max(maxdof, H->L[k].n * H->L[k].dr);
  PF_HIP(hipMalloc(&H->gersh, sizeof(double) * maxdof));
  PF_HIP(hipMalloc(&H->powpart, sizeof(double) * 2 * AMG_POW_PARTS));
  PF_HIP(hipMalloc(&H->lam, sizeof(double) * (AMG_MAX_LEVELS + 4)));  // + non-positive flag + diagonal scale + positive flag + power-iteration scratch
  return PF_OK;
}

// which: 0 -> first factor (dinv), 1 -> second (dinv2); AMG_SMOOTH0 uses dinv for its first sweep and dinv2 for the second
static int amg_spmv(pf_ctx* ctx, AmgLevel& Lv, int level, const double* x, double* y, int mode = AMG_AX,
                    const double* b = nullptr, int which = 0, double* dot_part = nullptr, int nv = 1) {
  const double* dv = which ? Lv.dinv2 : Lv.dinv;
  if (level == 0) {
    if (mode == AMG_SMOOTH0) return pf_fail(ctx, PF_E_INVALID, "amg_spmv: AMG_SMOOTH0 is a coarse-level mode");
    if (mode == AMG_POW)  // y = (A x) ./ diag, dot_part <- partials of |y|^2 (Lv.dinv holds the raw diagonal during the setup)
      return pf_launch_spmv(ctx, Lv.vals, x, y, dot_part, nullptr, nullptr, Lv.vals32, nullptr, nullptr, 1, Lv.dinv);
    return pf_launch_spmv(ctx, Lv.vals, x, y, dot_part, mode == AMG_SMOOTH ? dv : nullptr, mode == AMG_AX ? nullptr : b,
                          Lv.vals32, nullptr, nullptr, nv);
  }
  const double* d2arg = mode == AMG_POW ? dot_part : Lv.dinv2;  // AMG_POW: the per-wave partials of |y|^2 travel in the dinv2 slot
  const bool small = Lv.n <= AMG_SMALL_LEVEL;
  const unsigned g = (unsigned)std::min<int64_t>(cdiv(Lv.n, small ? 16 : 64), 4096);
  if (nv == 2) {  // two vectors at once: scalar levels only (amg_cycle checks)
#define PF_AS2(M_)                                                                                                  \
  do {                                                                                                              \
    if (Lv.vals32 && small) k_amg_spmv<1, M_, float, 16, 2><<<g, 64, 0, ctx->stream>>>(Lv.n, Lv.ptr, Lv.idx, Lv.vals32, x, y, dv, d2arg, b); \
    else if (Lv.vals32) k_amg_spmv<1, M_, float, 64, 2><<<g, 64, 0, ctx->stream>>>(Lv.n, Lv.ptr, Lv.idx, Lv.vals32, x, y, dv, d2arg, b); \
    else if (small) k_amg_spmv<1, M_, double, 16, 2><<<g, 64, 0, ctx->stream>>>(Lv.n, Lv.ptr, Lv.idx, Lv.vals, x, y, dv, d2arg, b); \
    else k_amg_spmv<1, M_, double, 64, 2><<<g, 64, 0, ctx->stream>>>(Lv.n, Lv.ptr, Lv.idx, Lv.vals, x, y, dv, d2arg, b); \
  } while (0)
    if (mode == AMG_AX) PF_AS2(AMG_AX);
    else if (mode == AMG_SMOOTH) PF_AS2(AMG_SMOOTH);
    else if (mode == AMG_RESID) PF_AS2(AMG_RESID);
    else PF_AS2(AMG_SMOOTH0);
#undef PF_AS2
    PF_LAUNCH_CHECK();
    return PF_OK;
  }
#define PF_AS(D_, M_)                                                                                               \
  do {                                                                                                              \
    if (Lv.vals32 && small) k_amg_spmv<D_, M_, float, 16><<<g, 64, 0, ctx->stream>>>(Lv.n, Lv.ptr, Lv.idx, Lv.vals32, x, y, dv, d2arg, b); \
    else if (Lv.vals32) k_amg_spmv<D_, M_, float><<<g, 64, 0, ctx->stream>>>(Lv.n, Lv.ptr, Lv.idx, Lv.vals32, x, y, dv, d2arg, b); \
    else if (small) k_amg_spmv<D_, M_, double, 16><<<g, 64, 0, ctx->stream>>>(Lv.n, Lv.ptr, Lv.idx, Lv.vals, x, y, dv, d2arg, b); \
    else k_amg_spmv<D_, M_, double><<<g, 64, 0, ctx->stream>>>(Lv.n, Lv.ptr, Lv.idx, Lv.vals, x, y, dv, d2arg, b); \
  } while (0)
#define PF_ASD(M_) do { if (Lv.dr == 1) PF_AS(1, M_); else if (Lv.dr == 6) PF_AS(6, M_); else PF_AS(3, M_); } while (0)
  if (mode == AMG_POW) PF_ASD(AMG_POW);  // dv = the raw diagonal
  else if (mode == AMG_AX) PF_ASD(AMG_AX);
  else if (mode == AMG_SMOOTH) PF_ASD(AMG_SMOOTH);
  else if (mode == AMG_RESID) PF_ASD(AMG_RESID);
  else PF_ASD(AMG_SMOOTH0);
#undef PF_ASD
#undef PF_AS
  PF_LAUNCH_CHECK();
  return PF_OK;
}

__global__ void k_amg_to_f32(int64_t n, const double* __restrict__ a, float* __restrict__ o) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) o[i] = (float)a[i];
}

__global__ void k_round_f32(int64_t n, double* __restrict__ a) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) a[i] = (double)(float)a[i];
}

extern "C" int pf_vals_to_f32(pf_ctx* ctx, int64_t n, const double* vals, float* out) {
  if (!ctx) return pf_fail(nullptr, PF_E_INVALID, "ctx is NULL");
  PF_REQUIRE(n >= 0 && vals && out, "pf_vals_to_f32: NULL argument");
  if (n == 0) return PF_OK;
  k_amg_to_f32<<<(unsigned)std::min<int64_t>(cdiv(n, 256), 8192), 256, 0, ctx->stream>>>(n, vals, out);
  PF_LAUNCH_CHECK();
  return PF_OK;
}

__global__ void k_amg_pow_init(int64_t n, double* __restrict__ v) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) v[i] = 1.0 + 0.37 * (double)((uint32_t)(i * 2654435761u) % 1000u) * 1e-3;  // fixed start vector
}
__global__ void k_amg_pow_div(int64_t n, const double* __restrict__ diag, const double* __restrict__ t,
                              double* __restrict__ v) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) v[i] = t[i] / diag[i];
}
// lambda_max(D^-1 A) by a few power iterations, all on the device: lam2 (one device double) ends up
// holding |D^-1 A v|^2 with |v| = 1.  Lv.dinv still holds the raw diagonal; v is scratch of at least ndof
// doubles.  Unlike a Gershgorin row sum this is invariant under the scaling of individual dofs, which
// matters on the coarse levels where translations and the rotation have different units.
// lam2[0] = (sum of partB) / (sum of partA): one block, fixed order
__global__ __launch_bounds__(256) void k_amg_pow_ratio_parts(double* __restrict__ lam2, const double* __restrict__ partA, int nA,
                                                            const double* __restrict__ partB, int nB) {
  __shared__ double sh[8];
  double a = 0.0, b = 0.0;
  for (int i = threadIdx.x; i < nA; i += 256) a += partA[i];
  for (int i = threadIdx.x; i < nB; i += 256) b += partB[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
  if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6] = a; sh[4 + (threadIdx.x >> 6)] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double den = sh[0] + sh[1] + sh[2] + sh[3], num = sh[4] + sh[5] + sh[6] + sh[7];
    lam2[0] = den > 0.0 ? num / den : 0.0;
  }
}
__global__ void k_amg_pow_ratio(double* __restrict__ num, const double* __restrict__ den) { num[0] = den[0] > 0.0 ? num[0] / den[0] : 0.0; }

// The iterate is NOT normalised between steps: D^-1 A has a spectral radius of a few units, ten steps grow the start
// vector by at most that to the tenth, and only the last two norms are needed.  One launch per step: the product divides
// by the diagonal in its epilogue and leaves per-wave partials of |y|^2 (AMG_POW; round 2 and early round 3: product,
// divide, and for the last two steps dot + finalize -- 26 launches per level, most of them on levels where a launch is
// all latency).  The iterate ping-pongs between v and Lv.t.  3x3-block finest levels (k_spmv3) keep the separate kernels.
static int amg_power_lambda(pf_ctx* ctx, pf_amg* H, AmgLevel& Lv, int level, double* v, double* lam2, double* den, int steps) {
  const int64_t ndof = Lv.n * Lv.dr;
  const unsigned g = (unsigned)cdiv(ndof, 256);
  k_amg_pow_init<<<g, 256, 0, ctx->stream>>>(ndof, v);
  if (level == 0 && ctx->d == 3) {
    for (int it = 0; it < steps; ++it) {
      int rc;
      if (it == steps - 1 && (rc = pf_dot(ctx, ndof, v, v, nullptr, den))) return rc;  // |v_9|^2
      rc = amg_spmv(ctx, Lv, level, v, Lv.t);
      if (rc) return rc;
      k_amg_pow_div<<<g, 256, 0, ctx->stream>>>(ndof, Lv.dinv, Lv.t, v);
    }
    int rc = pf_dot(ctx, ndof, v, v, nullptr, lam2);  // |v_10|^2
    if (rc) return rc;
    k_amg_pow_ratio<<<1, 1, 0, ctx->stream>>>(lam2, den);  // |D^-1 A v_9|^2 / |v_9|^2
    PF_LAUNCH_CHECK();
    return PF_OK;
  }
  double *a = v, *b = Lv.t;
  const int nparts = level == 0 ? pf_spmv_waves(ctx) : (int)std::min<int64_t>(cdiv(Lv.n, Lv.n <= AMG_SMALL_LEVEL ? 16 : 64), 4096);
  for (int it = 0; it < steps; ++it) {
    double* parts = it == steps - 2 ? H->powpart : it == steps - 1 ? H->powpart + AMG_POW_PARTS : nullptr;  // |v_9|^2, |v_10|^2
    if (int rc = amg_spmv(ctx, Lv, level, a, b, AMG_POW, nullptr, 0, parts)) return rc;
    std::swap(a, b);
  }
  k_amg_pow_ratio_parts<<<1, 256, 0, ctx->stream>>>(lam2, H->powpart, nparts, H->powpart + AMG_POW_PARTS, nparts);
  PF_LAUNCH_CHECK();
  return PF_OK;
}

// numeric part, once per matrix: diagonals, Galerkin products, damping factors, coarsest inverse -- all
// queued on the stream with ONE host read at the end (the per-level lambda estimates and the SPD flag).
// Returns PF_E_UNSUPPORTED (without touching ctx->err) when the coupled dofs have diagonal entries of both signs, so
// pf_cg can use Jacobi.  A negative definite K -- the arm mesh, whose triangles all have detJ < 0 --
// is accepted: every step of the cycle is linear in A, and PCG only sees r.z and p.Ap change sign together.
// With reuse != 0 the numeric setup of the previous call is kept if it was made for the same value array (the caller asserts the values are
// unchanged, e.g. the constant Helmholtz filter matrix).
int pf_amg_more_power_steps(pf_ctx* ctx) {
  if (!ctx->amg) return pf_fail(ctx, PF_E_INVALID, "pf_amg_more_power_steps: no hierarchy");
  ctx->amg->pow_iters = std::min(3 * ctx->amg->pow_iters, 90);
  ctx->amg->numeric_ok = false;
  return PF_OK;
}

// HOST-settable number of power-iteration steps of the lambda_max(D^-1 A) estimates (default 10; see pf_cg's rescue):
// takes effect at the next numeric setup of this context's hierarchy, which is built if it does not exist yet.
extern "C" int pf_amg_set_power_steps(pf_ctx* ctx, int steps) {
  if (!ctx) return pf_fail(nullptr, PF_E_INVALID, "ctx is NULL");
  PF_REQUIRE(ctx->d && steps >= 2 && steps <= 200, "pf_amg_set_power_steps: no pattern, or steps outside 2..200");
  PF_HIP(hipSetDevice(ctx->device));
  if (!ctx->amg) {
    const int rc = amg_build_hierarchy(ctx);
    if (rc) { pf_amg_free(ctx); return rc; }
  }
  ctx->amg->pow_iters = steps;
  ctx->amg->numeric_ok = false;
  return PF_OK;
}

int pf_amg_setup(pf_ctx* ctx, const double* vals, int reuse) {
  if (!ctx->amg) {
    int rc = amg_build_hierarchy(ctx);
    if (rc) { pf_amg_free(ctx); return rc; }
  }
  pf_amg* H = ctx->amg;
  if (reuse && H->numeric_ok && H->L[0].vals == vals) return PF_OK;
  // reuse == 2 (PF_PRECOND_AMG_STALE): the setup of an EARLIER matrix stays in use.  The cycle reads only what the hierarchy
  // owns -- the fp32 copy of that matrix on the finest level, the coarse operators, the transfer operators, the damping
  // factors -- never the caller's buffer, which may hold other values by now (or be gone).
  if (reuse == 2 && H->numeric_ok && H->L[0].vals32) return PF_OK;
  H->numeric_ok = false;
  H->L[0].vals = const_cast<double*>(vals);
  if (const char* e = pf_exp_env("PF_AMG_NSMOOTH")) H->nsmooth = std::max(1, atoi(e));
  if (const char* e = pf_exp_env("PF_AMG_TWOSTEP")) H->two_step = atoi(e) != 0;
  if (const char* e = pf_exp_env("PF_AMG_POW")) H->pow_iters = std::max(2, atoi(e));
  if (const char* e = pf_exp_env("PF_AMG_OVER")) H->over = atof(e);
  if (const char* e = pf_exp_env("PF_AMG_OVER_SA")) H->over_sa = atof(e);
  if (const char* e = pf_exp_env("PF_AMG_CHEB")) H->cheb = atof(e);
  if (const char* e = pf_exp_env("PF_AMG_OMEGA_SCALE")) H->omega_scale = atof(e);
  if (const char* e = pf_exp_env("PF_AMG_PSMOOTH")) H->psmooth = atof(e);
  // The cycle's finest-level operator in fp32 (PF_AMG_FP32=0 keeps fp64): the preconditioner only has to be a fixed
  // symmetric positive definite operator, which the V-cycle of the rounded matrix is as well -- all arithmetic and all
  // vectors stay fp64, PCG's own product uses the fp64 matrix.  Five of the six fine SpMVs per iteration then read
  // 18 instead of 34 bytes per 2x2 block (40 instead of 76 per 3x3 block); the coarse operators get the same copy.
  const char* fp32_env = pf_exp_env("PF_AMG_FP32");
  const bool fp32 = !(fp32_env && atoi(fp32_env) == 0);
  {
    AmgLevel& L0 = H->L[0];
    const int64_t nv = L0.nblk * L0.dr * L0.dr;
    if (!fp32) {
      pf_free(L0.vals32);
    } else {
      if (!L0.vals32) PF_HIP(hipMalloc(&L0.vals32, sizeof(float) * (size_t)nv));  // written by k_amg_rows below
    }
  }
  double* bad = H->lam + AMG_MAX_LEVELS;
  PF_HIP(hipMemsetAsync(H->lam, 0, sizeof(double) * (AMG_MAX_LEVELS + 4), ctx->stream));
  for (int l = 0; l < H->nlevels; ++l) {
    AmgLevel& Lv = H->L[l];
    const unsigned gr = (unsigned)cdiv(Lv.n * AMG_ROW_LANES, 256);
    if (Lv.dr == 1) k_amg_rows<1><<<gr, 256, 0, ctx->stream>>>(Lv.n, Lv.ptr, Lv.idx, Lv.vals, Lv.iso, Lv.dinv, bad, H->lam + AMG_MAX_LEVELS + 1, l == 0, Lv.vals32);
    else if (Lv.dr == 2) k_amg_rows<2><<<gr, 256, 0, ctx->stream>>>(Lv.n, Lv.ptr, Lv.idx, Lv.vals, Lv.iso, Lv.dinv, bad, H->lam + AMG_MAX_LEVELS + 1, l == 0, Lv.vals32);
    else if (Lv.dr == 6) k_amg_rows<6><<<gr, 256, 0, ctx->stream>>>(Lv.n, Lv.ptr, Lv.idx, Lv.vals, Lv.iso, Lv.dinv, bad, H->lam + AMG_MAX_LEVELS + 1, l == 0, Lv.vals32);
    else k_amg_rows<3><<<gr, 256, 0, ctx->stream>>>(Lv.n, Lv.ptr, Lv.idx, Lv.vals, Lv.iso, Lv.dinv, bad, H->lam + AMG_MAX_LEVELS + 1, l == 0, Lv.vals32);
    PF_LAUNCH_CHECK();
    // lambda_max(D^-1 A) of this level (power iteration under-estimates slightly, hence the 1.1 in the consumers):
    // damping of the smoother and of the prolongator smoothing.  Lv.dinv still holds the raw diagonal.
    const int64_t ndof = Lv.n * Lv.dr;
    int rc = amg_power_lambda(ctx, H, Lv, l, H->gersh, H->lam + l, H->lam + AMG_MAX_LEVELS + 3, H->pow_iters);
    if (rc) return rc;
    if (l + 1 < H->nlevels) {
      AmgLevel& Nx = H->L[l + 1];
      const unsigned g = (unsigned)cdiv(Lv.nc, 128);
      if (Lv.sa) {
        const unsigned gn = (unsigned)cdiv(Lv.n, 128);
#define PF_SAP(DIM_, DR_, DC_)                                                                                   \
        k_sa_smooth_P<DIM_, DR_, DC_><<<gn, 128, 0, ctx->stream>>>(Lv.n, Lv.ptr, Lv.idx, Lv.vals, Lv.agg, Lv.iso, Lv.xy, \
                                                                   Nx.xy, Lv.dinv, H->lam + l, Lv.pptr, Lv.pidx, Lv.pvals, H->psmooth)
#define PF_SAG(DR_, DC_)                                                                                         \
        do {                                                                                                     \
          const int gw_ = std::max(4, std::min(32, Lv.max_row));                                                 \
          if (Lv.gpair && Lv.apvals && H->two_step) { /* A P first, then P^T (A P) through the gather plan: no atomics */ \
            if (Lv.aplpair)                                                                                       \
              k_sa_ap_planned<DR_, DC_><<<(unsigned)cdiv(Lv.nap, 256), 256, 0, ctx->stream>>>(Lv.nap, Lv.aplptr, Lv.aplpair, Lv.vals, \
                                                                                              Lv.pvals, Lv.apvals); \
            else                                                                                                  \
            k_sa_ap<DR_, DC_><<<(unsigned)cdiv(Lv.nap, 256), 256, 0, ctx->stream>>>(Lv.nap, Lv.aprow, Lv.apidx, Lv.ptr, Lv.idx, \
                                                                                    Lv.vals, Lv.pptr, Lv.pidx, Lv.pvals, Lv.apvals); \
            k_sa_ptap<DR_, DC_><<<(unsigned)cdiv(Nx.nblk, 256), 256, 0, ctx->stream>>>(Nx.nblk, Lv.gptr, Lv.gpair, Lv.pvals, \
                                                                                       Lv.apvals, Nx.vals);     \
            k_sa_mirror_lower<DC_><<<g, 128, 0, ctx->stream>>>(Lv.nc, Nx.ptr, Nx.idx, Nx.vals);                  \
          } else if (Lv.max_crow <= 64) {                                                                        \
            if (Lv.np >= 48 * Lv.nc)                                                                             \
              k_sa_galerkin_wave<DR_, DC_, 4><<<(unsigned)std::min<int64_t>(Lv.nc, 65536), 256, 0, ctx->stream>>>( \
                  Lv.nc, Lv.tptr, Lv.tnode, Lv.tslot, Lv.ptr, Lv.idx, Lv.vals, Lv.pptr, Lv.pidx, Lv.pvals, Nx.ptr, Nx.idx, Nx.vals, gw_); \
            else                                                                                                 \
              k_sa_galerkin_wave<DR_, DC_, 1><<<(unsigned)std::min<int64_t>(Lv.nc, 65536), 64, 0, ctx->stream>>>( \
                  Lv.nc, Lv.tptr, Lv.tnode, Lv.tslot, Lv.ptr, Lv.idx, Lv.vals, Lv.pptr, Lv.pidx, Lv.pvals, Nx.ptr, Nx.idx, Nx.vals, gw_); \
            k_sa_mirror_lower<DC_><<<g, 128, 0, ctx->stream>>>(Lv.nc, Nx.ptr, Nx.idx, Nx.vals);                  \
          } else                                                                                                   \
            k_sa_galerkin<DR_, DC_><<<g, 128, 0, ctx->stream>>>(Lv.nc, Lv.tptr, Lv.tnode, Lv.tslot, Lv.ptr, Lv.idx, Lv.vals, \
                                                                Lv.pptr, Lv.pidx, Lv.pvals, Nx.ptr, Nx.idx, Nx.vals); \
          const int bs_ = DR_ * DC_;                                                                             \
          const size_t nb_ = (size_t)std::max<int64_t>(Lv.np * bs_, 1);                                          \
          if (fp32) {  /* the cycle's transfers read fp32 copies (row order for P, transpose order for P^T) */    \
            if (!Lv.pvals32) PF_HIP(hipMalloc(&Lv.pvals32, sizeof(float) * nb_));                                \
            if (!Lv.ptvals32) PF_HIP(hipMalloc(&Lv.ptvals32, sizeof(float) * nb_));                              \
            k_amg_to_f32<<<(unsigned)std::min<int64_t>(cdiv(Lv.np * bs_, 256), 8192), 256, 0, ctx->stream>>>(Lv.np * bs_, Lv.pvals, Lv.pvals32); \
            k_sa_transpose_vals<float><<<(unsigned)cdiv(Lv.np * bs_, 256), 256, 0, ctx->stream>>>(Lv.np, bs_, Lv.tslot, Lv.pvals, Lv.ptvals32); \
          } else {                                                                                               \
            pf_free(Lv.pvals32); pf_free(Lv.ptvals32);                                                           \
            if (!Lv.ptvals) PF_HIP(hipMalloc(&Lv.ptvals, sizeof(double) * nb_));                                 \
            k_sa_transpose_vals<double><<<(unsigned)cdiv(Lv.np * bs_, 256), 256, 0, ctx->stream>>>(Lv.np, bs_, Lv.tslot, Lv.pvals, Lv.ptvals); \
          }                                                                                                      \
        } while (0)
        if (Lv.dr == 1) { PF_SAP(2, 1, 1); PF_SAG(1, 1); }
        else if (H->dim == 3 && Lv.dr == 3) { PF_SAP(3, 3, 6); PF_SAG(3, 6); }
        else if (H->dim == 3) { PF_SAP(3, 6, 6); PF_SAG(6, 6); }
        else if (Lv.dr == 2) { PF_SAP(2, 2, 3); PF_SAG(2, 3); }
        else { PF_SAP(2, 3, 3); PF_SAG(3, 3); }
#undef PF_SAP
#undef PF_SAG
      } else {
#define PF_GAL(DIM_, DR_, DC_)                                                                                \
        do {                                                                                                  \
          if (Lv.max_crow <= 64) {                                                                            \
            k_amg_galerkin_wave<DIM_, DR_, DC_><<<(unsigned)std::min<int64_t>(Lv.nc, 65536), 64, 0, ctx->stream>>>( \
                Lv.nc, Lv.aptr, Lv.amem, Lv.agg, Lv.ptr, Lv.idx, Lv.vals, Lv.iso, Lv.xy, Nx.xy, Nx.ptr, Nx.idx, Nx.vals); \
            k_sa_mirror_lower<DC_><<<g, 128, 0, ctx->stream>>>(Lv.nc, Nx.ptr, Nx.idx, Nx.vals);               \
          } else                                                                                                \
            k_amg_galerkin<DIM_, DR_, DC_><<<g, 128, 0, ctx->stream>>>(Lv.nc, Lv.aptr, Lv.amem, Lv.agg, Lv.ptr, Lv.idx, Lv.vals, \
                                                                       Lv.iso, Lv.xy, Nx.xy, Nx.ptr, Nx.idx, Nx.vals); \
        } while (0)
        if (Lv.dr == 1) PF_GAL(2, 1, 1);
        else if (H->dim == 3 && Lv.dr == 3) PF_GAL(3, 3, 6);
        else if (H->dim == 3) PF_GAL(3, 6, 6);
        else if (Lv.dr == 2) PF_GAL(2, 2, 3);
        else PF_GAL(2, 3, 3);
#undef PF_GAL
      }
      PF_LAUNCH_CHECK();
      const int64_t nvc = Nx.nblk * Nx.dr * Nx.dr;
      if (fp32) {
        if (!Nx.vals32) PF_HIP(hipMalloc(&Nx.vals32, sizeof(float) * (size_t)std::max<int64_t>(nvc, 1)));  // written by the next level's k_amg_rows
      } else {
        pf_free(Nx.vals32);
      }
    }
    k_amg_scale_dinv<<<(unsigned)cdiv(ndof, 256), 256, 0, ctx->stream>>>(ndof, H->lam + l, H->omega_scale,
                                                                         H->nsmooth == 2 ? H->cheb : 0.0, Lv.dinv, Lv.dinv2, Lv.iso);
    PF_LAUNCH_CHECK();
  }
  if (H->dense_inv) {
    AmgLevel& C = H->L[H->nlevels - 1];
    if (!H->side) {
      PF_HIP(hipStreamCreateWithFlags(&H->side, hipStreamNonBlocking));
      PF_HIP(hipEventCreateWithFlags(&H->ev_coarse, hipEventDisableTiming));
      PF_HIP(hipEventCreateWithFlags(&H->ev_inv, hipEventDisableTiming));
    }
    const char* side_env = pf_exp_env("PF_AMG_SIDE");
    const bool use_side = !(side_env && atoi(side_env) == 0);
    hipStream_t inv_stream = use_side ? H->side : ctx->stream;
    PF_HIP(hipEventRecord(H->ev_coarse, ctx->stream));  // the coarsest operator is complete, the old inverse no longer in use
    PF_HIP(hipStreamWaitEvent(inv_stream, H->ev_coarse, 0));
    PF_HIP(hipMemsetAsync(H->dense_inv, 0, sizeof(double) * H->dense_m * H->dense_m, inv_stream));
    const unsigned g = (unsigned)cdiv(C.n, 128);
    if (C.dr == 1) k_amg_to_dense<1><<<g, 128, 0, inv_stream>>>(C.n, C.ptr, C.idx, C.vals, H->dense_inv, H->dense_m);
    else if (C.dr == 6) k_amg_to_dense<6><<<g, 128, 0, inv_stream>>>(C.n, C.ptr, C.idx, C.vals, H->dense_inv, H->dense_m);
    else k_amg_to_dense<3><<<g, 128, 0, inv_stream>>>(C.n, C.ptr, C.idx, C.vals, H->dense_inv, H->dense_m);
    if (H->dense_m <= 128) k_amg_dense_invert<4><<<1, 1024, 0, inv_stream>>>(H->dense_inv, (int)H->dense_m);
    else k_amg_dense_invert<5><<<1, 1024, 0, inv_stream>>>(H->dense_inv, (int)H->dense_m);
    PF_LAUNCH_CHECK();
    PF_HIP(hipEventRecord(H->ev_inv, inv_stream));
    H->inv_pending = true;
  }
  double h[AMG_MAX_LEVELS + 3];
  PF_HIP(hipMemcpyAsync(h, H->lam, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
  PF_HIP(hipStreamSynchronize(ctx->stream));
  if (ctx->amg_verbose) {
    fprintf(stderr, "[pf_amg] numeric setup: non-positive / positive diagonal seen %g / %g, lambda_max^2 per level:",
            h[AMG_MAX_LEVELS], h[AMG_MAX_LEVELS + 2]);
    for (int l = 0; l < H->nlevels; ++l) fprintf(stderr, " %g", h[l]);
    fprintf(stderr, "\n");
  }
  if (h[AMG_MAX_LEVELS] != 0.0 && h[AMG_MAX_LEVELS + 2] != 0.0) return PF_E_UNSUPPORTED;  // diagonal signs mixed: indefinite
  for (int l = 0; l < H->nlevels; ++l) {
    const double lam = 1.1 * sqrt(h[l]);
    if (!(lam > 0.0) || !(lam < 1e300)) return PF_E_UNSUPPORTED;
    H->L[l].omega = H->omega_scale * 4.0 / (3.0 * lam);
  }
  H->numeric_ok = true;
  return PF_OK;
}

// z = M^-1 r : one V(nsmooth, nsmooth) cycle.  r and z are fine-level vectors of the caller.  The damped
// Jacobi sweeps are fused into the products (x_out = x_in + dinv (b - A x_in)), so the iterate ping-pongs
// between the level's x and t buffers: it changes buffer 2 nsmooth - 1 times per level (always odd), so
// it starts in t and the level's result lands in x (= z on the fine level).
static int amg_restrict_level(pf_ctx* ctx, pf_amg* H, int l, const double* o, double* out, int nv = 1);
static int amg_prolong_level(pf_ctx* ctx, pf_amg* H, int l, const double* xc, double* c, int nv = 1);

// l0 = 0: the whole cycle on (r, z).  l0 = 1: levels >= 1 only, from L[1].b to L[1].x (r, z unused): the part of the
// cycle every rank runs redundantly when the finest level is distributed (dist.py).
// nv = 2: two right-hand sides at once (r, z and the level vectors hold the pair of every dof next to each other); scalar
// hierarchies with smoothed prolongators only.  Every matrix and transfer stream and every launch is shared by the two.
static int amg_cycle(pf_ctx* ctx, const double* r, double* z, int l0, double* rz_part = nullptr, bool* rz_done = nullptr, int nv = 1) {
  if (rz_done) *rz_done = false;
  pf_amg* H = ctx->amg;
  if (!H) return pf_fail(ctx, PF_E_INVALID, "pf_amg_apply: no hierarchy");
  hipStream_t st = ctx->stream;
  const int Lc = H->nlevels - 1;
  if (nv != 1) {
    if (nv != 2) return pf_fail(ctx, PF_E_INVALID, "pf_amg_apply: one or two right-hand sides");
    for (int l = 0; l <= Lc; ++l)
      if (H->L[l].dr != 1 || (l < Lc && !H->L[l].sa)) return PF_E_UNSUPPORTED;  // the caller solves one by one
    if (!(H->dense_inv && Lc > 0)) return PF_E_UNSUPPORTED;
    if (H->nv_cap < nv) {  // the level vectors, once, at twice their length
      for (int l = 0; l <= Lc; ++l) {
        AmgLevel& Lv = H->L[l];
        const size_t bytes = sizeof(double) * (size_t)(Lv.n * Lv.dr) * nv;
        pf_free(Lv.t); PF_HIP(hipMalloc(&Lv.t, bytes));
        if (l > 0) { pf_free(Lv.x); pf_free(Lv.b); PF_HIP(hipMalloc(&Lv.x, bytes)); PF_HIP(hipMalloc(&Lv.b, bytes)); }
      }
      H->nv_cap = nv;
    }
  }
  auto vg = [](int64_t n) { return (unsigned)std::max<int64_t>(1, std::min<int64_t>(cdiv(n, 256), 2048)); };
  double* cur[AMG_MAX_LEVELS];
  double* oth[AMG_MAX_LEVELS];
  int rc;
  // (Round 3 tried the levels below a few thousand nodes in ONE single-workgroup launch -- two sweeps, residual,
  // restriction, dense solve, prolongation, two sweeps, __syncthreads between the phases -- to replace 15-22 dependent
  // launches of ~5.5 us.  One compute unit needs 3-8 us per phase itself (two or three dependent trips to a cold L2 per
  // phase, ~130 GB/s of L2 bandwidth): k_amg_tail took 119 us for the filter's three smallest levels and 53 us for the
  // elasticity hierarchy's two, against 82 - 120 us as launches; solves 6.5 against 5.9 ms and 25.5 against 25.4 ms
  // (profiles/r03_vcycle_tail_ab.jsonl).  Not kept.)
  // down
  for (int l = l0; l < Lc; ++l) {
    AmgLevel& Lv = H->L[l];
    AmgLevel& Nx = H->L[l + 1];
    const double* b = (l == 0) ? r : Lv.b;
    double* c = Lv.t;
    double* o = (l == 0) ? z : Lv.x;
    const int64_t ndof = Lv.n * Lv.dr;
    int done;
    if (l > 0 && H->nsmooth >= 2) {  // first two sweeps in one launch; counts as one buffer change
      if ((rc = amg_spmv(ctx, Lv, l, nullptr, o, AMG_SMOOTH0, b, 0, nullptr, nv))) return rc;
      std::swap(c, o);
      done = 2;
    } else {
      if (nv == 2) k_amg_jacobi0<2><<<vg(2 * ndof), 256, 0, st>>>(ndof, Lv.dinv, b, c);
      else k_amg_jacobi0<1><<<vg(ndof), 256, 0, st>>>(ndof, Lv.dinv, b, c);
      done = 1;
    }
    for (; done < H->nsmooth; ++done) {  // sweep s uses factor s & 1 on the way down ...
      if ((rc = amg_spmv(ctx, Lv, l, c, o, AMG_SMOOTH, b, done & 1, nullptr, nv))) return rc;
      std::swap(c, o);
    }
    if ((rc = amg_spmv(ctx, Lv, l, c, o, AMG_RESID, b, 0, nullptr, nv))) return rc;
    if ((rc = amg_restrict_level(ctx, H, l, o, Nx.b, nv))) return rc;
    PF_LAUNCH_CHECK();
    cur[l] = c;
    oth[l] = o;
  }
  // coarsest
  {
    AmgLevel& C = H->L[Lc];
    const double* b = (Lc == 0) ? r : C.b;
    double* x = (Lc == 0) ? z : C.x;
    const int64_t ndof = C.n * C.dr;
    if (H->dense_inv && Lc > 0) {
      if (H->inv_pending) {  // the inverse of this numeric setup, made on the side stream
        PF_HIP(hipStreamWaitEvent(st, H->ev_inv, 0));
        H->inv_pending = false;
      }
      if (nv == 2) k_amg_dense_apply<2><<<1, 768, 0, st>>>((int)ndof, H->dense_inv, b, x);
      else k_amg_dense_apply<1><<<1, 768, 0, st>>>((int)ndof, H->dense_inv, b, x);
    } else {  // could not coarsen down to the dense size (or a one-level hierarchy): Jacobi sweeps
      double* c = x;
      double* o = C.t;
      k_amg_jacobi0<1><<<vg(ndof), 256, 0, st>>>(ndof, C.dinv, b, c);
      for (int s = 0; s < 4 && Lc > 0; ++s) {  // even count: the result is back in x
        if ((rc = amg_spmv(ctx, C, Lc, c, o, AMG_SMOOTH, b))) return rc;
        std::swap(c, o);
      }
    }
    PF_LAUNCH_CHECK();
  }
  // up
  for (int l = Lc - 1; l >= l0; --l) {
    AmgLevel& Lv = H->L[l];
    AmgLevel& Nx = H->L[l + 1];
    const double* b = (l == 0) ? r : Lv.b;
    double* c = cur[l];
    double* o = oth[l];
    if ((rc = amg_prolong_level(ctx, H, l, Nx.x, c, nv))) return rc;
    for (int s = H->nsmooth - 1; s >= 0; --s) {  // ... and the reverse order on the way up: the cycle stays symmetric
      // the finest level's LAST sweep writes z: the caller's r.z partials come out of its epilogue (b = r)
      double* dp = (l == 0 && s == 0 && rz_part) ? rz_part : nullptr;
      if ((rc = amg_spmv(ctx, Lv, l, c, o, AMG_SMOOTH, b, s & 1, dp, nv))) return rc;
      if (dp && rz_done) *rz_done = true;
      std::swap(c, o);
    }
    PF_LAUNCH_CHECK();
    if (c != ((l == 0) ? z : Lv.x)) return pf_fail(ctx, PF_E_INVALID, "pf_amg_apply: buffer parity");
  }
  return PF_OK;
}

static int amg_restrict_level(pf_ctx* ctx, pf_amg* H, int l, const double* o, double* out, int nv) {
  AmgLevel& Lv = H->L[l];
  AmgLevel& Nx = H->L[l + 1];
  (void)Nx;
    const unsigned g = (unsigned)cdiv(Lv.nc, 128);
    if (nv == 2) {  // scalar smoothed-aggregation levels only (amg_cycle checks)
      const unsigned gs = (unsigned)std::min<int64_t>(cdiv(Lv.nc, 64), 4096), g16 = (unsigned)std::min<int64_t>(cdiv(Lv.nc, 16), 4096);
      if (Lv.np >= 48 * Lv.nc) {
        if (Lv.ptvals32) k_sa_restrict_wave<1, 1, float, 2><<<(unsigned)cdiv(Lv.nc, 4), 256, 0, ctx->stream>>>(Lv.nc, Lv.tptr, Lv.tnode, Lv.ptvals32, o, out);
        else k_sa_restrict_wave<1, 1, double, 2><<<(unsigned)cdiv(Lv.nc, 4), 256, 0, ctx->stream>>>(Lv.nc, Lv.tptr, Lv.tnode, Lv.ptvals, o, out);
      } else if (Lv.nc <= AMG_SMALL_LEVEL) {
        if (Lv.ptvals32) k_sa_restrict_stream<1, 1, float, 16, 2><<<g16, 64, 0, ctx->stream>>>(Lv.nc, Lv.tptr, Lv.tnode, Lv.ptvals32, o, out);
        else k_sa_restrict_stream<1, 1, double, 16, 2><<<g16, 64, 0, ctx->stream>>>(Lv.nc, Lv.tptr, Lv.tnode, Lv.ptvals, o, out);
      } else {
        if (Lv.ptvals32) k_sa_restrict_stream<1, 1, float, 64, 2><<<gs, 64, 0, ctx->stream>>>(Lv.nc, Lv.tptr, Lv.tnode, Lv.ptvals32, o, out);
        else k_sa_restrict_stream<1, 1, double, 64, 2><<<gs, 64, 0, ctx->stream>>>(Lv.nc, Lv.tptr, Lv.tnode, Lv.ptvals, o, out);
      }
      PF_LAUNCH_CHECK();
      return PF_OK;
    }
    if (Lv.sa) {
#define PF_SAR(DR_, DC_)                                                                                    \
  do {                                                                                                      \
    const unsigned gs_ = (unsigned)std::min<int64_t>(cdiv(Lv.nc, 64), 4096);                                 \
    if (Lv.np >= 48 * Lv.nc) { /* long columns, few of them: a wave per coarse row */                        \
      if (Lv.ptvals32) k_sa_restrict_wave<DR_, DC_, float><<<(unsigned)cdiv(Lv.nc, 4), 256, 0, ctx->stream>>>(Lv.nc, Lv.tptr, Lv.tnode, Lv.ptvals32, o, out); \
      else k_sa_restrict_wave<DR_, DC_, double><<<(unsigned)cdiv(Lv.nc, 4), 256, 0, ctx->stream>>>(Lv.nc, Lv.tptr, Lv.tnode, Lv.ptvals, o, out); \
    } else if (Lv.nc <= AMG_SMALL_LEVEL) { /* a small next level: 16 coarse rows per wave, one chunk each */  \
      const unsigned g16_ = (unsigned)std::min<int64_t>(cdiv(Lv.nc, 16), 4096);                               \
      if (Lv.ptvals32) k_sa_restrict_stream<DR_, DC_, float, 16><<<g16_, 64, 0, ctx->stream>>>(Lv.nc, Lv.tptr, Lv.tnode, Lv.ptvals32, o, out); \
      else k_sa_restrict_stream<DR_, DC_, double, 16><<<g16_, 64, 0, ctx->stream>>>(Lv.nc, Lv.tptr, Lv.tnode, Lv.ptvals, o, out); \
    } else {                                                                                                 \
      if (Lv.ptvals32) k_sa_restrict_stream<DR_, DC_, float><<<gs_, 64, 0, ctx->stream>>>(Lv.nc, Lv.tptr, Lv.tnode, Lv.ptvals32, o, out); \
      else k_sa_restrict_stream<DR_, DC_, double><<<gs_, 64, 0, ctx->stream>>>(Lv.nc, Lv.tptr, Lv.tnode, Lv.ptvals, o, out); \
    }                                                                                                        \
  } while (0)
      if (Lv.dr == 1) PF_SAR(1, 1);
      else if (H->dim == 3 && Lv.dr == 3) PF_SAR(3, 6);
      else if (H->dim == 3) PF_SAR(6, 6);
      else if (Lv.dr == 2) PF_SAR(2, 3);
      else PF_SAR(3, 3);
#undef PF_SAR
    } else
    if (Lv.dr == 1) k_amg_restrict<2, 1, 1><<<g, 128, 0, ctx->stream>>>(Lv.nc, Lv.aptr, Lv.amem, Lv.iso, Lv.xy, Nx.xy, o, out);
    else if (H->dim == 3 && Lv.dr == 3) k_amg_restrict<3, 3, 6><<<g, 128, 0, ctx->stream>>>(Lv.nc, Lv.aptr, Lv.amem, Lv.iso, Lv.xy, Nx.xy, o, out);
    else if (H->dim == 3) k_amg_restrict<3, 6, 6><<<g, 128, 0, ctx->stream>>>(Lv.nc, Lv.aptr, Lv.amem, Lv.iso, Lv.xy, Nx.xy, o, out);
    else if (Lv.dr == 2) k_amg_restrict<2, 2, 3><<<g, 128, 0, ctx->stream>>>(Lv.nc, Lv.aptr, Lv.amem, Lv.iso, Lv.xy, Nx.xy, o, out);
    else k_amg_restrict<2, 3, 3><<<g, 128, 0, ctx->stream>>>(Lv.nc, Lv.aptr, Lv.amem, Lv.iso, Lv.xy, Nx.xy, o, out);
  PF_LAUNCH_CHECK();
  return PF_OK;
}

static int amg_prolong_level(pf_ctx* ctx, pf_amg* H, int l, const double* xc, double* c, int nv) {
  AmgLevel& Lv = H->L[l];
  AmgLevel& Nx = H->L[l + 1];
  (void)Nx;
    const unsigned g = (unsigned)cdiv(Lv.n, 256);
    const double over = Lv.sa ? H->over_sa : H->over;
    if (nv == 2) {  // scalar smoothed-aggregation levels only (amg_cycle checks)
      if (Lv.pvals32) k_sa_prolong<1, 1, float, 2><<<g, 256, 0, ctx->stream>>>(Lv.n, Lv.pptr, Lv.pidx, Lv.pvals32, xc, c, over);
      else k_sa_prolong<1, 1, double, 2><<<g, 256, 0, ctx->stream>>>(Lv.n, Lv.pptr, Lv.pidx, Lv.pvals, xc, c, over);
      PF_LAUNCH_CHECK();
      return PF_OK;
    }
    if (Lv.sa) {
#define PF_SAO(DR_, DC_)                                                                                              \
  do {                                                                                                                \
    if (Lv.pvals32) k_sa_prolong<DR_, DC_, float><<<g, 256, 0, ctx->stream>>>(Lv.n, Lv.pptr, Lv.pidx, Lv.pvals32, xc, c, over); \
    else k_sa_prolong<DR_, DC_, double><<<g, 256, 0, ctx->stream>>>(Lv.n, Lv.pptr, Lv.pidx, Lv.pvals, xc, c, over);  \
  } while (0)
      if (Lv.dr == 1) PF_SAO(1, 1);
      else if (H->dim == 3 && Lv.dr == 3) PF_SAO(3, 6);
      else if (H->dim == 3) PF_SAO(6, 6);
      else if (Lv.dr == 2) PF_SAO(2, 3);
      else PF_SAO(3, 3);
#undef PF_SAO
    } else
    if (Lv.dr == 1) k_amg_prolong<2, 1, 1><<<g, 256, 0, ctx->stream>>>(Lv.n, Lv.agg, Lv.iso, Lv.xy, Nx.xy, xc, c, over);
    else if (H->dim == 3 && Lv.dr == 3) k_amg_prolong<3, 3, 6><<<g, 256, 0, ctx->stream>>>(Lv.n, Lv.agg, Lv.iso, Lv.xy, Nx.xy, xc, c, over);
    else if (H->dim == 3) k_amg_prolong<3, 6, 6><<<g, 256, 0, ctx->stream>>>(Lv.n, Lv.agg, Lv.iso, Lv.xy, Nx.xy, xc, c, over);
    else if (Lv.dr == 2) k_amg_prolong<2, 2, 3><<<g, 256, 0, ctx->stream>>>(Lv.n, Lv.agg, Lv.iso, Lv.xy, Nx.xy, xc, c, over);
    else k_amg_prolong<2, 3, 3><<<g, 256, 0, ctx->stream>>>(Lv.n, Lv.agg, Lv.iso, Lv.xy, Nx.xy, xc, c, over);
  PF_LAUNCH_CHECK();
  return PF_OK;
}

int pf_amg_apply(pf_ctx* ctx, const double* r, double* z, double* rz_part, bool* rz_done, int nv) { return amg_cycle(ctx, r, z, 0, rz_part, rz_done, nv); }

int pf_amg_describe(pf_ctx* ctx, int32_t* nlevels, int64_t* nodes, int32_t* dofs_per_node, double* omega, int cap) {
  pf_amg* H = ctx->amg;
  if (!H) return pf_fail(ctx, PF_E_INVALID, "no AMG hierarchy has been built on this context");
  *nlevels = H->nlevels;
  for (int l = 0; l < H->nlevels && l < cap; ++l) {
    nodes[l] = H->L[l].n;
    dofs_per_node[l] = H->L[l].dr;
    omega[l] = H->L[l].omega;
  }
  return PF_OK;
}

extern "C" int pf_amg_info(pf_ctx* ctx, int32_t* nlevels, int64_t* nodes, int32_t* dofs_per_node, double* omega,
                           int cap) {
  if (!ctx) return pf_fail(nullptr, PF_E_INVALID, "ctx is NULL");
  PF_REQUIRE(nlevels && nodes && dofs_per_node && omega && cap > 0, "pf_amg_info: NULL argument");
  return pf_amg_describe(ctx, nlevels, nodes, dofs_per_node, omega, cap);
}

// The V-cycle as a stand-alone operator, for drivers that run their own Krylov loop (dist.py: one local AMG per
// rank inside an additive Schwarz preconditioner).  pf_amg_prepare = numeric setup for `vals` (PF_E_UNSUPPORTED if
// the matrix has a non-positive diagonal entry); pf_amg_vcycle: z = M^-1 r, both of length nrows, z != r.
extern "C" int pf_amg_prepare(pf_ctx* ctx, const double* vals, int reuse) {
  if (!ctx) return pf_fail(nullptr, PF_E_INVALID, "ctx is NULL");
  PF_REQUIRE(ctx->d && vals, "pf_amg_prepare: no pattern / NULL argument");
  PF_HIP(hipSetDevice(ctx->device));
  const int rc = pf_amg_setup(ctx, vals, reuse);
  if (rc == PF_E_UNSUPPORTED) return pf_fail(ctx, rc, "pf_amg_prepare: the matrix has a non-positive diagonal entry");
  return rc;
}

extern "C" int pf_amg_vcycle(pf_ctx* ctx, const double* r, double* z) {
  if (!ctx) return pf_fail(nullptr, PF_E_INVALID, "ctx is NULL");
  PF_REQUIRE(r && z && r != z, "pf_amg_vcycle: NULL argument or z aliases r");
  PF_REQUIRE(ctx->amg && ctx->amg->numeric_ok, "pf_amg_vcycle: call pf_amg_prepare first");
  return pf_amg_apply(ctx, r, z);
}

// ---- the hierarchy's pieces for a DISTRIBUTED finest level (dist.py: every rank holds the same hierarchy, set up
// redundantly from the assembled matrix; smoothing and residuals on the finest level run on the rank's rows only, with
// the interface sums of pf_halo; the transfer to level 1 and everything below run redundantly on every rank) ----------
extern "C" int pf_amg_level0(pf_ctx* ctx, double* dinv1, double* dinv2, int64_t* n1_dofs_host) {
  if (!ctx) return pf_fail(nullptr, PF_E_INVALID, "ctx is NULL");
  PF_REQUIRE(ctx->amg && ctx->amg->numeric_ok, "pf_amg_level0: call pf_amg_prepare first");
  pf_amg* H = ctx->amg;
  if (H->nlevels < 2) return pf_fail(ctx, PF_E_UNSUPPORTED, "pf_amg_level0: the hierarchy has a single level");
  const int64_t n0 = H->L[0].n * H->L[0].dr;
  if (dinv1) PF_HIP(hipMemcpyAsync(dinv1, H->L[0].dinv, sizeof(double) * n0, hipMemcpyDeviceToDevice, ctx->stream));
  if (dinv2) PF_HIP(hipMemcpyAsync(dinv2, H->L[0].dinv2, sizeof(double) * n0, hipMemcpyDeviceToDevice, ctx->stream));
  if (n1_dofs_host) *n1_dofs_host = H->L[1].n * H->L[1].dr;
  return PF_OK;
}

extern "C" int pf_amg_restrict0(pf_ctx* ctx, const double* res, double* b1) {
  if (!ctx) return pf_fail(nullptr, PF_E_INVALID, "ctx is NULL");
  PF_REQUIRE(ctx->amg && ctx->amg->numeric_ok && ctx->amg->nlevels >= 2 && res && b1, "pf_amg_restrict0: no hierarchy / NULL argument");
  return amg_restrict_level(ctx, ctx->amg, 0, res, b1);
}

extern "C" int pf_amg_coarse_solve(pf_ctx* ctx, const double* b1, double* x1) {
  if (!ctx) return pf_fail(nullptr, PF_E_INVALID, "ctx is NULL");
  PF_REQUIRE(ctx->amg && ctx->amg->numeric_ok && ctx->amg->nlevels >= 2 && b1 && x1, "pf_amg_coarse_solve: no hierarchy / NULL argument");
  pf_amg* H = ctx->amg;
  const size_t bytes = sizeof(double) * (size_t)(H->L[1].n * H->L[1].dr);
  PF_HIP(hipMemcpyAsync(H->L[1].b, b1, bytes, hipMemcpyDeviceToDevice, ctx->stream));
  int rc = amg_cycle(ctx, nullptr, nullptr, 1);
  if (rc) return rc;
  PF_HIP(hipMemcpyAsync(x1, H->L[1].x, bytes, hipMemcpyDeviceToDevice, ctx->stream));
  return PF_OK;
}

extern "C" int pf_amg_prolong0(pf_ctx* ctx, const double* x1, double* c) {
  if (!ctx) return pf_fail(nullptr, PF_E_INVALID, "ctx is NULL");
  PF_REQUIRE(ctx->amg && ctx->amg->numeric_ok && ctx->amg->nlevels >= 2 && x1 && c, "pf_amg_prolong0: no hierarchy / NULL argument");
  return amg_prolong_level(ctx, ctx->amg, 0, x1, c);
}

// The same two transfers over the CALLER'S rows of P0 only (dist.py: a rank restricts and prolongs its own nodes
// instead of streaming the whole prolongator):
//   k_sa_prolong_rows:   x_local[i] += over * sum_J P[grow[i], J] e_J          (grow: global node of local node i)
//   k_sa_restrict_rows:  bc[I] = sum over the entries e of column I that lie in rows the caller OWNS of
//                        P[e]^T res_local[ltnode[e]]; (ltptr, ltnode, ltslot) is the transpose pattern of those rows,
//                        ltslot the entry's index in pvals.  Columns without such entries get 0.  One thread per
//                        coarse node, entries in ascending local node order: deterministic.
template <int DR, int DC, typename PT = double>
__global__ void k_sa_prolong_rows(int64_t nl, const int32_t* __restrict__ grow, const int32_t* __restrict__ pptr,
                                  const int32_t* __restrict__ pidx, const PT* __restrict__ pvals,
                                  const double* __restrict__ ec, double* __restrict__ x, double over) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= nl) return;
  const int64_t g = grow[i];
  double s[DR];
#pragma unroll
  for (int r = 0; r < DR; ++r) s[r] = 0.0;
  const int32_t qb = pptr[g], qe = pptr[g + 1];
  for (int32_t q0 = qb; q0 < qe; q0 += 4) {  // four entries in flight, like k_sa_prolong
    int32_t qq[4];
    int64_t JJ[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      qq[u] = min(q0 + u, qe - 1);
      JJ[u] = pidx[qq[u]];
    }
    double Pv[4][DR * DC], ev[4][DC];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const PT* Pn = pvals + (int64_t)qq[u] * DR * DC;
#pragma unroll
      for (int k = 0; k < DR * DC; ++k) Pv[u][k] = (double)Pn[k];
#pragma unroll
      for (int c = 0; c < DC; ++c) ev[u][c] = ec[DC * JJ[u] + c];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (q0 + u >= qe) continue;
#pragma unroll
      for (int r = 0; r < DR; ++r)
#pragma unroll
        for (int c = 0; c < DC; ++c) s[r] += Pv[u][r * DC + c] * ev[u][c];
    }
  }
#pragma unroll
  for (int r = 0; r < DR; ++r) x[DR * i + r] += over * s[r];
}

template <int DR, int DC>
__global__ void k_sa_restrict_rows(int64_t nc, const int32_t* __restrict__ ltptr, const int32_t* __restrict__ ltnode,
                                   const int32_t* __restrict__ ltslot, const double* __restrict__ pvals,
                                   const double* __restrict__ res, double* __restrict__ bc) {
  const int64_t I = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (I >= nc) return;
  double acc[DC];
#pragma unroll
  for (int c = 0; c < DC; ++c) acc[c] = 0.0;
  const int32_t eb = ltptr[I], ee = ltptr[I + 1];
  for (int32_t e0 = eb; e0 < ee; e0 += 4) {  // four entries in flight; ascending order of the adds kept
    int64_t sl[4], nd[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int32_t e = min(e0 + u, ee - 1);
      sl[u] = ltslot[e];
      nd[u] = ltnode[e];
    }
    double Pv[4][DR * DC], rv[4][DR];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const double* Pn = pvals + sl[u] * DR * DC;
#pragma unroll
      for (int k = 0; k < DR * DC; ++k) Pv[u][k] = Pn[k];
#pragma unroll
      for (int r = 0; r < DR; ++r) rv[u][r] = res[DR * nd[u] + r];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (e0 + u >= ee) continue;
#pragma unroll
      for (int r = 0; r < DR; ++r)
#pragma unroll
        for (int c = 0; c < DC; ++c) acc[c] += Pv[u][r * DC + c] * rv[u][r];
    }
  }
#pragma unroll
  for (int c = 0; c < DC; ++c) bc[DC * I + c] = acc[c];
}

static int amg_level0_sa(pf_ctx* ctx, const char* who) {
  PF_REQUIRE(ctx->amg && ctx->amg->numeric_ok && ctx->amg->nlevels >= 2, std::string(who) + ": call pf_amg_prepare first (and the hierarchy needs two levels)");
  PF_REQUIRE(ctx->amg->L[0].sa && ctx->amg->dim == 2 && (ctx->amg->L[0].dr == 1 || ctx->amg->L[0].dr == 2),
             std::string(who) + ": the finest level must carry a stored (smoothed) prolongator: 2-D scalar or plane problems");
  return PF_OK;
}

extern "C" int pf_amg_p0_pattern(pf_ctx* ctx, int64_t* np_host, int64_t* nc_host, int32_t* pptr, int32_t* pidx) {
  if (!ctx) return pf_fail(nullptr, PF_E_INVALID, "ctx is NULL");
  if (int rc = amg_level0_sa(ctx, "pf_amg_p0_pattern")) return rc;
  AmgLevel& L0 = ctx->amg->L[0];
  if (np_host) *np_host = L0.np;
  if (nc_host) *nc_host = L0.nc;
  if (pptr) PF_HIP(hipMemcpyAsync(pptr, L0.pptr, sizeof(int32_t) * (L0.n + 1), hipMemcpyDeviceToDevice, ctx->stream));
  if (pidx) PF_HIP(hipMemcpyAsync(pidx, L0.pidx, sizeof(int32_t) * L0.np, hipMemcpyDeviceToDevice, ctx->stream));
  return PF_OK;
}

extern "C" int pf_amg_prolong0_rows(pf_ctx* ctx, int64_t nl, const int32_t* grow, const double* x1, double* c_local) {
  if (!ctx) return pf_fail(nullptr, PF_E_INVALID, "ctx is NULL");
  if (int rc = amg_level0_sa(ctx, "pf_amg_prolong0_rows")) return rc;
  PF_REQUIRE(nl >= 0 && grow && x1 && c_local, "pf_amg_prolong0_rows: NULL argument");
  pf_amg* H = ctx->amg;
  AmgLevel& L0 = H->L[0];
  if (nl == 0) return PF_OK;
  const unsigned g = (unsigned)cdiv(nl, 256);
  if (L0.pvals32) {  // the cycle's transfers read the fp32 copy, like the single-GPU cycle: the same operator
    if (L0.dr == 1) k_sa_prolong_rows<1, 1, float><<<g, 256, 0, ctx->stream>>>(nl, grow, L0.pptr, L0.pidx, L0.pvals32, x1, c_local, H->over_sa);
    else k_sa_prolong_rows<2, 3, float><<<g, 256, 0, ctx->stream>>>(nl, grow, L0.pptr, L0.pidx, L0.pvals32, x1, c_local, H->over_sa);
  } else {
    if (L0.dr == 1) k_sa_prolong_rows<1, 1, double><<<g, 256, 0, ctx->stream>>>(nl, grow, L0.pptr, L0.pidx, L0.pvals, x1, c_local, H->over_sa);
    else k_sa_prolong_rows<2, 3, double><<<g, 256, 0, ctx->stream>>>(nl, grow, L0.pptr, L0.pidx, L0.pvals, x1, c_local, H->over_sa);
  }
  PF_LAUNCH_CHECK();
  return PF_OK;
}

extern "C" int pf_amg_restrict0_rows(pf_ctx* ctx, const int32_t* ltptr, const int32_t* ltnode, const int32_t* ltslot,
                                     const double* res_local, double* b1) {
  if (!ctx) return pf_fail(nullptr, PF_E_INVALID, "ctx is NULL");
  if (int rc = amg_level0_sa(ctx, "pf_amg_restrict0_rows")) return rc;
  PF_REQUIRE(ltptr && ltnode && ltslot && res_local && b1, "pf_amg_restrict0_rows: NULL argument");
  AmgLevel& L0 = ctx->amg->L[0];
  const unsigned g = (unsigned)cdiv(L0.nc, 256);
  if (L0.dr == 1) k_sa_restrict_rows<1, 1><<<g, 256, 0, ctx->stream>>>(L0.nc, ltptr, ltnode, ltslot, L0.pvals, res_local, b1);
  else k_sa_restrict_rows<2, 3><<<g, 256, 0, ctx->stream>>>(L0.nc, ltptr, ltnode, ltslot, L0.pvals, res_local, b1);
  PF_LAUNCH_CHECK();
  return PF_OK;
}

// The streamed form of the same restriction (r03: the one-thread-per-coarse-node kernel above took 218 us on
// configuration 5's finest level against 78 us for the single-GPU k_sa_restrict_stream): the caller keeps ITS entries of
// P0 in transpose order (pf_amg_p0_gather, once per numeric setup: np_local blocks) and they are streamed like ptvals.
extern "C" int pf_amg_p0_gather(pf_ctx* ctx, int64_t nlt, const int32_t* ltslot, double* ltvals) {
  if (!ctx) return pf_fail(nullptr, PF_E_INVALID, "ctx is NULL");
  if (int rc = amg_level0_sa(ctx, "pf_amg_p0_gather")) return rc;
  PF_REQUIRE(nlt >= 0 && ltslot && ltvals, "pf_amg_p0_gather: NULL argument");
  if (nlt == 0) return PF_OK;
  AmgLevel& L0 = ctx->amg->L[0];
  const int bs = L0.dr * ctx->amg->L[1].dr;
  k_sa_transpose_vals<double><<<(unsigned)cdiv(nlt * bs, 256), 256, 0, ctx->stream>>>(nlt, bs, ltslot, L0.pvals, ltvals);
  if (L0.pvals32) {  // the single-GPU cycle restricts with the fp32-rounded blocks: round the gathered copy the same way
    k_round_f32<<<(unsigned)std::min<int64_t>(cdiv(nlt * bs, 256), 8192), 256, 0, ctx->stream>>>(nlt * bs, ltvals);
  }
  PF_LAUNCH_CHECK();
  return PF_OK;
}

extern "C" int pf_amg_restrict0_stream(pf_ctx* ctx, const int32_t* ltptr, const int32_t* ltnode, const double* ltvals,
                                       const double* res_local, double* b1) {
  if (!ctx) return pf_fail(nullptr, PF_E_INVALID, "ctx is NULL");
  if (int rc = amg_level0_sa(ctx, "pf_amg_restrict0_stream")) return rc;
  PF_REQUIRE(ltptr && ltnode && ltvals && res_local && b1, "pf_amg_restrict0_stream: NULL argument");
  AmgLevel& L0 = ctx->amg->L[0];
  const unsigned g = (unsigned)std::min<int64_t>(cdiv(L0.nc, 64), 4096);
  if (L0.dr == 1) k_sa_restrict_stream<1, 1><<<g, 64, 0, ctx->stream>>>(L0.nc, ltptr, ltnode, ltvals, res_local, b1);
  else k_sa_restrict_stream<2, 3><<<g, 64, 0, ctx->stream>>>(L0.nc, ltptr, ltnode, ltvals, res_local, b1);
  PF_LAUNCH_CHECK();
  return PF_OK;
}

// out = x + d .* (b - w)   (x == NULL: out = d .* b; w == NULL: out = x + d .* b; d == NULL: unit factors)
__global__ __launch_bounds__(256) void k_vec_smooth(int64_t n, const double* __restrict__ x, const double* __restrict__ d,
                                                    const double* __restrict__ b, const double* __restrict__ w,
                                                    double* __restrict__ out) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const double res = b[i] - (w ? w[i] : 0.0);
    out[i] = (x ? x[i] : 0.0) + (d ? d[i] : 1.0) * res;
  }
}

extern "C" int pf_vec_smooth(pf_ctx* ctx, int64_t n, const double* x, const double* d, const double* b, const double* w,
                             double* out) {
  if (!ctx) return pf_fail(nullptr, PF_E_INVALID, "ctx is NULL");
  PF_REQUIRE(n >= 0 && b && out, "pf_vec_smooth: NULL argument");
  if (n == 0) return PF_OK;
  k_vec_smooth<<<(unsigned)std::min<int64_t>(cdiv(n, 256), 2048), 256, 0, ctx->stream>>>(n, x, d, b, w, out);
  PF_LAUNCH_CHECK();
  return PF_OK;
}
