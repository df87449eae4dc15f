// Dirichlet BCs, SpMV and PCG (Jacobi or the AMG V-cycle of pf_amg.hip) on the blocked value layout (gfx950).
//
// Replaces (reference file:line)
//   ModelBase.apply_dirichlet_bcs                pyfem.py:781-835
//   scipy CSR matvec call sites                   pyfem.py:2119 (Helmholtz rhs = R x), inside cg
//   Assembler._solve_linear_system "cg"           pyfem.py:2403-2423  (scipy.sparse.linalg.cg preconditioned with
//       pyamg SA-AMG, :2393-2400; here pf_cg with the smoothed-aggregation V-cycle of pf_amg.hip, or Jacobi)
//
// SpMV design: HBM-bound streaming of the value array (8 B/nnz) plus one 4-byte node index per
// d x d block (so 9 B/nnz at d=2 instead of CSR's 12).  One wave owns 64 consecutive nodes whose
// blocks are ONE contiguous range: lanes stream that range block-per-lane (32 B values + 4 B index,
// fully coalesced, three blocks per lane in flight), gather x[node] as one 16-byte load, and drop
// the d-vector product into an LDS strip; then lane-per-node sums its row's strip entries
// (conflict-free: consecutive rows start 9 x 16 B apart) and writes y coalesced.  Waves are
// persistent (grid-stride over 64-node tiles).
//
// Reductions: NO same-address atomics (the r01 profile: 16 k atomics on one word cost ~190 us per
// CG update).  Every producing block stores one partial; the consuming kernel's blocks each re-sum
// the (<= 4096, L2-resident) partials in a fixed order.  That removes the serialisation, needs no
// extra launch and no fence (the kernel boundary publishes the partials), and makes CG bitwise
// reproducible run to run.
#include <stdlib.h>

#include "pf_common.h"

// SpMV shapes: <D, rows per lane, blocks per lane in flight, LDS strip entries> (10 KB strips -> 16 waves/CU)
#ifndef PF_SPMV1
#define PF_SPMV1 1, 2, 6, 1280
#endif
#ifndef PF_SPMV2
#define PF_SPMV2 2, 1, 3, 640
#endif
#ifndef PF_SPMV1F
#define PF_SPMV1F 1, 2, 9, 1280   // fp32 value copy of scalar matrices (nine in flight: a 128-row tile of a quad mesh in two batches; r03: filter solve -2 % against six)
#endif
#ifndef PF_SPMV2F
#define PF_SPMV2F 2, 1, 9, 640  // fp32 value copy (AMG smoother): 16-byte blocks; nine in flight = a 64-row tile of a quad mesh in one batch (r03: elasticity solve -1.6 % against six)
#endif
static constexpr int SPMV_TILE1 = PF_SPMV_TILE1, SPMV_TILE2 = PF_SPMV_TILE2;  // nodes per wave tile (64 * rows per lane)
// the Morton tile walk (pf_pattern.hip spmv_perm), the 16-bit column offsets and the fused exchange's tile walk
// (pf_halo_plan_fuse) are all built for these tile sizes: a shape macro with another R must not compile
template <int D, int R, int U, int CH>
struct SpmvShape { static constexpr int d = D, tile = 64 * R; };
static_assert(SpmvShape<PF_SPMV1>::tile == SPMV_TILE1 && SpmvShape<PF_SPMV1F>::tile == SPMV_TILE1, "d = 1 SpMV shapes must keep 128-node tiles");
static_assert(SpmvShape<PF_SPMV2>::tile == SPMV_TILE2 && SpmvShape<PF_SPMV2F>::tile == SPMV_TILE2, "d = 2 SpMV shapes must keep 64-node tiles");
static constexpr int SPMV_MAXGRID = 4096;
static constexpr int PF_AMG_RESCUE_AT = 150;  // AMG-PCG iterations after which pf_cg re-estimates lambda_max once (see there)
static constexpr int SPMV2_MAXGRID = 2048;  // k_spmv_nv2 leaves two partials per wave
static_assert(2 * SPMV2_MAXGRID <= PF_NPART && 2 * 64 == PF_SPMV_TILE1, "k_spmv_nv2: partial capacity / tile of the 16-bit offsets");
static constexpr int VEC_BLOCKS = 1024;  // persistent blocks of the vector kernels (256 threads)
static constexpr int NPART = PF_NPART;   // capacity of one partial array

// partial arrays inside ctx->scal (each NPART doubles)
enum { PA_PAP = 0, PA_RZ0 = 1, PA_RZ1 = 2, PA_RR = 3, PA_BB = 4, PA_GAMMA = 5, PA_DELTA = 6, PA_RR2 = 7 };
static_assert(PA_RR2 < PF_NPART_ARRAYS, "partial arrays exceed the ctx scratch");
static_assert(PA_BB == PA_RR + 1, "pf_cg copies RR and BB with one transfer");

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

// butterfly sum: every lane ends with the same bits (a+b == b+a), order fixed => reproducible
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// all threads of a 256-thread block: sum of part[0..n) in a fixed order, same value in every thread
__device__ __forceinline__ double block_sum_partials(const double* __restrict__ part, int n, double* sh) {
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) s += part[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  double t = 0.0;
  const int nw = (blockDim.x + 63) >> 6;
  for (int w = 0; w < nw; ++w) t += sh[w];
  __syncthreads();
  return t;
}

// three partial arrays at once: the loads of all three are in flight together and the block meets twice, not six times
// (block 0 of k_dcg_update: every other block of the launch waits for what it sends).  Same per-array order as
// block_sum_partials, so the same bits.
__device__ __forceinline__ void block_sum_partials3(const double* __restrict__ p0, int n0, const double* __restrict__ p1, int n1,
                                                    const double* __restrict__ p2, int n2, double* sh12, double out[3]) {
  double s0 = 0.0, s1 = 0.0, s2 = 0.0;
  for (int i = threadIdx.x; i < n0; i += blockDim.x) s0 += p0[i];
  for (int i = threadIdx.x; i < n1; i += blockDim.x) s1 += p1[i];
  for (int i = threadIdx.x; i < n2; i += blockDim.x) s2 += p2[i];
  s0 = wave_sum(s0); s1 = wave_sum(s1); s2 = wave_sum(s2);
  const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  if ((threadIdx.x & 63) == 0) { sh12[w] = s0; sh12[4 + w] = s1; sh12[8 + w] = s2; }
  __syncthreads();
  out[0] = out[1] = out[2] = 0.0;
  for (int k = 0; k < nw; ++k) { out[0] += sh12[k]; out[1] += sh12[4 + k]; out[2] += sh12[8 + k]; }
  __syncthreads();
}

// block-level sum of per-thread values -> out[blockIdx.x]
__device__ __forceinline__ void block_store_partial(double s, double* __restrict__ out, double* sh) {
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    const int nw = (blockDim.x + 63) >> 6;
    for (int w = 0; w < nw; ++w) t += sh[w];
    out[blockIdx.x] = t;
  }
  __syncthreads();
}

// ------------------------------------------------------------------------------------------
// y = A x ; optionally dot_part[blockIdx.x] = sum over this wave's rows of x_i y_i (b_i y_i with a smoother epilogue)
//   smoother epilogues for the AMG cycle (no extra pass over the vectors): with sm_b and sm_dinv
//   y = x + dinv .* (b - A x) (one damped-Jacobi sweep, y != x), with sm_b alone y = b - A x.
//   R = node rows per lane (tile = 64*R nodes), U = blocks per lane in flight, CH = LDS strip entries.
//   d=1 moves a third of the bytes of d=2 per node, so it takes R=2 / U=6 to amortise the per-tile
//   overhead (r01: 50 % of peak at R=1/U=3).
//   VT = float reads an fp32 copy of the values (arithmetic stays fp64): the smoother sweeps of the AMG cycle, whose
//   operator only has to be a fixed symmetric one.
// ------------------------------------------------------------------------------------------
// The value stream is read once per product while x is gathered again and again: loading the values NON-TEMPORALLY
// leaves the cache to x.  Scalar matrices (d = 1): 0.0968 -> 0.0822 ms on the Helmholtz 2048^2 matrix.  2x2 blocks: with
// a lane per 32-byte block each of its two 16-byte loads touches HALF of 16 lines and the cache joins them, so
// non-temporal loads cost 11 % (0.261 -> 0.291 ms); with one block ROW per lane and access every instruction covers
// whole lines and they gain 10 % (0.242 -> 0.218 ms = 6.6 TB/s of the bytes the layout moves; r02, tools/spmv_ab.py).
template <bool NT, typename T>
__device__ __forceinline__ T spmv_ld(const T* p) {
  if constexpr (NT) return __builtin_nontemporal_load(p);
  else return *p;
}
// NTV (template parameter of k_spmv): the value stream is read non-temporally.  Right for matrices that do not fit the
// 256 MB Infinity Cache (the 4M-quad matrix: 0.242 -> 0.218 ms), WRONG for those that do: one rank's share of the metric
// mesh at N = 8 (161 MB of values) runs at 28.6 us with plain loads against 31.3 us with non-temporal ones -- the stream
// of the previous product is still in the memory-side cache and the hint throws that away; at the N = 4 share (322 MB) the
// hint wins again (58.0 against 62.6 us; r03, tools/strip_cg_ab.py).  pf_launch_spmv decides by the size of the stream --
// except inside an AMG-preconditioned solve (ctx->multi_stream), where an iteration streams K, its fp32 copy and the
// transfer operators in turn and nothing survives in that cache: the filter's 113 MB fp32 stream ran 3 % slower plain.
static constexpr int64_t SPMV_NT_MIN_BYTES = (int64_t)224 << 20;
#ifndef PF_AB_SPMV_XCD
#define PF_AB_SPMV_XCD 1  // the XCD-aware block renumbering on banded numberings too: 0.2153 -> 0.2081 ms at d = 2 (0: only with tperm)
#endif
#ifndef PF_AB_SPMV_NT32
#define PF_AB_SPMV_NT32 1  // non-temporal loads of the fp32 value copy (AMG smoother products): -3 % on a configuration-5 solve with six blocks in flight (0: plain)
#endif
#ifndef PF_AB_SPMV_HALF
#define PF_AB_SPMV_HALF 1  // one 16-byte block row per lane and access in the fp64 2x2 product (0: a lane per block)
#endif
#ifndef PF_AB_SPMV_NT2
#define PF_AB_SPMV_NT2 1  // non-temporal loads of the fp64 2x2 value stream (only pays with whole-line accesses: PF_AB_SPMV_HALF)
#endif
// SEND (the product over an element-block partition, one process per GPU): `snd` names the interface nodes and, per
// node, the peers that share it and its place in the message to each.  The tiles that hold interface nodes come first in
// tperm; for their interface rows the RAW product goes to y (no smoother epilogue: the row is not complete yet) and,
// from the same registers, straight into the neighbours' windows over xGMI; the wave that finishes the last interface
// tile raises this rank's row flag at every neighbour.  The interior tiles follow while the rows travel.  The kernel that
// consumes y (k_dcg_update, k_halo_recv_apply) sums the interface rows in ascending rank order.
struct pf_spmv_nosend {};
template <int D, int R, int U, int CH, bool IDX16 = false, typename VT = double, bool SEND = false, bool NTV = true>
__global__ __launch_bounds__(64) void k_spmv(int64_t nnodes, const int32_t* __restrict__ nptr,
                                             const int32_t* __restrict__ nidx, const VT* __restrict__ vals,
                                             const double* __restrict__ x, double* __restrict__ y,
                                             double* __restrict__ dot_part, const double* __restrict__ sm_dinv,
                                             const double* __restrict__ sm_b, const int32_t* __restrict__ tperm,
                                             const double* __restrict__ pw_diag,
                                             const std::conditional_t<SEND, pf_spmv_send, pf_spmv_nosend> snd) {
  using V = typename Vec<D>::T;
  constexpr int TILE = 64 * R;
  __shared__ V prod[CH];
  if constexpr (SEND) {
    if (snd.stop && *snd.stop != 0.0) return;  // converged (pf_dist_cg): nothing is sent, and no consumer waits for it
  }
  const int lane = threadIdx.x;
  double dacc = 0.0;
  const int64_t ntiles = (nnodes + TILE - 1) / TILE;
  // Workgroups are dealt to the 8 XCDs round-robin: renumber the blocks so that one XCD's waves take a contiguous eighth
  // of every grid-wide row of tiles -- the x gathers of the waves that share an L2 then overlap.  On scattered numberings
  // the tiles are additionally walked along a Morton curve (tperm, pf_pattern_build): 0.0748 -> 0.0708 ms on the refined
  // arm.  The rows of a tile and what is written for them do not change.
  const int64_t vblk = ((tperm || PF_AB_SPMV_XCD) && (gridDim.x & 7) == 0) ? (int64_t)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3) : blockIdx.x;
  for (int64_t tslot = vblk; tslot < ntiles; tslot += gridDim.x) {
    const int64_t tile = tperm ? tperm[tslot] : tslot;
    int32_t rb[R], re[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int64_t n = tile * TILE + r * 64 + lane;
      const bool valid = n < nnodes;
      rb[r] = valid ? nptr[n] : 0;
      re[r] = valid ? nptr[n + 1] : 0;
    }
    const int32_t base = __shfl(rb[0], 0);
    const int64_t nlast = min(nnodes, (tile + 1) * TILE);
    const int32_t end = nptr[nlast];  // uniform
    V acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if constexpr (D == 1) acc[r] = 0.0; else acc[r] = make_double2(0.0, 0.0);
    }
    for (int32_t c0 = base; c0 < end; c0 += CH) {
      const int32_t c1 = min(end, c0 + CH);
      // phase A: stream blocks [c0, c1), U per lane in flight
      if constexpr (D == 2 && sizeof(VT) == 8 && PF_AB_SPMV_HALF) {
        // fp64 2x2 blocks, one block ROW (16 bytes) per lane and access: every load instruction covers 1 KiB of whole
        // lines (a lane per 32-byte block makes each of its two 16-byte loads touch half of 16 lines and leaves the
        // joining to the cache), which is also what lets the value stream go non-temporal.
        typedef double f64x2 __attribute__((ext_vector_type(2)));
        constexpr int UH = 2 * U;
        double* prodh = reinterpret_cast<double*>(prod);
        const int32_t h1 = 2 * c1;
        for (int32_t hb0 = 2 * c0 + lane; hb0 < h1; hb0 += 64 * UH) {
          int32_t hb[UH], m[UH];
#pragma unroll
          for (int u = 0; u < UH; ++u) {
            hb[u] = min(hb0 + 64 * u, h1 - 1);  // clamped: loads stay in range, result discarded
            if constexpr (IDX16) m[u] = (int32_t)(tile * TILE) + reinterpret_cast<const int16_t*>(nidx)[hb[u] >> 1];
            else m[u] = nidx[hb[u] >> 1];
          }
          f64x2 kv[UH];
          double2 xv[UH];
#pragma unroll
          for (int u = 0; u < UH; ++u) {
            const f64x2* vp = reinterpret_cast<const f64x2*>(vals) + hb[u];
            kv[u] = (NTV && PF_AB_SPMV_NT2) ? __builtin_nontemporal_load(vp) : *vp;
          }
#pragma unroll
          for (int u = 0; u < UH; ++u) xv[u] = *reinterpret_cast<const double2*>(x + 2 * (int64_t)m[u]);
#pragma unroll
          for (int u = 0; u < UH; ++u)
            if (hb0 + 64 * u < h1) prodh[hb[u] - 2 * c0] = kv[u].x * xv[u].x + kv[u].y * xv[u].y;
        }
      } else
      for (int32_t g0 = c0 + lane; g0 < c1; g0 += 64 * U) {
        int32_t g[U], m[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          g[u] = min(g0 + 64 * u, c1 - 1);  // clamped: loads stay in range, result discarded
          if constexpr (IDX16)  // 2-byte offsets from the tile's first node: 10 instead of 12 bytes per nonzero
            m[u] = (int32_t)(tile * TILE) + reinterpret_cast<const int16_t*>(nidx)[g[u]];
          else
            m[u] = nidx[g[u]];
#ifdef PF_ABLATE_GATHER  // timing experiment (wrong results): every lane gathers its own tile's node `lane` -- coalesced, index stream still read
          m[u] = (int32_t)min((int64_t)nnodes - 1, tile * TILE + lane + (m[u] & 0));
#endif
        }
        if constexpr (D == 1) {
          double kv[U], xv[U];
#pragma unroll
          for (int u = 0; u < U; ++u) kv[u] = (double)spmv_ld<NTV>(vals + g[u]);
#pragma unroll
          for (int u = 0; u < U; ++u) xv[u] = x[m[u]];
#pragma unroll
          for (int u = 0; u < U; ++u)
            if (g0 + 64 * u < c1) prod[g[u] - c0] = kv[u] * xv[u];
        } else {
          double2 k0[U], k1[U], xv[U];
#pragma unroll
          for (int u = 0; u < U; ++u) {
            if constexpr (sizeof(VT) == 8) {
              typedef double f64x2 __attribute__((ext_vector_type(2)));
              const f64x2* vp = reinterpret_cast<const f64x2*>(vals + 4 * (int64_t)g[u]);
              const f64x2 a0 = (NTV && PF_AB_SPMV_NT2) ? __builtin_nontemporal_load(vp) : vp[0];
              const f64x2 a1 = (NTV && PF_AB_SPMV_NT2) ? __builtin_nontemporal_load(vp + 1) : vp[1];
              k0[u] = make_double2(a0.x, a0.y);
              k1[u] = make_double2(a1.x, a1.y);
            } else {  // fp32 copy of the operator (the AMG cycle's smoother): one 16-byte load per block
              typedef float f32x4 __attribute__((ext_vector_type(4)));
              const f32x4* vp = reinterpret_cast<const f32x4*>(vals + 4 * (int64_t)g[u]);
              const f32x4 f = (NTV && PF_AB_SPMV_NT32) ? __builtin_nontemporal_load(vp) : vp[0];
              k0[u] = make_double2((double)f.x, (double)f.y);
              k1[u] = make_double2((double)f.z, (double)f.w);
            }
          }
#pragma unroll
          for (int u = 0; u < U; ++u) xv[u] = *reinterpret_cast<const double2*>(x + 2 * (int64_t)m[u]);
#pragma unroll
          for (int u = 0; u < U; ++u)
            if (g0 + 64 * u < c1)
              prod[g[u] - c0] = make_double2(k0[u].x * xv[u].x + k0[u].y * xv[u].y,
                                             k1[u].x * xv[u].x + k1[u].y * xv[u].y);
        }
      }
      __syncthreads();
      // phase B: lane-per-node partial row sums over this chunk
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int32_t lo = max(rb[r], c0), hi = min(re[r], c1);
        for (int32_t gq = lo; gq < hi; ++gq) {
          const V pv = prod[gq - c0];
          if constexpr (D == 1) acc[r] += pv; else { acc[r].x += pv.x; acc[r].y += pv.y; }
        }
      }
      __syncthreads();
    }
    bool iftile = false;
    if constexpr (SEND) iftile = tslot < snd.n_iftiles;  // wave-uniform
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int64_t n = tile * TILE + r * 64 + lane;
      if (n < nnodes) {
        int32_t ifs = -1;
        if constexpr (SEND) {
          if (iftile) ifs = snd.ifslot[n];
          if (ifs >= 0) {  // an interface row: its partial sum goes to the sharing ranks as it is
            for (int32_t k = snd.sh_ptr[ifs]; k < snd.sh_ptr[ifs + 1]; ++k) {
              double* dst = snd.peer_win[snd.sh_peer[k]] + snd.slot_off + snd.sh_pos[k];
              if constexpr (D == 1) pf_st_sys_f64(dst, acc[r]);
              else { pf_st_sys_f64(dst, acc[r].x); pf_st_sys_f64(dst + 1, acc[r].y); }
            }
          }
        }
        if (sm_b && ifs < 0) {
          if constexpr (D == 1) {
            const double res = sm_b[n] - acc[r];
            acc[r] = sm_dinv ? x[n] + sm_dinv[n] * res : res;
          } else {
            const double2 bv = *reinterpret_cast<const double2*>(sm_b + 2 * n);
            double2 res = make_double2(bv.x - acc[r].x, bv.y - acc[r].y);
            if (sm_dinv) {
              const double2 xv = *reinterpret_cast<const double2*>(x + 2 * n);
              const double2 dv = *reinterpret_cast<const double2*>(sm_dinv + 2 * n);
              res = make_double2(xv.x + dv.x * res.x, xv.y + dv.y * res.y);
            }
            acc[r] = res;
          }
        }
        // partials: x.y for a plain product (p.Ap of CG); with a smoother epilogue b.y -- the r.z of AMG-PCG when this is
        // the cycle's last sweep (b = r, y = z): no separate dot kernel, no extra pass over r and z.  pw_diag (a step of
        // the power iteration on D^-1 A, AMG numeric setup): y = (A x) ./ diag and the partials are those of |y|^2.
        if constexpr (D == 1) {
          if (pw_diag) acc[r] = acc[r] / pw_diag[n];
          y[n] = acc[r];
          if (dot_part) dacc += (pw_diag ? acc[r] : sm_b ? sm_b[n] : x[n]) * acc[r];
        } else {
          if (pw_diag) {
            const double2 dg = *reinterpret_cast<const double2*>(pw_diag + 2 * n);
            acc[r] = make_double2(acc[r].x / dg.x, acc[r].y / dg.y);
          }
          *reinterpret_cast<double2*>(y + 2 * n) = acc[r];
          if (dot_part) {
            const double2 xv = pw_diag ? acc[r] : *reinterpret_cast<const double2*>((sm_b ? sm_b : x) + 2 * n);
            dacc += xv.x * acc[r].x + xv.y * acc[r].y;
          }
        }
      }
    }
    if constexpr (SEND) {
      if (iftile) {
        pf_release_sys();  // this wave's rows have arrived before it is counted (write-back + wait: no cache is invalidated)
        if (lane == 0) {
          const unsigned long long prev = __hip_atomic_fetch_add(snd.rows_done, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (prev + 1 == (unsigned long long)snd.n_iftiles) {  // the last interface tile of this product: tell the neighbours
            __hip_atomic_store(snd.rows_done, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // for the next launch (stream order)
            for (int k = 0; k < snd.nnbr; ++k)
              __hip_atomic_store(reinterpret_cast<uint64_t*>(snd.peer_win[snd.nbr[k]]) + snd.rflag_off, snd.rseq, __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_SYSTEM);
          }
        }
      }
    }
  }
  if (dot_part) {
    dacc = wave_sum(dacc);
    if (lane == 0) dot_part[blockIdx.x] = dacc;
  }
}

// ------------------------------------------------------------------------------------------
// Scalar matrix times TWO vectors (x, y, sm_b interleaved per node: double2 per node), for the pair of gradient filters of
// a topology iteration (pf_cg2): the value and index streams -- two thirds of a scalar product's bytes -- are read once
// for both.  Same tiling as k_spmv<PF_SPMV1>, same epilogues (sm_dinv is per node), dot_part[2 * wave + v].  The value
// stream is always read non-temporally (this only runs inside AMG-preconditioned solves, see SPMV_NT_MIN_BYTES).
// ------------------------------------------------------------------------------------------
template <int R, int U, int CH, bool IDX16, typename VT>
__global__ __launch_bounds__(64) void k_spmv_nv2(int64_t nnodes, const int32_t* __restrict__ nptr,
                                                 const int32_t* __restrict__ nidx, const VT* __restrict__ vals,
                                                 const double* __restrict__ x, double* __restrict__ y,
                                                 double* __restrict__ dot_part, const double* __restrict__ sm_dinv,
                                                 const double* __restrict__ sm_b, const int32_t* __restrict__ tperm) {
  constexpr int TILE = 64 * R;
  __shared__ double2 prod[CH];
  const int lane = threadIdx.x;
  double dacc0 = 0.0, dacc1 = 0.0;
  const int64_t ntiles = (nnodes + TILE - 1) / TILE;
  const int64_t vblk = ((tperm || PF_AB_SPMV_XCD) && (gridDim.x & 7) == 0) ? (int64_t)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3) : blockIdx.x;
  for (int64_t tslot = vblk; tslot < ntiles; tslot += gridDim.x) {
    const int64_t tile = tperm ? tperm[tslot] : tslot;
    int32_t rb[R], re[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int64_t n = tile * TILE + r * 64 + lane;
      const bool valid = n < nnodes;
      rb[r] = valid ? nptr[n] : 0;
      re[r] = valid ? nptr[n + 1] : 0;
    }
    const int32_t base = __shfl(rb[0], 0);
    const int64_t nlast = min(nnodes, (tile + 1) * TILE);
    const int32_t end = nptr[nlast];  // uniform
    double2 acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = make_double2(0.0, 0.0);
    for (int32_t c0 = base; c0 < end; c0 += CH) {
      const int32_t c1 = min(end, c0 + CH);
      for (int32_t g0 = c0 + lane; g0 < c1; g0 += 64 * U) {
        int32_t g[U], m[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          g[u] = min(g0 + 64 * u, c1 - 1);  // clamped: loads stay in range, result discarded
          if constexpr (IDX16) m[u] = (int32_t)(tile * TILE) + reinterpret_cast<const int16_t*>(nidx)[g[u]];
          else m[u] = nidx[g[u]];
        }
        double kv[U];
        double2 xv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) kv[u] = (double)__builtin_nontemporal_load(vals + g[u]);
#pragma unroll
        for (int u = 0; u < U; ++u) xv[u] = *reinterpret_cast<const double2*>(x + 2 * (int64_t)m[u]);
#pragma unroll
        for (int u = 0; u < U; ++u)
          if (g0 + 64 * u < c1) prod[g[u] - c0] = make_double2(kv[u] * xv[u].x, kv[u] * xv[u].y);
      }
      __syncthreads();
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int32_t lo = max(rb[r], c0), hi = min(re[r], c1);
        for (int32_t gq = lo; gq < hi; ++gq) {
          const double2 pv = prod[gq - c0];
          acc[r].x += pv.x; acc[r].y += pv.y;
        }
      }
      __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int64_t n = tile * TILE + r * 64 + lane;
      if (n < nnodes) {
        if (sm_b) {
          const double2 bv = *reinterpret_cast<const double2*>(sm_b + 2 * n);
          double2 res = make_double2(bv.x - acc[r].x, bv.y - acc[r].y);
          if (sm_dinv) {
            const double2 xv = *reinterpret_cast<const double2*>(x + 2 * n);
            const double dv = sm_dinv[n];
            res = make_double2(xv.x + dv * res.x, xv.y + dv * res.y);
          }
          acc[r] = res;
        }
        *reinterpret_cast<double2*>(y + 2 * n) = acc[r];
        if (dot_part) {  // x.y per vector (b.y with a smoother epilogue), as in k_spmv
          const double2 xv = *reinterpret_cast<const double2*>((sm_b ? sm_b : x) + 2 * n);
          dacc0 += xv.x * acc[r].x;
          dacc1 += xv.y * acc[r].y;
        }
      }
    }
  }
  if (dot_part) {
    dacc0 = wave_sum(dacc0);
    dacc1 = wave_sum(dacc1);
    if (lane == 0) { dot_part[2 * blockIdx.x] = dacc0; dot_part[2 * blockIdx.x + 1] = dacc1; }
  }
}

// ------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------
// d = 3 (3-D elasticity): 3x3 blocks, 72 B of values + 4 B of index per block.  One persistent wave per
// 64-node tile.  The tile's contiguous block range is streamed 64 blocks at a time: the 4.6 KB of values
// are copied to an LDS staging strip with nine fully coalesced 512-byte wave loads (a lane reading "its"
// block directly would make every load instruction touch 36 different lines), each lane then reads its
// block back at a stride of 9 doubles (odd: conflict-free), multiplies by the gathered x and parks the
// 3-vector product in the strip; lane-per-node sums its row.  One wave per block: LDS operations execute in
// program order, so the staging needs no barrier.  Same epilogues as k_spmv.
// ------------------------------------------------------------------------------------------
template <typename VT>
__global__ __launch_bounds__(64) void k_spmv3(int64_t nnodes, const int32_t* __restrict__ nptr,
                                              const int32_t* __restrict__ nidx, const VT* __restrict__ vals,
                                              const double* __restrict__ x, double* __restrict__ y,
                                              double* __restrict__ dot_part, const double* __restrict__ sm_dinv,
                                              const double* __restrict__ sm_b) {
  constexpr int CH = 384, U = 3;
  __shared__ double prod[3 * CH];
  __shared__ double stage[64 * 9];
  const int lane = threadIdx.x;
  double dacc = 0.0;
  const int64_t ntiles = (nnodes + 63) / 64;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t n = tile * 64 + lane;
    const bool valid = n < nnodes;
    const int32_t rb = valid ? nptr[n] : 0, re = valid ? nptr[n + 1] : 0;
    const int32_t base = __shfl(rb, 0);
    const int32_t end = nptr[min(nnodes, (tile + 1) * 64)];
    double acc[3] = {0.0, 0.0, 0.0};
    for (int32_t c0 = base; c0 < end; c0 += CH) {
      const int32_t c1 = min(end, c0 + CH);
      for (int32_t g0 = c0; g0 < c1; g0 += 64 * U) {
        // U groups of 64 blocks in flight per trip; a group past the end is clamped onto the last block of the
        // chunk (loads stay in range, nothing is stored for it)
        int32_t nb[U], g[U];
        int64_t m[U];
        double v[U][9], xv[U][3];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int32_t gu = min(g0 + 64 * u, c1 - 1);
          nb[u] = (g0 + 64 * u < c1) ? min(64, c1 - gu) : 0;  // wave-uniform
          g[u] = min(gu + lane, c1 - 1);
          m[u] = nidx[g[u]];
          const VT* src = vals + 9 * (int64_t)gu;
          const int32_t last = max(nb[u], 1) * 9 - 1;
#pragma unroll
          for (int k = 0; k < 9; ++k) v[u][k] = (double)src[min(k * 64 + lane, last)];  // (non-temporal: +4 % at 96^3, 0 at 128^3: not used)
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          xv[u][0] = x[3 * m[u]]; xv[u][1] = x[3 * m[u] + 1]; xv[u][2] = x[3 * m[u] + 2];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
          for (int k = 0; k < 9; ++k) stage[k * 64 + lane] = v[u][k];
          const double* a = stage + 9 * min(lane, max(nb[u], 1) - 1);
          const double p0 = a[0] * xv[u][0] + a[1] * xv[u][1] + a[2] * xv[u][2];
          const double p1 = a[3] * xv[u][0] + a[4] * xv[u][1] + a[5] * xv[u][2];
          const double p2 = a[6] * xv[u][0] + a[7] * xv[u][1] + a[8] * xv[u][2];
          if (lane < nb[u]) {
            prod[g[u] - c0] = p0;  // plane layout: conflict-free strips
            prod[CH + g[u] - c0] = p1;
            prod[2 * CH + g[u] - c0] = p2;
          }
        }
      }
      for (int32_t g = max(rb, c0); g < min(re, c1); ++g) {
        acc[0] += prod[g - c0];
        acc[1] += prod[CH + g - c0];
        acc[2] += prod[2 * CH + g - c0];
      }
    }
    if (valid) {
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int64_t q = 3 * n + k;
        double out = acc[k];
        if (sm_b) {
          const double res = sm_b[q] - out;
          out = sm_dinv ? x[q] + sm_dinv[q] * res : res;
        }
        y[q] = out;
        if (dot_part) dacc += (sm_b ? sm_b[q] : x[q]) * out;
      }
    }
  }
  if (dot_part) {
    dacc = wave_sum(dacc);
    if (lane == 0) dot_part[blockIdx.x] = dacc;
  }
}

template <int D>
__global__ void k_extract_diag(int64_t nnodes, const int32_t* __restrict__ nptr, const uint8_t* __restrict__ diagpos,
                               const double* __restrict__ vals, double* __restrict__ diag, int invert) {
  const int64_t n = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (n >= nnodes) return;
  const bool empty = nptr[n] == nptr[n + 1];  // a node no element references (pyfem.py:683-684 allows them): unit diagonal
  const int64_t b = ((int64_t)nptr[n] + diagpos[n]) * D * D;
#pragma unroll
  for (int k = 0; k < D; ++k) {
    const double v = empty ? 1.0 : vals[b + k * D + k];
    diag[D * n + k] = invert ? 1.0 / v : v;
  }
}

// vals[diagonal of dof q] += add[q]
template <int D>
__global__ void k_add_to_diag(int64_t nnodes, const int32_t* __restrict__ nptr, const uint8_t* __restrict__ diagpos,
                              double* __restrict__ vals, const double* __restrict__ add) {
  const int64_t n = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (n >= nnodes || nptr[n] == nptr[n + 1]) return;  // empty row: nothing to add to
  const int64_t b = ((int64_t)nptr[n] + diagpos[n]) * D * D;
#pragma unroll
  for (int k = 0; k < D; ++k) vals[b + k * D + k] += add[D * n + k];
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
// vector kernels: persistent 256-thread blocks, 16-byte accesses, per-block partial sums
// ------------------------------------------------------------------------------------------
#define PF_VEC_LOOP(i, n2) \
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < (n2); i += (int64_t)gridDim.x * blockDim.x)

__device__ __forceinline__ double2 ld2(const double* p, int64_t i) { return reinterpret_cast<const double2*>(p)[i]; }
__device__ __forceinline__ void st2(double* p, int64_t i, double2 v) { reinterpret_cast<double2*>(p)[i] = v; }
// non-temporal forms for the streams an iteration touches once (x, r, Ap, dinv; p and s in the single-reduction form):
// they stop evicting what IS re-read soon -- p / z, which the next product gathers, and the fresh Ap / w
#ifndef PF_AB_CG_NT
#define PF_AB_CG_NT 1  // r02, 8.4 M dofs: 2435-2595 -> 2663-2905 Jacobi-PCG iterations/s (tools/spmv_ab.py); 0 = plain accesses
#endif
typedef double pf_vec_f64x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double2 ld2s(const double* p, int64_t i) {
  if (!PF_AB_CG_NT) return ld2(p, i);
  const pf_vec_f64x2 v = __builtin_nontemporal_load(reinterpret_cast<const pf_vec_f64x2*>(p) + i);
  return make_double2(v.x, v.y);
}
__device__ __forceinline__ void st2s(double* p, int64_t i, double2 v) {
  if (!PF_AB_CG_NT) { st2(p, i, v); return; }
  __builtin_nontemporal_store(pf_vec_f64x2{v.x, v.y}, reinterpret_cast<pf_vec_f64x2*>(p) + i);
}

// out_part[b] = sum w a b
__global__ __launch_bounds__(256) void k_dot(int64_t n, const double* __restrict__ a, const double* __restrict__ b,
                                             const double* __restrict__ w, double* __restrict__ out_part) {
  __shared__ double sh[4];
  double s = 0.0;
  PF_VEC_LOOP(i, n) s += (w ? w[i] : 1.0) * a[i] * b[i];
  block_store_partial(s, out_part, sh);
}

// out[k] = sum of partial array k, k < narr; one 256-thread block
__global__ __launch_bounds__(256) void k_finalize(int narr, const double* __restrict__ p0, int n0,
                                                  const double* __restrict__ p1, int n1,
                                                  const double* __restrict__ p2, int n2, double* __restrict__ out,
                                                  double* __restrict__ zero, int64_t nzero) {
  __shared__ double sh[4];
  for (int64_t i = threadIdx.x; i < nzero; i += blockDim.x) zero[i] = 0.0;
  double t = block_sum_partials(p0, n0, sh);
  if (threadIdx.x == 0) out[0] = t;
  if (narr > 1) {
    t = block_sum_partials(p1, n1, sh);
    if (threadIdx.x == 0) out[1] = t;
  }
  if (narr > 2) {
    t = block_sum_partials(p2, n2, sh);
    if (threadIdx.x == 0) out[2] = t;
  }
}

// alpha = rz/pAp; x += alpha p; r -= alpha Ap; z = dinv r; partials of r.z and r.r
__global__ __launch_bounds__(256) void k_cg_update_xr(int64_t n, const double* __restrict__ rz_part, int n_rz,
                                                      const double* __restrict__ pAp_part, int n_pAp,
                                                      const double* __restrict__ p, const double* __restrict__ Ap,
                                                      const double* __restrict__ dinv, double* __restrict__ x,
                                                      double* __restrict__ r, double* __restrict__ z,
                                                      double* __restrict__ rz_new_part, double* __restrict__ rr_part, int write_z) {
  // write_z == 0 (AMG-PCG): z and r.z are the V-cycle's to produce; this kernel neither stores z nor its r.z partials
  __shared__ double sh[4];
  const double rz = block_sum_partials(rz_part, n_rz, sh);
  const double pAp = block_sum_partials(pAp_part, n_pAp, sh);
  const double alpha = rz / pAp;
  double s1 = 0.0, s2 = 0.0;
  const int64_t n2 = n >> 1;
  PF_VEC_LOOP(i, n2) {
    const double2 pv = ld2(p, i), av = ld2s(Ap, i);
    double2 xv = ld2s(x, i), rv = ld2s(r, i), zv;
    xv.x += alpha * pv.x; xv.y += alpha * pv.y;
    rv.x -= alpha * av.x; rv.y -= alpha * av.y;
    if (dinv) { const double2 dv = ld2s(dinv, i); zv.x = dv.x * rv.x; zv.y = dv.y * rv.y; } else zv = rv;
    st2s(x, i, xv); st2s(r, i, rv);
    if (write_z) st2(z, i, zv);
    s1 += rv.x * zv.x + rv.y * zv.y;
    s2 += rv.x * rv.x + rv.y * rv.y;
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    const int64_t i = n - 1;
    x[i] += alpha * p[i];
    const double ri = r[i] - alpha * Ap[i];
    r[i] = ri;
    const double zi = dinv ? dinv[i] * ri : ri;
    if (write_z) z[i] = zi;
    s1 += ri * zi;
    s2 += ri * ri;
  }
  if (write_z) block_store_partial(s1, rz_new_part, sh);
  block_store_partial(s2, rr_part, sh);
}

// beta = rz_new/rz_old; p = z + beta p
__global__ __launch_bounds__(256) void k_cg_update_p(int64_t n, const double* __restrict__ rz_new_part, int n_new,
                                                     const double* __restrict__ rz_old_part, int n_old,
                                                     const double* __restrict__ z, double* __restrict__ p) {
  __shared__ double sh[4];
  const double rzn = block_sum_partials(rz_new_part, n_new, sh);
  const double rzo = block_sum_partials(rz_old_part, n_old, sh);
  const double beta = rzn / rzo;
  const int64_t n2 = n >> 1;
  PF_VEC_LOOP(i, n2) {
    const double2 zv = ld2(z, i);
    double2 pv = ld2(p, i);
    pv.x = zv.x + beta * pv.x; pv.y = zv.y + beta * pv.y;
    st2(p, i, pv);
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) p[n - 1] = z[n - 1] + beta * p[n - 1];
}

// r = b - Ap ; z = dinv r ; p = z ; partials of r.z, b.b, r.r
__global__ __launch_bounds__(256) void k_cg_init(int64_t n, const double* __restrict__ b, const double* __restrict__ Ap,
                                                 const double* __restrict__ dinv, double* __restrict__ r,
                                                 double* __restrict__ z, double* __restrict__ p,
                                                 double* __restrict__ rz_part, double* __restrict__ bb_part,
                                                 double* __restrict__ rr_part) {
  __shared__ double sh[4];
  double s1 = 0.0, s2 = 0.0, s3 = 0.0;
  PF_VEC_LOOP(i, n) {
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
  block_store_partial(s1, rz_part, sh);
  block_store_partial(s2, bb_part, sh);
  block_store_partial(s3, rr_part, sh);
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
                                                         double* __restrict__ z, double* __restrict__ gamma_part,
                                                         double* __restrict__ rr_part) {
  __shared__ double sh[4];
  const double alpha = state[0];
  double s1 = 0.0, s2 = 0.0;
  const int64_t n2 = n >> 1;
  PF_VEC_LOOP(i, n2) {
    const double2 pv = ld2(p, i), sv = ld2(s, i);
    double2 xv = ld2(x, i), rv = ld2(r, i), zv;
    xv.x += alpha * pv.x; xv.y += alpha * pv.y;
    rv.x -= alpha * sv.x; rv.y -= alpha * sv.y;
    if (dinv) { const double2 dv = ld2(dinv, i); zv.x = dv.x * rv.x; zv.y = dv.y * rv.y; } else zv = rv;
    st2(x, i, xv); st2(r, i, rv); st2(z, i, zv);
    double2 wv = make_double2(1.0, 1.0);
    if (w) wv = ld2(w, i);
    s1 += wv.x * rv.x * zv.x + wv.y * rv.y * zv.y;
    s2 += wv.x * rv.x * rv.x + wv.y * rv.y * rv.y;
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    const int64_t i = n - 1;
    x[i] += alpha * p[i];
    const double ri = r[i] - alpha * s[i];
    r[i] = ri;
    const double zi = dinv ? dinv[i] * ri : ri;
    z[i] = zi;
    const double wi = w ? w[i] : 1.0;
    s1 += wi * ri * zi;
    s2 += wi * ri * ri;
  }
  block_store_partial(s1, gamma_part, sh);
  block_store_partial(s2, rr_part, sh);
}

// scalar recurrences of the single-reduction CG, evaluated redundantly by every block:
//   beta = first ? 0 : gamma_new/gamma ; alpha = gamma_new/(delta - beta*gamma_new/alpha)
// state_in = [alpha, beta, gamma] of the previous iteration, red = reduced [gamma_new, delta, rr];
// block 0 publishes the new state to state_out (ping-pong: nobody reads what it overwrites).
__global__ __launch_bounds__(256) void k_cgcg_update_ps(int64_t n, const double* __restrict__ state_in,
                                                        double* __restrict__ state_out,
                                                        const double* __restrict__ red, int first,
                                                        const double* __restrict__ z, const double* __restrict__ w,
                                                        double* __restrict__ p, double* __restrict__ s) {
  const double gamma_new = red[0], delta = red[1];
  double alpha, beta;
  if (first) {
    beta = 0.0;
    alpha = gamma_new / delta;
  } else {
    beta = gamma_new / state_in[2];
    alpha = gamma_new / (delta - beta * gamma_new / state_in[0]);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    state_out[0] = alpha;
    state_out[1] = beta;
    state_out[2] = gamma_new;
  }
  const int64_t n2 = n >> 1;
  PF_VEC_LOOP(i, n2) {
    const double2 zv = ld2(z, i), wv = ld2(w, i);
    double2 pv = ld2(p, i), sv = ld2(s, i);
    pv.x = zv.x + beta * pv.x; pv.y = zv.y + beta * pv.y;
    sv.x = wv.x + beta * sv.x; sv.y = wv.y + beta * sv.y;
    st2(p, i, pv); st2(s, i, sv);
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    p[n - 1] = z[n - 1] + beta * p[n - 1];
    s[n - 1] = w[n - 1] + beta * s[n - 1];
  }
}

// The two vector kernels of one single-reduction iteration in ONE pass (pf_dist_cg): from the reduced scalars
// red = [gamma, delta, rr] and the previous state,  beta, alpha as in k_cgcg_update_ps;  p = z + beta p;  s = w + beta s;
// x += alpha p;  r -= alpha s;  z = dinv r;  partials of sum own r z and sum own r r for the next reduction.
// p, s, x, r are read and written once instead of twice, and the iteration has one launch less.
__global__ __launch_bounds__(256) void k_cgcg_fused(int64_t n, const double* __restrict__ state_in, double* __restrict__ state_out,
                                                    const double* __restrict__ red, int first,
                                                    const double* __restrict__ w, const double* __restrict__ dinv,
                                                    const double* __restrict__ own, double* __restrict__ z,
                                                    double* __restrict__ p, double* __restrict__ s, double* __restrict__ x,
                                                    double* __restrict__ r, double* __restrict__ gamma_part,
                                                    double* __restrict__ rr_part) {
  __shared__ double sh[4];
  const double gamma_new = red[0], delta = red[1];
  double alpha, beta;
  if (first) {
    beta = 0.0;
    alpha = gamma_new / delta;
  } else {
    beta = gamma_new / state_in[2];
    alpha = gamma_new / (delta - beta * gamma_new / state_in[0]);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    state_out[0] = alpha;
    state_out[1] = beta;
    state_out[2] = gamma_new;
  }
  double s1 = 0.0, s2 = 0.0;
  const int64_t n2 = n >> 1;
  PF_VEC_LOOP(i, n2) {
    const double2 zv = ld2(z, i), wv = ld2(w, i);
    double2 pv = ld2s(p, i), sv = ld2s(s, i), xv = ld2s(x, i), rv = ld2s(r, i), zn;
    pv.x = zv.x + beta * pv.x; pv.y = zv.y + beta * pv.y;
    sv.x = wv.x + beta * sv.x; sv.y = wv.y + beta * sv.y;
    xv.x += alpha * pv.x; xv.y += alpha * pv.y;
    rv.x -= alpha * sv.x; rv.y -= alpha * sv.y;
    if (dinv) { const double2 dv = ld2s(dinv, i); zn.x = dv.x * rv.x; zn.y = dv.y * rv.y; } else zn = rv;
    st2s(p, i, pv); st2s(s, i, sv); st2s(x, i, xv); st2s(r, i, rv); st2(z, i, zn);
    double2 ov = make_double2(1.0, 1.0);
    if (own) ov = ld2s(own, i);
    s1 += ov.x * rv.x * zn.x + ov.y * rv.y * zn.y;
    s2 += ov.x * rv.x * rv.x + ov.y * rv.y * rv.y;
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    const int64_t i = n - 1;
    const double pi = z[i] + beta * p[i], si = w[i] + beta * s[i];
    p[i] = pi;
    s[i] = si;
    x[i] += alpha * pi;
    const double ri = r[i] - alpha * si;
    r[i] = ri;
    const double zi = dinv ? dinv[i] * ri : ri;
    z[i] = zi;
    const double oi = own ? own[i] : 1.0;
    s1 += oi * ri * zi;
    s2 += oi * ri * ri;
  }
  block_store_partial(s1, gamma_part, sh);
  block_store_partial(s2, rr_part, sh);
}

// ------------------------------------------------------------------------------------------
// The update kernel of the distributed single-reduction PCG (pf_dist_cg): everything between two products in ONE launch.
//   1. every block sums this rank's partials (gamma = own r.z and r.r from the previous update, delta = z.Kz from the
//      product) in a fixed order -- the same bits in every block; block 0 stores the three scalars into every OTHER rank's
//      window and raises its scalar flag there;
//   2. every block waits for the scalar flags of the other ranks (bounded) and sums the scalars in rank order, its own in
//      its place: the all-reduce of the iteration, deterministic, without a collective and without a launch of its own;
//   3. the first blocks wait for the row flags of the neighbours -- the rows left the neighbours' PRODUCT kernels
//      (k_spmv<..., SEND>) while those were still computing their interior -- and form w on the interface entries as the
//      sum over the sharing ranks in ascending rank order (same bits on every sharer), then update those entries;
//   4. all blocks run the fused vector update (k_cgcg_fused's arithmetic) over the dofs that are NOT on an interface
//      (flags8 bit 1), with the ownership weight of the dot products from bit 0.
// keep_z (AMG-PCG, pf_dist_pcg_amg): z is the V-cycle's to write; this kernel leaves it alone, and the r.z partials it
// stores are not used (the caller forms r.z after the cycle).
// No thread of this kernel waits for another block of the same kernel, and block 0 never waits before it has sent: the
// grid drains whenever every rank reaches its update launch (or the timeout fires).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void dcg_point(double wv, double zv, double beta, double alpha, double dv, bool has_d, double& p, double& s,
                                          double& x, double& r, double& zn) {
  p = zv + beta * p;
  s = wv + beta * s;
  x += alpha * p;
  r -= alpha * s;
  zn = has_d ? dv * r : r;
}

__global__ __launch_bounds__(256) void k_dcg_update(int64_t n, pf_halo_recv rc, const double* __restrict__ gpart, int ngpart,
                                                    const double* __restrict__ rrpart, int nrrpart,
                                                    const double* __restrict__ dpart, int ndpart,
                                                    const double* __restrict__ state_in, double* __restrict__ state_out,
                                                    double* __restrict__ redg, int first, double tol2, int it, int keep_z,
                                                    const double* __restrict__ w,
                                                    const double* __restrict__ dinv, const uint8_t* __restrict__ flg,
                                                    double* __restrict__ z, double* __restrict__ p, double* __restrict__ s,
                                                    double* __restrict__ x, double* __restrict__ r,
                                                    double* __restrict__ gamma_part, double* __restrict__ rr_part) {
  __shared__ double sh[12];
  __shared__ int bad;
  __shared__ double gsum[3];
  const double stopped = redg[3];
  if (threadIdx.x == 0) bad = *rc.err;
  const int spar = (int)(rc.sseq & 1);
  // The loads of this thread's FIRST pair do not depend on the scalars: they are issued before the wait, so that the
  // flag's round trip (block 0's reduction, the store into every window, xGMI) hides behind them.  At one rank's share
  // of the 4M-quad mesh at N = 8 a thread has two pairs in all.
  const int64_t n2 = n >> 1;
  const uint16_t* flg2 = reinterpret_cast<const uint16_t*>(flg);
  const int64_t i0 = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  uint32_t f0 = 0x0202u;
  double2 z0, w0, p0, s0, x0, r0, d0 = make_double2(1.0, 1.0);
  if (i0 < n2) {
    f0 = flg2[i0];
    if (!(f0 & 0x0202u)) {
      z0 = ld2(z, i0); w0 = ld2(w, i0); p0 = ld2s(p, i0); s0 = ld2s(s, i0); x0 = ld2s(x, i0); r0 = ld2s(r, i0);
      if (dinv) d0 = ld2s(dinv, i0);
    }
  }
  // stopped != 0: an earlier update kernel of this solve found ||r|| <= tol (every rank forms the same sum of the same
  // scalars in the same order, so every rank stops at the same iteration): nothing is sent, waited for or updated.  (Read
  // above, tested here: the prefetch loads are harmless and need not wait for it.)
  if (stopped != 0.0) return;
  // 1. this rank's scalars: EVERY block sums the partial arrays itself (fixed order: the same bits in every block, and the
  //    bits block 0 sends).  Round 3, late: until then only block 0 summed, stored the result into every window including
  //    its own and all other blocks waited for that flag -- a serial 5-8 us at the head of a 27 us kernel on one rank's
  //    share at N = 8.  The partial arrays this kernel WRITES at its end are therefore not the ones it reads (a fast block
  //    would overwrite its slot while a slow one still sums): the host alternates two sets (pf_dist_update).
  double loc[3];
  block_sum_partials3(gpart, ngpart, dpart, ndpart, rrpart, nrrpart, sh, loc);
  if (blockIdx.x == 0 && (int)threadIdx.x < rc.world && (int)threadIdx.x != rc.rank) {  // ... and block 0 sends them to the others
    double* pw = rc.peer_win[threadIdx.x];
    double* dst = pw + halo_scal2_off(rc.world) + ((size_t)spar * rc.world + rc.rank) * PF_HALO_NSCAL;
    pf_st_sys_f64(dst, loc[0]);
    pf_st_sys_f64(dst + 1, loc[1]);
    pf_st_sys_f64(dst + 2, loc[2]);
    pf_release_sys();  // one wave of the whole launch
    __hip_atomic_store(reinterpret_cast<uint64_t*>(pw) + halo_sflag_off(rc.world) + rc.rank, rc.sseq, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_SYSTEM);
  }
  // 2. wait for every OTHER rank's scalars
  if ((int)threadIdx.x < rc.world && (int)threadIdx.x != rc.rank && !bad) {
    const uint64_t* flag = reinterpret_cast<const uint64_t*>(rc.win) + halo_sflag_off(rc.world) + threadIdx.x;
    if (pf_wait_flag(flag, rc.sseq, rc.timeout_ticks)) bad = 1;
  }
  __syncthreads();
  if (bad) {
    if (threadIdx.x == 0) *rc.err = 1;
    return;
  }
  if (threadIdx.x < 3) {
    const double* sc = rc.win + halo_scal2_off(rc.world) + (size_t)spar * rc.world * PF_HALO_NSCAL + threadIdx.x;
    double t = 0.0;
    for (int q = 0; q < rc.world; ++q) t += (q == rc.rank) ? loc[threadIdx.x] : pf_ld_sys_f64(sc + (size_t)q * PF_HALO_NSCAL);
    gsum[threadIdx.x] = t;
  }
  __syncthreads();
  const double gamma_new = gsum[0], delta = gsum[1];
  double alpha, beta;
  if (first) {
    beta = 0.0;
    alpha = gamma_new / delta;
  } else {
    beta = gamma_new / state_in[2];
    alpha = gamma_new / (delta - beta * gamma_new / state_in[0]);
  }
  const bool conv = gsum[2] <= tol2;  // the residual after `it` updates meets the tolerance: x stays as it is
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    state_out[0] = alpha;
    state_out[1] = beta;
    state_out[2] = gamma_new;
    redg[0] = gamma_new;
    redg[1] = delta;
    redg[2] = gsum[2];  // r.r of the residual after `it` updates (what the host reads)
    redg[4] = (double)it;
    if (conv) redg[3] = 1.0;
  }
  if (conv) return;
  double s1 = 0.0, s2 = 0.0;
  // 3. interface entries: ordered sums of the rows that arrived, then the same update
  const int nrb = (int)min((int64_t)gridDim.x, (rc.n_if + 255) / 256);
  if ((int)blockIdx.x < nrb) {
    if ((int)threadIdx.x < rc.nnbr) {
      const uint64_t* flag = reinterpret_cast<const uint64_t*>(rc.win) + halo_rflag_off(rc.world) + rc.nbr[threadIdx.x];
      if (pf_wait_flag(flag, rc.rseq, rc.timeout_ticks)) bad = 1;
    }
    __syncthreads();
    if (bad) {  // block-uniform: the vectors of this solve are invalid from here on, the host raises
      if (threadIdx.x == 0) *rc.err = 1;
      return;
    }
    const double* recv = rc.win + halo_recv2_off(rc.world, rc.cap) + (size_t)(rc.rseq & 1) * rc.world * rc.cap;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < rc.n_if; i += (int64_t)nrb * blockDim.x) {
      const int32_t li = rc.if_idx[i];
      const double own = w[li];
      double ws = 0.0;
      for (int32_t k = rc.src_ptr[i]; k < rc.src_ptr[i + 1]; ++k) {
        const int32_t o = rc.src_off[k];
        ws += (o < 0) ? own : pf_ld_sys_f64(recv + o);
      }
      double pv = p[li], sv = s[li], xv = x[li], rv = r[li], zn;
      dcg_point(ws, z[li], beta, alpha, dinv ? dinv[li] : 1.0, dinv != nullptr, pv, sv, xv, rv, zn);
      p[li] = pv; s[li] = sv; x[li] = xv; r[li] = rv; if (!keep_z) z[li] = zn;
      if (flg[li] & 1) { s1 += rv * zn; s2 += rv * rv; }
    }
  }
  // 4. everything that is not on an interface
  auto pair = [&](int64_t i, uint32_t f, double2 zv, double2 wv, double2 pv, double2 sv, double2 xv, double2 rv, double2 dv) {
    double2 zn;
    dcg_point(wv.x, zv.x, beta, alpha, dv.x, dinv != nullptr, pv.x, sv.x, xv.x, rv.x, zn.x);
    dcg_point(wv.y, zv.y, beta, alpha, dv.y, dinv != nullptr, pv.y, sv.y, xv.y, rv.y, zn.y);
    st2s(p, i, pv); st2s(s, i, sv); st2s(x, i, xv); st2s(r, i, rv); if (!keep_z) st2(z, i, zn);
    if (f & 0x0001u) { s1 += rv.x * zn.x; s2 += rv.x * rv.x; }
    if (f & 0x0100u) { s1 += rv.y * zn.y; s2 += rv.y * rv.y; }
  };
  auto mixed = [&](int64_t i, uint32_t f) {  // a pair that touches an interface (d = 1 meshes: a node pair with one interface node)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const uint32_t fc = (f >> (8 * c)) & 0xffu;
      if (fc & 2u) continue;
      const int64_t q = 2 * i + c;
      double pv = p[q], sv = s[q], xv = x[q], rv = r[q], zn;
      dcg_point(w[q], z[q], beta, alpha, dinv ? dinv[q] : 1.0, dinv != nullptr, pv, sv, xv, rv, zn);
      p[q] = pv; s[q] = sv; x[q] = xv; r[q] = rv; if (!keep_z) z[q] = zn;
      if (fc & 1u) { s1 += rv * zn; s2 += rv * rv; }
    }
  };
  if (i0 < n2) {
    if (!(f0 & 0x0202u)) pair(i0, f0, z0, w0, p0, s0, x0, r0, d0);
    else mixed(i0, f0);
  }
  for (int64_t i = i0 + (int64_t)gridDim.x * blockDim.x; i < n2; i += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t f = flg2[i];
    if (!(f & 0x0202u)) {
      double2 dv = make_double2(1.0, 1.0);
      const double2 zv = ld2(z, i), wv = ld2(w, i), pv = ld2s(p, i), sv = ld2s(s, i), xv = ld2s(x, i), rv = ld2s(r, i);
      if (dinv) dv = ld2s(dinv, i);
      pair(i, f, zv, wv, pv, sv, xv, rv, dv);
    } else {
      mixed(i, f);
    }
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0 && !(flg[n - 1] & 2)) {
    const int64_t q = n - 1;
    double pv = p[q], sv = s[q], xv = x[q], rv = r[q], zn;
    dcg_point(w[q], z[q], beta, alpha, dinv ? dinv[q] : 1.0, dinv != nullptr, pv, sv, xv, rv, zn);
    p[q] = pv; s[q] = sv; x[q] = xv; r[q] = rv; if (!keep_z) z[q] = zn;
    if (flg[q] & 1) { s1 += rv * zn; s2 += rv * rv; }
  }
  block_store_partial(s1, gamma_part, sh);
  block_store_partial(s2, rr_part, sh);
}

// ------------------------------------------------------------------------------------------
static inline unsigned vec_grid(int64_t n) {
  return (unsigned)std::max<int64_t>(1, std::min<int64_t>(cdiv(n, 512), VEC_BLOCKS));
}
static inline unsigned spmv_grid_d(int64_t nnodes, int d) {
  return (unsigned)std::min<int64_t>(cdiv(nnodes, d == 1 ? SPMV_TILE1 : SPMV_TILE2), SPMV_MAXGRID);
}
static inline double* part(pf_ctx* ctx, int which) { return ctx->scal + (size_t)which * NPART; }
int pf_spmv_waves(pf_ctx* ctx) { return (int)spmv_grid_d(ctx->nnodes, ctx->d); }

// 16-bit column offsets for the d = 1 and d = 2 products (12 -> 10 bytes per scalar nonzero, 36 -> 34 per 2x2 block):
// neighbour id minus the first node of the
// row's tile.  Fits whenever the numbering keeps neighbours within +-32 k node ids (any mesh with a sensible
// ordering); otherwise flag[0] is raised and the 32-bit indices stay in use.
__global__ void k_build_idx16(int64_t nnodes, int tile_nodes, const int32_t* __restrict__ nptr,
                              const int32_t* __restrict__ nidx, int16_t* __restrict__ out, int32_t* __restrict__ flag) {
  const int64_t n = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (n >= nnodes) return;
  const int64_t base = (n / tile_nodes) * tile_nodes;
  for (int32_t g = nptr[n]; g < nptr[n + 1]; ++g) {
    const int64_t rel = (int64_t)nidx[g] - base;
    if (rel < -32768 || rel > 32767) { flag[0] = 1; return; }
    out[g] = (int16_t)rel;
  }
}

static int ensure_idx16(pf_ctx* ctx) {
  if (ctx->nidx16_state) return PF_OK;
  const char* e = pf_exp_env("PF_SPMV_IDX16");
  if (e && atoi(e) == 0) { ctx->nidx16_state = -1; return PF_OK; }
  int32_t* flag = nullptr;
  PF_HIP(hipMalloc(&ctx->nidx16, sizeof(int16_t) * (size_t)std::max<int64_t>(1, ctx->nblocks)));
  PF_HIP(hipMalloc(&flag, sizeof(int32_t)));
  PF_HIP(hipMemsetAsync(flag, 0, sizeof(int32_t), ctx->stream));
  k_build_idx16<<<(unsigned)cdiv(ctx->nnodes, 256), 256, 0, ctx->stream>>>(ctx->nnodes, ctx->d == 1 ? SPMV_TILE1 : SPMV_TILE2,
                                                                          ctx->nptr, ctx->nidx, ctx->nidx16, flag);
  int32_t f = 0;
  PF_HIP(hipMemcpyAsync(&f, flag, sizeof(f), hipMemcpyDeviceToHost, ctx->stream));
  PF_HIP(hipStreamSynchronize(ctx->stream));
  (void)hipFree(flag);
  ctx->nidx16_state = f ? -1 : 1;
  if (f) { pf_free(ctx->nidx16); }
  return PF_OK;
}

// vals32 != nullptr: the product uses that fp32 copy of the values instead of vals (AMG smoother sweeps)
// one launch of k_spmv<SHAPE..., VT> for the runtime choices (16-bit columns, fused send, non-temporal value stream)
template <int D, int R, int U, int CH, typename VT>
static void spmv_dispatch(pf_ctx* ctx, unsigned grid, bool i16, bool ntv, const int32_t* cols, const VT* v, const double* x, double* y,
                          double* dot_part, const double* sm_dinv, const double* sm_b, const pf_spmv_send* snd,
                          const int32_t* tperm, const double* pw_diag) {
#define PF_GO(I16_, SEND_, NTV_, SND_)                                                                                \
  k_spmv<D, R, U, CH, I16_, VT, SEND_, NTV_><<<grid, 64, 0, ctx->stream>>>(ctx->nnodes, ctx->nptr, cols, v, x, y, dot_part, sm_dinv, \
                                                                           sm_b, tperm, pw_diag, SND_)
  if (snd) {
    if (i16) { if (ntv) PF_GO(true, true, true, *snd); else PF_GO(true, true, false, *snd); }
    else { if (ntv) PF_GO(false, true, true, *snd); else PF_GO(false, true, false, *snd); }
  } else {
    if (i16) { if (ntv) PF_GO(true, false, true, pf_spmv_nosend{}); else PF_GO(true, false, false, pf_spmv_nosend{}); }
    else { if (ntv) PF_GO(false, false, true, pf_spmv_nosend{}); else PF_GO(false, false, false, pf_spmv_nosend{}); }
  }
#undef PF_GO
}

int pf_launch_spmv(pf_ctx* ctx, const double* vals, const double* x, double* y, double* dot_part,
                   const double* sm_dinv, const double* sm_b, const float* vals32, const pf_spmv_send* snd,
                   const int32_t* snd_tperm, int nv, const double* pw_diag) {
  unsigned grid = spmv_grid_d(ctx->nnodes, ctx->d);
  if (pw_diag && (ctx->d == 3 || nv != 1 || snd || sm_b || sm_dinv))
    return pf_fail(ctx, PF_E_UNSUPPORTED, "power-iteration product: plain one-vector product of 1 or 2 dofs per node");
  if (nv == 2) {  // scalar matrix, two vectors
    if (ctx->d != 1 || snd) return pf_fail(ctx, PF_E_UNSUPPORTED, "two-vector product: scalar matrices, no fused exchange");
    if (int rc16 = ensure_idx16(ctx)) return rc16;
    const bool i16 = ctx->nidx16_state == 1;
    const int32_t* cols = i16 ? reinterpret_cast<const int32_t*>(ctx->nidx16) : ctx->nidx;
    grid = std::min(grid, (unsigned)SPMV2_MAXGRID);  // two partials per wave in one partial array
#ifndef PF_AB_NV2_CH
#define PF_AB_NV2_CH 1280
#endif
#ifndef PF_AB_NV2_U
#define PF_AB_NV2_U 9
#endif
#define PF_GO2(I16_, VT_, V_) k_spmv_nv2<2, PF_AB_NV2_U, PF_AB_NV2_CH, I16_, VT_><<<grid, 64, 0, ctx->stream>>>(ctx->nnodes, ctx->nptr, cols, V_, x, y, dot_part, sm_dinv, sm_b, ctx->spmv_perm)
    if (vals32) { if (i16) PF_GO2(true, float, vals32); else PF_GO2(false, float, vals32); }
    else { if (i16) PF_GO2(true, double, vals); else PF_GO2(false, double, vals); }
#undef PF_GO2
    PF_LAUNCH_CHECK();
    return PF_OK;
  }
  if (nv != 1) return pf_fail(ctx, PF_E_INVALID, "pf_launch_spmv: one or two vectors");
  if (ctx->d == 3) {
    if (snd) return pf_fail(ctx, PF_E_UNSUPPORTED, "fused exchange: 1 or 2 dofs per node");
    if (vals32) k_spmv3<float><<<grid, 64, 0, ctx->stream>>>(ctx->nnodes, ctx->nptr, ctx->nidx, vals32, x, y, dot_part, sm_dinv, sm_b);
    else k_spmv3<double><<<grid, 64, 0, ctx->stream>>>(ctx->nnodes, ctx->nptr, ctx->nidx, vals, x, y, dot_part, sm_dinv, sm_b);
    PF_LAUNCH_CHECK();
    return PF_OK;
  }
#ifdef PF_SPMV_SWEEP
  const char* sh = pf_exp_env("PF_SPMV_SHAPE");
  const int shape = (sh && !vals32 && !snd && !pw_diag) ? atoi(sh) : 0;
#define PF_TRY(ID, D_, R_, U_, CH_)                                                                         \
  if (shape == ID && ctx->d == D_) {                                                                        \
    grid = (unsigned)std::min<int64_t>(cdiv(ctx->nnodes, 64 * R_), SPMV_MAXGRID);                            \
    k_spmv<D_, R_, U_, CH_><<<grid, 64, 0, ctx->stream>>>(ctx->nnodes, ctx->nptr, ctx->nidx, vals, x, y, dot_part, sm_dinv, sm_b, \
                                                          64 * R_ == (D_ == 1 ? SPMV_TILE1 : SPMV_TILE2) ? ctx->spmv_perm : nullptr, pw_diag, pf_spmv_nosend{}); \
    PF_LAUNCH_CHECK();                                                                                      \
    return PF_OK;                                                                                           \
  }
  PF_TRY(1, 1, 4, 6, 2560) PF_TRY(2, 1, 2, 9, 1280) PF_TRY(3, 1, 4, 9, 2560) PF_TRY(4, 1, 1, 6, 640) PF_TRY(5, 1, 2, 3, 1280)
  PF_TRY(1, 2, 2, 3, 1280) PF_TRY(2, 2, 1, 5, 640) PF_TRY(3, 2, 2, 5, 1280) PF_TRY(4, 2, 1, 2, 640) PF_TRY(5, 2, 1, 3, 1024)
#undef PF_TRY
#endif
  if (int rc16 = ensure_idx16(ctx)) return rc16;
  const bool i16 = ctx->nidx16_state == 1;
  const int32_t* cols = i16 ? reinterpret_cast<const int32_t*>(ctx->nidx16) : ctx->nidx;
  const int32_t* tperm = snd ? snd_tperm : ctx->spmv_perm;
  // the value stream (+ its column ids): non-temporal only when it cannot stay in the 256 MB memory-side cache anyway
  const int64_t stream = ctx->nblocks * ((int64_t)ctx->d * ctx->d * (vals32 ? 4 : 8) + (i16 ? 2 : 4));
  const bool ntv = ctx->multi_stream || stream >= SPMV_NT_MIN_BYTES;
  if (vals32) {
    if (ctx->d == 1) spmv_dispatch<PF_SPMV1F, float>(ctx, grid, i16, ntv, cols, vals32, x, y, dot_part, sm_dinv, sm_b, snd, tperm, pw_diag);
    else spmv_dispatch<PF_SPMV2F, float>(ctx, grid, i16, ntv, cols, vals32, x, y, dot_part, sm_dinv, sm_b, snd, tperm, pw_diag);
  } else {
    if (ctx->d == 1) spmv_dispatch<PF_SPMV1, double>(ctx, grid, i16, ntv, cols, vals, x, y, dot_part, sm_dinv, sm_b, snd, tperm, pw_diag);
    else spmv_dispatch<PF_SPMV2, double>(ctx, grid, i16, ntv, cols, vals, x, y, dot_part, sm_dinv, sm_b, snd, tperm, pw_diag);
  }
  PF_LAUNCH_CHECK();
  return PF_OK;
}

extern "C" int pf_spmv(pf_ctx* ctx, const double* vals, const double* x, double* y) {
  if (!ctx) return pf_fail(nullptr, PF_E_INVALID, "ctx is NULL");
  PF_REQUIRE(ctx->d, "pf_spmv: no pattern built");
  PF_REQUIRE(vals && x && y && x != y, "pf_spmv: NULL argument or x aliases y");
  return pf_launch_spmv(ctx, vals, x, y, nullptr);
}

extern "C" int pf_extract_diag(pf_ctx* ctx, const double* vals, double* diag) {
  if (!ctx) return pf_fail(nullptr, PF_E_INVALID, "ctx is NULL");
  PF_REQUIRE(ctx->d && vals && diag, "pf_extract_diag: no pattern / NULL argument");
  const unsigned g = (unsigned)cdiv(ctx->nnodes, 256);
  if (ctx->d == 1) k_extract_diag<1><<<g, 256, 0, ctx->stream>>>(ctx->nnodes, ctx->nptr, ctx->diagpos, vals, diag, 0);
  else if (ctx->d == 3) k_extract_diag<3><<<g, 256, 0, ctx->stream>>>(ctx->nnodes, ctx->nptr, ctx->diagpos, vals, diag, 0);
  else k_extract_diag<2><<<g, 256, 0, ctx->stream>>>(ctx->nnodes, ctx->nptr, ctx->diagpos, vals, diag, 0);
  PF_LAUNCH_CHECK();
  return PF_OK;
}

extern "C" int pf_add_to_diag(pf_ctx* ctx, double* vals, const double* add) {
  if (!ctx) return pf_fail(nullptr, PF_E_INVALID, "ctx is NULL");
  PF_REQUIRE(ctx->d && vals && add, "pf_add_to_diag: no pattern / NULL argument");
  const unsigned g = (unsigned)cdiv(ctx->nnodes, 256);
  if (ctx->d == 1) k_add_to_diag<1><<<g, 256, 0, ctx->stream>>>(ctx->nnodes, ctx->nptr, ctx->diagpos, vals, add);
  else if (ctx->d == 3) k_add_to_diag<3><<<g, 256, 0, ctx->stream>>>(ctx->nnodes, ctx->nptr, ctx->diagpos, vals, add);
  else k_add_to_diag<2><<<g, 256, 0, ctx->stream>>>(ctx->nnodes, ctx->nptr, ctx->diagpos, vals, add);
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
  else if (ctx->d == 3)
    k_apply_dirichlet<3><<<g, 128, 0, ctx->stream>>>(ctx->nnodes, ctx->nptr, ctx->nidx, Kvals, rhs, fixed_mask, fixed_vals, symmetric, diag_scale);
  else
    k_apply_dirichlet<2><<<g, 128, 0, ctx->stream>>>(ctx->nnodes, ctx->nptr, ctx->nidx, Kvals, rhs, fixed_mask, fixed_vals, symmetric, diag_scale);
  PF_LAUNCH_CHECK();
  return PF_OK;
}

extern "C" int pf_dot(pf_ctx* ctx, int64_t n, const double* a, const double* b, const double* w, double* out) {
  if (!ctx) return pf_fail(nullptr, PF_E_INVALID, "ctx is NULL");
  PF_REQUIRE(a && b && out && n >= 0, "pf_dot: NULL argument");
  const unsigned g = vec_grid(n);
  k_dot<<<g, 256, 0, ctx->stream>>>(n, a, b, w, part(ctx, PA_BB));
  k_finalize<<<1, 256, 0, ctx->stream>>>(1, part(ctx, PA_BB), (int)g, nullptr, 0, nullptr, 0, out, nullptr, 0);
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

// ---- single-reduction CG steps (partials live in the ctx; pf_cgcg_reduce turns them into scalars) ----
extern "C" int pf_spmv_dot(pf_ctx* ctx, const double* vals, const double* x, double* y) {
  if (!ctx) return pf_fail(nullptr, PF_E_INVALID, "ctx is NULL");
  PF_REQUIRE(ctx->d, "pf_spmv_dot: no pattern built");
  PF_REQUIRE(vals && x && y && x != y, "pf_spmv_dot: NULL argument or x aliases y");
  return pf_launch_spmv(ctx, vals, x, y, part(ctx, PA_DELTA));
}

extern "C" int pf_cgcg_update_xrz(pf_ctx* ctx, const double* state, const double* p, const double* s,
                                  const double* dinv, const double* w, double* x, double* r, double* z) {
  if (!ctx) return pf_fail(nullptr, PF_E_INVALID, "ctx is NULL");
  PF_REQUIRE(ctx->d && state && p && s && x && r && z, "pf_cgcg_update_xrz: no pattern / NULL argument");
  const int64_t n = (int64_t)ctx->d * ctx->nnodes;
  k_cgcg_update_xrz<<<vec_grid(n), 256, 0, ctx->stream>>>(n, state, p, s, dinv, w, x, r, z, part(ctx, PA_GAMMA),
                                                          part(ctx, PA_RR2));
  PF_LAUNCH_CHECK();
  return PF_OK;
}

extern "C" int pf_cgcg_reduce(pf_ctx* ctx, double* red_out, double* zero, int64_t nzero) {
  if (!ctx) return pf_fail(nullptr, PF_E_INVALID, "ctx is NULL");
  PF_REQUIRE(ctx->d && red_out && (zero || nzero == 0), "pf_cgcg_reduce: no pattern / NULL argument");
  const int64_t n = (int64_t)ctx->d * ctx->nnodes;
  const int nv = (int)vec_grid(n), ns = (int)spmv_grid_d(ctx->nnodes, ctx->d);
  k_finalize<<<1, 256, 0, ctx->stream>>>(3, part(ctx, PA_GAMMA), nv, part(ctx, PA_DELTA), ns, part(ctx, PA_RR2), nv,
                                         red_out, zero, nzero);
  PF_LAUNCH_CHECK();
  return PF_OK;
}

extern "C" int pf_cgcg_update_ps(pf_ctx* ctx, const double* state_in, double* state_out, const double* red,
                                 int first, const double* z, const double* w, double* p, double* s) {
  if (!ctx) return pf_fail(nullptr, PF_E_INVALID, "ctx is NULL");
  PF_REQUIRE(ctx->d && state_in && state_out && state_in != state_out && red && z && w && p && s,
             "pf_cgcg_update_ps: no pattern / NULL argument / state buffers alias");
  const int64_t n = (int64_t)ctx->d * ctx->nnodes;
  k_cgcg_update_ps<<<vec_grid(n), 256, 0, ctx->stream>>>(n, state_in, state_out, red, first, z, w, p, s);
  PF_LAUNCH_CHECK();
  return PF_OK;
}

// ------------------------------------------------------------------------------------------
// native PCG loop: three launches per iteration, reductions through partial arrays, the host only
// reads the r.r partials back every `check_every` iterations.
// ------------------------------------------------------------------------------------------
static double host_sum(const double* v, int n) {
  double s = 0.0;
  for (int i = 0; i < n; ++i) s += v[i];
  return s;
}

// Residual read-back of the PCG loops without a device stall.  |r|^2 is known right after the x / r update, long before
// the iteration ends (the V-cycle and the direction update follow): the copy to PINNED host memory and an event are queued
// at that point, the rest of the iteration is queued behind them, and only then does the host wait for the event -- the
// device is still busy, and the next iteration's first launch is in its queue before it runs dry.  (r02 / early r03: copy
// to pageable memory and a stream synchronise at the end of every iteration -- a 42 us hole per AMG-PCG iteration in the
// kernel trace, 2 ms of a 40 ms topology iteration.)
int pf_pin_copy(pf_ctx* ctx, size_t double_offset, const void* dev, size_t bytes) {  // one piece into ctx->pin (2 * PF_NPART doubles)
  if (!ctx->pin) {
    PF_HIP(hipHostMalloc(reinterpret_cast<void**>(&ctx->pin), sizeof(double) * 2 * NPART, hipHostMallocDefault));
    PF_HIP(hipEventCreateWithFlags(&ctx->pin_ev, hipEventDisableTiming));
  }
  PF_HIP(hipMemcpyAsync(ctx->pin + double_offset, dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
  return PF_OK;
}
int pf_pin_mark(pf_ctx* ctx) {
  PF_HIP(hipEventRecord(ctx->pin_ev, ctx->stream));
  return PF_OK;
}
int pf_pin_wait(pf_ctx* ctx) {
  PF_HIP(hipEventSynchronize(ctx->pin_ev));
  return PF_OK;
}
static int readback_queue(pf_ctx* ctx, const double* dev, size_t count) {
  if (int rc = pf_pin_copy(ctx, 0, dev, sizeof(double) * count)) return rc;
  return pf_pin_mark(ctx);
}
static int readback_wait(pf_ctx* ctx) { return pf_pin_wait(ctx); }

// The product's output vector of the PCG loops, owned by the context.  Round 2 "placed it by measurement" (three candidate
// allocations, the fastest of a few launches each kept), on the belief that the product runs at two speeds depending on
// where its OUTPUT lies.  Round 3 took that apart (profiles/r03_spmv_placement.md): in a warm process every output
// allocation gives the same time to 0.4 %; what the three-candidate timing saw was the ramp of the first ~100 launches
// after the device switches kernels, and the real two-speed effect (0.212 against 0.235 ms) belongs to the MATRIX value
// buffer, which the caller owns.  So: one allocation, no measurement.
static int cg_place_ap(pf_ctx* ctx, const double* vals, double* xprobe, int64_t n) {
  (void)vals;
  (void)xprobe;
  if (ctx->cg_ap && ctx->cg_ap_n == n) return PF_OK;
  pf_free(ctx->cg_ap);
  ctx->cg_ap_n = 0;
  PF_HIP(hipMalloc(&ctx->cg_ap, sizeof(double) * (size_t)std::max<int64_t>(n, 1)));
  ctx->cg_ap_n = n;
  return PF_OK;
}

extern "C" int pf_cg(pf_ctx* ctx, const double* vals, const double* b, double* x, double rtol, double atol,
                     int maxiter, int precond, int check_every, int32_t* iters_host, double* relres_host) {
  if (!ctx) return pf_fail(nullptr, PF_E_INVALID, "ctx is NULL");
  PF_REQUIRE(ctx->d && vals && b && x, "pf_cg: no pattern / NULL argument");
  PF_REQUIRE(precond == PF_PRECOND_NONE || precond == PF_PRECOND_JACOBI || precond == PF_PRECOND_AMG ||
                 precond == PF_PRECOND_AMG_REUSE || precond == PF_PRECOND_AMG_STALE,
             "pf_cg: unknown preconditioner");
  const int amg_reuse = precond == PF_PRECOND_AMG_REUSE ? 1 : precond == PF_PRECOND_AMG_STALE ? 2 : 0;  // pf_amg_setup's `reuse`
  if (amg_reuse) precond = PF_PRECOND_AMG;
  PF_HIP(hipSetDevice(ctx->device));
  const int64_t n = (int64_t)ctx->d * ctx->nnodes;
  if (maxiter <= 0) maxiter = (int)std::min<int64_t>(10 * n, INT32_MAX);  // scipy's default
  if (check_every <= 0) check_every = (precond == PF_PRECOND_AMG) ? 1 : 50;  // AMG: 10-30 iterations of ~1 ms; a read-back is ~20 us
  if (ctx->cgw_cap < 5 * n) {
    pf_free(ctx->cgw);
    PF_HIP(hipMalloc(&ctx->cgw, sizeof(double) * 5 * n));
    ctx->cgw_cap = 5 * n;
  }
  double *r = ctx->cgw, *z = r + n, *p = z + n, *dinv = p + 2 * n;
  if (int prc = cg_place_ap(ctx, vals, p, n)) return prc;
  double* Ap = ctx->cg_ap;
  hipStream_t st = ctx->stream;
  const unsigned vg = vec_grid(n);
  const int ns = (int)spmv_grid_d(ctx->nnodes, ctx->d);

  if (precond == PF_PRECOND_AMG) {
    const int arc = pf_amg_setup(ctx, vals, amg_reuse);
    if (arc == PF_E_UNSUPPORTED) {  // diagonal entries of both signs among the coupled dofs: not a definite matrix
      precond = PF_PRECOND_JACOBI;
      ctx->warn = "pf_cg: the matrix is not definite (diagonal entries of both signs, e.g. negative densities); the AMG "
                  "preconditioner was refused and Jacobi is used instead";
    } else if (arc) return arc;
  }
  const bool amg = precond == PF_PRECOND_AMG;
  struct MultiStream {  // restored on every return path
    pf_ctx* c;
    bool was;
    MultiStream(pf_ctx* c_, bool on) : c(c_), was(c_->multi_stream) { c->multi_stream = on; }
    ~MultiStream() { c->multi_stream = was; }
  } multi_stream_guard(ctx, amg);
  if (precond == PF_PRECOND_JACOBI) {
    const unsigned g = (unsigned)cdiv(ctx->nnodes, 256);
    if (ctx->d == 1) k_extract_diag<1><<<g, 256, 0, st>>>(ctx->nnodes, ctx->nptr, ctx->diagpos, vals, dinv, 1);
    else if (ctx->d == 3) k_extract_diag<3><<<g, 256, 0, st>>>(ctx->nnodes, ctx->nptr, ctx->diagpos, vals, dinv, 1);
    else k_extract_diag<2><<<g, 256, 0, st>>>(ctx->nnodes, ctx->nptr, ctx->diagpos, vals, dinv, 1);
    PF_LAUNCH_CHECK();
  } else {
    dinv = nullptr;
  }
  int rc;
  // AMG-PCG: the r.z partials come out of the cycle's last sweep (one per product wave: ns of them); k_dot only when the
  // hierarchy has a single level and the cycle ends without a product
  int nrz = (int)vg;
  auto cycle = [&](double* rz_part) -> int {
    bool dotted = false;
    if (int e = pf_amg_apply(ctx, r, z, rz_part, &dotted)) return e;
    if (dotted) nrz = ns;
    else { k_dot<<<vg, 256, 0, st>>>(n, r, z, nullptr, rz_part); nrz = (int)vg; }
    return PF_OK;
  };
  double rnorm = 0.0, rnorm_prev = 0.0, bnorm = 0.0;
  // (re)start from the iterate in x: r = b - A x, z = M^-1 r, p = z, the partials of r.z / b.b / r.r
  auto start = [&]() -> int {
    if (int e = pf_launch_spmv(ctx, vals, x, Ap, nullptr)) return e;  // Ap = A x
    k_cg_init<<<vg, 256, 0, st>>>(n, b, Ap, dinv, r, z, p, part(ctx, PA_RZ0), part(ctx, PA_BB), part(ctx, PA_RR));
    PF_LAUNCH_CHECK();
    if (int e = readback_queue(ctx, part(ctx, PA_RR), 2 * NPART)) return e;  // RR | BB, awaited behind the first cycle
    nrz = (int)vg;
    if (amg) {  // z = M^-1 r by one V-cycle, p = z, r.z
      if (int e = cycle(part(ctx, PA_RZ0))) return e;
      PF_HIP(hipMemcpyAsync(p, z, sizeof(double) * n, hipMemcpyDeviceToDevice, st));
      PF_LAUNCH_CHECK();
    }
    if (int e = readback_wait(ctx)) return e;
    bnorm = sqrt(host_sum(ctx->pin + NPART, (int)vg));
    rnorm = rnorm_prev = sqrt(host_sum(ctx->pin, (int)vg));
    return PF_OK;
  };
  if ((rc = start())) return rc;
  const double tol = std::max(rtol * bnorm, atol);
  int it = 0;
  int cur = 0;  // which rz partial array is live
  // A healthy AMG-PCG solve takes 10 - 40 iterations.  One that is still running after PF_AMG_RESCUE_AT has, in every case
  // seen so far, a smoother or prolongator that over-relaxes because the power iteration under-estimated lambda_max(D^-1 A)
  // by more than the 1.1 margin (the refined arm mesh with six instead of ten steps: 1 500 iterations instead of 28,
  // profiles/r03_power_iteration_steps.txt).  Once per solve: redo the numeric setup with three times the power steps (the
  // hierarchy keeps them from then on), say so in the warning, and restart the recurrence from the iterate reached.
  bool rescued = false;
  while (rnorm > tol && it < maxiter) {
    if (amg && !rescued && it >= PF_AMG_RESCUE_AT) {
      rescued = true;
      if ((rc = pf_amg_more_power_steps(ctx))) return rc;
      if ((rc = pf_amg_setup(ctx, vals, 0))) return rc == PF_E_UNSUPPORTED ? pf_fail(ctx, rc, "pf_cg: the matrix is not definite") : rc;
      ctx->warn = "pf_cg: AMG-PCG had not converged after " + std::to_string(it) + " iterations; the hierarchy was set up again with "
                  "more power-iteration steps for its lambda_max estimates and the solve restarted from the iterate reached";
      if ((rc = start())) return rc;
      cur = 0;
      if (!(rnorm > tol)) break;
    }
    const int burst = std::min(check_every, maxiter - it);
    // Checked every iteration (AMG): when the residual's recent rate says THIS step probably ends the solve, the host
    // looks at |r|^2 before it queues the V-cycle and the direction update, and leaves them out if it does (they were a
    // whole wasted cycle per solve); otherwise it queues them first and looks afterwards (no stall, see readback_queue).
    const bool early = burst == 1 && rnorm * std::min(1.0, rnorm / rnorm_prev) <= 4.0 * tol;
    bool finished = false;
    for (int k = 0; k < burst; ++k) {
      double* rz_old = part(ctx, cur ? PA_RZ1 : PA_RZ0);
      double* rz_new = part(ctx, cur ? PA_RZ0 : PA_RZ1);
      rc = pf_launch_spmv(ctx, vals, p, Ap, part(ctx, PA_PAP));  // Ap = A p, partials of p.Ap
      if (rc) return rc;
      const int nrz_old = nrz;
      k_cg_update_xr<<<vg, 256, 0, st>>>(n, rz_old, nrz_old, part(ctx, PA_PAP), ns, p, Ap, dinv, x, r, z, rz_new,
                                         part(ctx, PA_RR), amg ? 0 : 1);
      if (k == burst - 1 && (rc = readback_queue(ctx, part(ctx, PA_RR), vg))) return rc;  // |r|^2 of the burst's last step
      if (early) {
        if ((rc = readback_wait(ctx))) return rc;
        rnorm_prev = rnorm;
        rnorm = sqrt(host_sum(ctx->pin, (int)vg));
        if (!(rnorm > tol) || it + 1 >= maxiter) { finished = true; break; }  // (NaN ends here too)
      }
      if (amg && (rc = cycle(rz_new))) return rc;
      k_cg_update_p<<<vg, 256, 0, st>>>(n, rz_new, nrz, rz_old, nrz_old, z, p);
      cur ^= 1;
    }
    PF_LAUNCH_CHECK();
    it += burst;
    if (finished) break;
    if (!early) {
      if ((rc = readback_wait(ctx))) return rc;
      rnorm_prev = rnorm;
      rnorm = sqrt(host_sum(ctx->pin, (int)vg));
    }
    if (!(rnorm == rnorm)) break;  // NaN: breakdown
  }
  if (iters_host) *iters_host = it;
  if (relres_host) *relres_host = bnorm > 0 ? rnorm / bnorm : rnorm;
  if (!(rnorm <= tol))
    return pf_fail(ctx, PF_E_NOCONV, "cg failed with code " + std::to_string(it));  // pyfem.py:2421-2422
  return PF_OK;
}

// ------------------------------------------------------------------------------------------
// AMG-PCG for TWO right-hand sides of one scalar matrix at once (the two gradient filters that topo.py:73-74 issues
// together; /root/reference/pyfem.py:2102-2115 solves them one after the other).  Vectors hold the pair of every node next
// to each other (an (n, 2) row-major array); every product, smoother sweep and transfer reads its operator ONCE for both
// (k_spmv_nv2, the NV = 2 forms of the cycle kernels) and the launch-bound coarse levels are walked once.  The two
// recurrences are independent -- each vector keeps its own alpha, beta and stopping test, so each gets the iterates pf_cg
// would give it alone -- and a vector that has converged (or is zero) is frozen: alpha = 0, x and r keep their values
// (the host, which reads |r|^2 after every iteration anyway, says which).
// Partial arrays hold the pair of every block next to each other.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double2 block_sum_partials2(const double* __restrict__ part, int nblk, double* sh8) {
  double s0 = 0.0, s1 = 0.0;
  for (int i = threadIdx.x; i < nblk; i += blockDim.x) { s0 += part[2 * i]; s1 += part[2 * i + 1]; }
  s0 = wave_sum(s0); s1 = wave_sum(s1);
  const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  if ((threadIdx.x & 63) == 0) { sh8[w] = s0; sh8[4 + w] = s1; }
  __syncthreads();
  double2 t = make_double2(0.0, 0.0);
  for (int k = 0; k < nw; ++k) { t.x += sh8[k]; t.y += sh8[4 + k]; }
  __syncthreads();
  return t;
}
__device__ __forceinline__ void block_store_partial2(double s0, double s1, double* __restrict__ out, double* sh8) {
  s0 = wave_sum(s0); s1 = wave_sum(s1);
  const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  if ((threadIdx.x & 63) == 0) { sh8[w] = s0; sh8[4 + w] = s1; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double t0 = 0.0, t1 = 0.0;
    for (int k = 0; k < nw; ++k) { t0 += sh8[k]; t1 += sh8[4 + k]; }
    out[2 * blockIdx.x] = t0;
    out[2 * blockIdx.x + 1] = t1;
  }
  __syncthreads();
}

// r = b - Ap; partials of b.b and r.r per vector (n = nodes; the vectors are pairs)
__global__ __launch_bounds__(256) void k_cg2_init(int64_t n, const double* __restrict__ b, const double* __restrict__ Ap,
                                                  double* __restrict__ r, double* __restrict__ bb_part, double* __restrict__ rr_part) {
  __shared__ double sh[8];
  double b0 = 0.0, b1 = 0.0, r0 = 0.0, r1 = 0.0;
  PF_VEC_LOOP(i, n) {
    const double2 bv = ld2(b, i), av = ld2(Ap, i);
    const double2 rv = make_double2(bv.x - av.x, bv.y - av.y);
    st2(r, i, rv);
    b0 += bv.x * bv.x; b1 += bv.y * bv.y;
    r0 += rv.x * rv.x; r1 += rv.y * rv.y;
  }
  block_store_partial2(b0, b1, bb_part, sh);
  block_store_partial2(r0, r1, rr_part, sh);
}

// alpha_v = rz_v / pAp_v (0 for a frozen vector: the host found it converged); x += alpha p; r -= alpha Ap; |r|^2
__global__ __launch_bounds__(256) void k_cg2_update_xr(int64_t n, const double* __restrict__ rz_part, int n_rz,
                                                       const double* __restrict__ pAp_part, int n_pAp, int frozen0, int frozen1,
                                                       const double* __restrict__ p, const double* __restrict__ Ap,
                                                       double* __restrict__ x, double* __restrict__ r,
                                                       double* __restrict__ rr_part) {
  __shared__ double sh[8];
  const double2 rz = block_sum_partials2(rz_part, n_rz, sh);
  const double2 pAp = block_sum_partials2(pAp_part, n_pAp, sh);
  const bool f0 = frozen0 || pAp.x == 0.0, f1 = frozen1 || pAp.y == 0.0;
  const double a0 = f0 ? 0.0 : rz.x / pAp.x, a1 = f1 ? 0.0 : rz.y / pAp.y;
  double s0 = 0.0, s1 = 0.0;
  PF_VEC_LOOP(i, n) {
    const double2 pv = ld2(p, i), av = ld2s(Ap, i);
    double2 xv = ld2s(x, i), rv = ld2s(r, i);
    if (!f0) { xv.x += a0 * pv.x; rv.x -= a0 * av.x; }
    if (!f1) { xv.y += a1 * pv.y; rv.y -= a1 * av.y; }
    st2s(x, i, xv);
    st2(r, i, rv);  // the cycle reads r next
    s0 += rv.x * rv.x; s1 += rv.y * rv.y;
  }
  block_store_partial2(s0, s1, rr_part, sh);
}

// beta_v = rz_new_v / rz_old_v; p = z + beta p
__global__ __launch_bounds__(256) void k_cg2_update_p(int64_t n, const double* __restrict__ rz_new_part, int n_new,
                                                      const double* __restrict__ rz_old_part, int n_old,
                                                      const double* __restrict__ z, double* __restrict__ p) {
  __shared__ double sh[8];
  const double2 rzn = block_sum_partials2(rz_new_part, n_new, sh);
  const double2 rzo = block_sum_partials2(rz_old_part, n_old, sh);
  const double b0 = rzo.x != 0.0 ? rzn.x / rzo.x : 0.0, b1 = rzo.y != 0.0 ? rzn.y / rzo.y : 0.0;
  PF_VEC_LOOP(i, n) {
    const double2 zv = ld2(z, i);
    double2 pv = ld2(p, i);
    pv.x = zv.x + b0 * pv.x; pv.y = zv.y + b1 * pv.y;
    st2(p, i, pv);
  }
}

// partials of r.z per vector (only when the cycle ends without a product: never for the hierarchies pf_cg2 accepts)
__global__ __launch_bounds__(256) void k_dot2(int64_t n, const double* __restrict__ a, const double* __restrict__ b, double* __restrict__ out_part) {
  __shared__ double sh[8];
  double s0 = 0.0, s1 = 0.0;
  PF_VEC_LOOP(i, n) { const double2 av = ld2(a, i), bv = ld2(b, i); s0 += av.x * bv.x; s1 += av.y * bv.y; }
  block_store_partial2(s0, s1, out_part, sh);
}

static void host_sum2(const double* v, int nblk, double out[2]) {
  out[0] = out[1] = 0.0;
  for (int i = 0; i < nblk; ++i) { out[0] += v[2 * i]; out[1] += v[2 * i + 1]; }
}

extern "C" int pf_cg2(pf_ctx* ctx, const double* vals, const double* b2, double* x2, double rtol, double atol, int maxiter,
                      int reuse, int32_t* iters_host, double* relres_host) {
  if (!ctx) return pf_fail(nullptr, PF_E_INVALID, "ctx is NULL");
  PF_REQUIRE(ctx->d && vals && b2 && x2 && iters_host && relres_host, "pf_cg2: no pattern / NULL argument");
  if (ctx->d != 1) return pf_fail(ctx, PF_E_UNSUPPORTED, "pf_cg2: scalar matrices (one dof per node)");
  PF_HIP(hipSetDevice(ctx->device));
  const int64_t n = ctx->nnodes, n2 = 2 * n;
  if (maxiter <= 0) maxiter = (int)std::min<int64_t>(10 * n, INT32_MAX);
  if (ctx->cgw_cap < 4 * n2) {  // r, z, p, Ap of the pair
    pf_free(ctx->cgw);
    ctx->cgw_cap = 0;
    PF_HIP(hipMalloc(&ctx->cgw, sizeof(double) * 4 * n2));
    ctx->cgw_cap = 4 * n2;
  }
  double *r = ctx->cgw, *z = r + n2, *p = z + n2, *Ap = p + n2;
  hipStream_t st = ctx->stream;
  const unsigned vg = vec_grid(n2);
  const int ns = (int)std::min<unsigned>(spmv_grid_d(n, 1), (unsigned)SPMV2_MAXGRID);
  {
    const int arc = pf_amg_setup(ctx, vals, reuse);
    if (arc == PF_E_UNSUPPORTED) return pf_fail(ctx, arc, "pf_cg2: the matrix is not positive definite");
    if (arc) return arc;
  }
  struct MultiStream {
    pf_ctx* c;
    bool was;
    explicit MultiStream(pf_ctx* c_) : c(c_), was(c_->multi_stream) { c->multi_stream = true; }
    ~MultiStream() { c->multi_stream = was; }
  } multi_stream_guard(ctx);
  int rc = pf_launch_spmv(ctx, vals, x2, Ap, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 2);  // Ap = A x0
  if (rc) return rc;
  // partial arrays (pairs): PA_BB b.b, PA_RR |r|^2, PA_RZ0 / PA_RZ1 r.z ping-pong, PA_PAP p.Ap
  k_cg2_init<<<vg, 256, 0, st>>>(n, b2, Ap, r, part(ctx, PA_BB), part(ctx, PA_RR));
  PF_LAUNCH_CHECK();
  if ((rc = readback_queue(ctx, part(ctx, PA_RR), 2 * NPART))) return rc;  // RR | BB, awaited behind the first cycle
  int nrz = ns;
  auto cycle = [&](double* rz_part) -> int {
    bool dotted = false;
    const int e = pf_amg_apply(ctx, r, z, rz_part, &dotted, 2);
    if (e == PF_E_UNSUPPORTED) return pf_fail(ctx, e, "pf_cg2: the hierarchy has a level the two-vector cycle does not cover");
    if (e) return e;
    if (dotted) nrz = ns;
    else { k_dot2<<<vg, 256, 0, st>>>(n, r, z, rz_part); nrz = (int)vg; }
    return PF_OK;
  };
  if ((rc = cycle(part(ctx, PA_RZ0)))) return rc;
  PF_HIP(hipMemcpyAsync(p, z, sizeof(double) * n2, hipMemcpyDeviceToDevice, st));
  if ((rc = readback_wait(ctx))) return rc;
  double bb[2], rr[2], rr_prev[2], tol[2], bnorm[2];
  host_sum2(ctx->pin + NPART, (int)vg, bb);
  host_sum2(ctx->pin, (int)vg, rr);
  rr_prev[0] = rr[0]; rr_prev[1] = rr[1];
  for (int v = 0; v < 2; ++v) {
    bnorm[v] = sqrt(bb[v]);
    tol[v] = std::max(rtol * bnorm[v], atol);
    iters_host[v] = 0;
  }
  bool done[2] = {!(sqrt(rr[0]) > tol[0]), !(sqrt(rr[1]) > tol[1])};
  int it = 0, cur = 0;
  bool broke = false;
  auto look = [&]() {  // the residuals of step `it`, just read back
    double now[2];
    host_sum2(ctx->pin, (int)vg, now);
    for (int v = 0; v < 2; ++v) {
      rr_prev[v] = rr[v];
      rr[v] = now[v];
      if (!(rr[v] == rr[v])) broke = true;  // NaN: breakdown
      if (!done[v]) {
        iters_host[v] = it;
        done[v] = !(sqrt(rr[v]) > tol[v]);
      }
    }
  };
  while (!(done[0] && done[1]) && it < maxiter && !broke) {
    double* rz_old = part(ctx, cur ? PA_RZ1 : PA_RZ0);
    double* rz_new = part(ctx, cur ? PA_RZ0 : PA_RZ1);
    // as in pf_cg: look before queuing the cycle when this step probably ends the solve, afterwards otherwise
    bool early = true;
    for (int v = 0; v < 2; ++v)
      if (!done[v]) {
        const double rn = sqrt(rr[v]), rp = sqrt(rr_prev[v]);
        early = early && rn * std::min(1.0, rn / rp) <= 4.0 * tol[v];
      }
    rc = pf_launch_spmv(ctx, vals, p, Ap, part(ctx, PA_PAP), nullptr, nullptr, nullptr, nullptr, nullptr, 2);  // Ap = A p, p.Ap
    if (rc) return rc;
    const int nrz_old = nrz;
    k_cg2_update_xr<<<vg, 256, 0, st>>>(n, rz_old, nrz_old, part(ctx, PA_PAP), ns, done[0] ? 1 : 0, done[1] ? 1 : 0, p, Ap, x2, r,
                                        part(ctx, PA_RR));
    if ((rc = readback_queue(ctx, part(ctx, PA_RR), 2 * (size_t)vg))) return rc;
    ++it;
    if (early) {
      if ((rc = readback_wait(ctx))) return rc;
      look();
      if ((done[0] && done[1]) || broke || it >= maxiter) break;
    }
    if ((rc = cycle(rz_new))) return rc;
    k_cg2_update_p<<<vg, 256, 0, st>>>(n, rz_new, nrz, rz_old, nrz_old, z, p);
    PF_LAUNCH_CHECK();
    cur ^= 1;
    if (!early) {
      if ((rc = readback_wait(ctx))) return rc;
      look();
    }
  }
  for (int v = 0; v < 2; ++v) relres_host[v] = bnorm[v] > 0 ? sqrt(rr[v]) / bnorm[v] : sqrt(rr[v]);
  if (broke || !(done[0] && done[1]))
    return pf_fail(ctx, PF_E_NOCONV, "cg failed with code " + std::to_string(it));  // pyfem.py:2421-2422
  return PF_OK;
}

// ------------------------------------------------------------------------------------------
// Distributed single-reduction (Chronopoulos-Gear) Jacobi-PCG, the whole loop in C (dist.py dist_cg is the same algorithm
// driven from Python over torch.distributed, kept for the gloo CPU tests and as the fallback transport).
// TWO launches per iteration (round 3; round 2: four, each behind the other):
//   k_spmv<..., SEND>   w = K z with the partials of z.Kz; the tiles that hold interface nodes run first and store their
//                       rows straight into the neighbours' windows, the interior follows while the rows travel
//   k_dcg_update        scalars to every rank, wait, sum (the iteration's all-reduce); interface rows summed in rank
//                       order; p, s, x, r, z updated in one pass; convergence tested ON THE DEVICE (every rank sees the
//                       same bits), after which the remaining launches of a burst return at once and x stays the
//                       converged iterate -- the result does not depend on check_every
// Nothing touches the host except the read-back every check_every iterations.  A plan that was not fused
// (pf_halo_plan_fuse) is an error: there is no second path.
// ------------------------------------------------------------------------------------------
// The two launches, for pf_dist_cg below and pf_dist_pcg_amg (pf_dist.hip):
//   pf_dist_product  y = K x on the rank's rows with the fused send (optional smoother epilogue / fp32 values / partials
//                    of x.y into PA_DELTA); interface entries of y hold the RAW row until the consumer has summed them
//   pf_dist_update   k_dcg_update on the context's partial arrays.  gamma_from_dot: the r.z partials were left in
//                    PA_GAMMA by k_dot (vec_grid blocks, AMG-PCG) instead of by the previous update (its own grid)
int pf_dist_product(pf_ctx* ctx, pf_halo* halo, int plan, const double* vals, const float* vals32, const double* x, double* y,
                    bool dot, const double* sm_dinv, const double* sm_b, const double* stop) {
  const pf_halo_fuse* F = pf_halo_fuse_of(halo, plan);
  PF_REQUIRE(F && F->d == ctx->d, "fused product: the plan was not prepared (pf_halo_plan_fuse)");
  pf_spmv_send snd;
  if (int rc = pf_halo_fused_send(halo, plan, &snd)) return rc;
  snd.stop = stop;
  if (int rc = pf_launch_spmv(ctx, vals, x, y, dot ? part(ctx, PA_DELTA) : nullptr, sm_dinv, sm_b, vals32, &snd, F->tperm)) return rc;
  pf_halo_fused_sent(halo, plan);
  return PF_OK;
}

static unsigned dist_update_grid(pf_ctx* ctx, pf_halo* halo, int64_t n) {
  // the update kernel's blocks wait (for scalars, for rows): when several ranks share ONE device (the rehearsal of the
  // tests; never in production) they must all fit beside each other's product kernels
  return std::max(1u, std::min(vec_grid(n), (unsigned)(4 * std::max(ctx->ncu, 1)) / (unsigned)std::max(pf_halo_colocated(halo), 1)));
}

int pf_dist_update(pf_ctx* ctx, pf_halo* halo, int plan, const double* w, const double* dinv, bool keep_z, bool gamma_from_dot,
                   bool rr_from_init, double* x, double* r, double* z, double* p, double* s, const double* state_in,
                   double* state_out, double* redg, bool first, double tol2, int it) {
  const pf_halo_fuse* F = pf_halo_fuse_of(halo, plan);
  PF_REQUIRE(F && F->d == ctx->d, "fused update: the plan was not prepared (pf_halo_plan_fuse)");
  const int64_t n = (int64_t)ctx->d * ctx->nnodes;
  const unsigned vg = vec_grid(n), ug = dist_update_grid(ctx, halo, n);
  const int ns = (int)spmv_grid_d(ctx->nnodes, ctx->d);
  pf_halo_recv rcv;
  pf_halo_fused_recv(halo, plan, true, &rcv);
  // partial arrays: the r.z / r.r sets alternate (ctx->dcg_par): this launch reads what the previous producer stored
  // (k_cgcg_update_xrz or k_dot: vg blocks; this kernel: ug blocks) and writes the OTHER set, because all its blocks sum the
  // input arrays on their own (see the kernel).  keep_z: r.z comes from k_dot into the set this launch reads, the kernel's
  // own r.z partials go to PA_RZ0, which nobody reads.
  const int gin = ctx->dcg_par ? PA_RZ1 : PA_GAMMA, gout = ctx->dcg_par ? PA_GAMMA : PA_RZ1;
  const int rin = ctx->dcg_par ? PA_RR : PA_RR2, rout = ctx->dcg_par ? PA_RR2 : PA_RR;
  k_dcg_update<<<ug, 256, 0, ctx->stream>>>(n, rcv, part(ctx, gin), (int)((gamma_from_dot || rr_from_init) ? vg : ug),
                                            part(ctx, rin), (int)(rr_from_init ? vg : ug), part(ctx, PA_DELTA), ns, state_in,
                                            state_out, redg, first ? 1 : 0, tol2, it, keep_z ? 1 : 0, w, dinv, F->flags8, z, p, s, x, r,
                                            part(ctx, keep_z ? PA_RZ0 : gout), part(ctx, rout));
  PF_LAUNCH_CHECK();
  ctx->dcg_par ^= 1;
  pf_halo_fused_scal(halo);
  return PF_OK;
}

// r.z over the owned dofs into PA_GAMMA (vec_grid partials), for pf_dist_update(gamma_from_dot)
int pf_dist_dot_gamma(pf_ctx* ctx, const double* r, const double* z, const double* own_w) {
  const int64_t n = (int64_t)ctx->d * ctx->nnodes;
  k_dot<<<vec_grid(n), 256, 0, ctx->stream>>>(n, r, z, own_w, part(ctx, ctx->dcg_par ? PA_RZ1 : PA_GAMMA));  // the set the next update reads
  PF_LAUNCH_CHECK();
  return PF_OK;
}

// z = dinv r (dinv == NULL: z = r) and the partials of own r.z, r.r for the first reduction, with the zeroed state
int pf_dist_init_partials(pf_ctx* ctx, const double* state0, const double* p, const double* s, const double* dinv,
                          const double* own_w, double* x, double* r, double* z) {
  const int64_t n = (int64_t)ctx->d * ctx->nnodes;
  ctx->dcg_par = 0;  // a new solve: the first update reads PA_GAMMA / PA_RR2
  k_cgcg_update_xrz<<<vec_grid(n), 256, 0, ctx->stream>>>(n, state0, p, s, dinv, own_w, x, r, z, part(ctx, PA_GAMMA), part(ctx, PA_RR2));
  PF_LAUNCH_CHECK();
  return PF_OK;
}

// y = K x over the partition, complete on return of the stream: the fused product (interface rows sent from its epilogue)
// followed by the ordered interface sums.  The building block the loops below are made of, exported for the self-test of
// the fused exchange (dist.py compares it bit for bit with pf_spmv + the packed exchange) and for callers' own loops.
extern "C" int pf_dist_spmv(pf_ctx* ctx, pf_halo* halo, int plan, const double* vals, const double* x, double* y) {
  if (!ctx) return pf_fail(nullptr, PF_E_INVALID, "ctx is NULL");
  PF_REQUIRE(halo && pf_halo_ctx(halo) == ctx, "pf_dist_spmv: the halo belongs to another context");
  PF_REQUIRE(ctx->d && vals && x && y && x != y, "pf_dist_spmv: no pattern / NULL argument / x aliases y");
  PF_HIP(hipSetDevice(ctx->device));
  if (int rc = pf_dist_product(ctx, halo, plan, vals, nullptr, x, y, false, nullptr, nullptr, nullptr)) return rc;
  return pf_halo_recv_apply(halo, plan, nullptr, nullptr, nullptr, y);
}

extern "C" int pf_dist_cg(pf_ctx* ctx, pf_halo* halo, int plan, const double* vals, const double* b, const double* dinv,
                          const double* own_w, double* x, double rtol, double atol, int maxiter, int check_every,
                          int32_t* iters_host, double* relres_host) {
  if (!ctx) return pf_fail(nullptr, PF_E_INVALID, "ctx is NULL");
  PF_REQUIRE(halo && pf_halo_ctx(halo) == ctx, "pf_dist_cg: the halo belongs to another context");
  PF_REQUIRE(ctx->d && vals && b && x && own_w, "pf_dist_cg: no pattern / NULL argument");
  const pf_halo_fuse* F = pf_halo_fuse_of(halo, plan);
  PF_REQUIRE(F && F->d == ctx->d, "pf_dist_cg: the plan was not prepared for the fused exchange (pf_halo_plan_fuse)");
  PF_HIP(hipSetDevice(ctx->device));
  const int64_t n = (int64_t)ctx->d * ctx->nnodes;
  if (maxiter <= 0) maxiter = (int)std::min<int64_t>(10 * n, INT32_MAX);
  if (check_every <= 0) check_every = 50;
  const int64_t need = 5 * n + 64;
  if (ctx->cgw_cap < need) {
    pf_free(ctx->cgw);
    PF_HIP(hipMalloc(&ctx->cgw, sizeof(double) * need));
    ctx->cgw_cap = need;
  }
  double *r = ctx->cgw, *z = r + n, *p = z + 2 * n, *s = p + n;
  if (int prc = cg_place_ap(ctx, vals, p, n)) return prc;  // the product's output vector, placed by measurement
  double* w = ctx->cg_ap;
  double* small = s + n;  // [0..2] state a, [8..10] state b, [16..23] local scalars, [24..31] reduced scalars + stop flag
  double *stA = small, *stB = small + 8, *red = small + 16, *redg = small + 24;
  hipStream_t st = ctx->stream;
  const unsigned vg = vec_grid(n);
  PF_HIP(hipMemsetAsync(x, 0, sizeof(double) * n, st));
  PF_HIP(hipMemsetAsync(p, 0, sizeof(double) * 2 * n + sizeof(double) * 64, st));  // p, s and the small block
  PF_HIP(hipMemcpyAsync(r, b, sizeof(double) * n, hipMemcpyDeviceToDevice, st));
  // ||b||^2 over the owned dofs, summed over ranks
  k_dot<<<vg, 256, 0, st>>>(n, b, b, own_w, part(ctx, PA_BB));
  k_finalize<<<1, 256, 0, st>>>(1, part(ctx, PA_BB), (int)vg, nullptr, 0, nullptr, 0, red, nullptr, 0);
  int rc = pf_halo_exchange_impl(halo, -1, nullptr, red, 1, redg);
  if (rc) return rc;
  double h[8];
  int32_t herr = 0;
  PF_HIP(hipMemcpyAsync(h, redg, sizeof(double), hipMemcpyDeviceToHost, st));
  PF_HIP(hipMemcpyAsync(&herr, pf_halo_err_ptr(halo), sizeof(herr), hipMemcpyDeviceToHost, st));
  PF_HIP(hipStreamSynchronize(st));
  if (herr) return pf_fail(ctx, PF_E_HIP, "pf_dist_cg: a peer did not arrive at the halo exchange (timeout)");
  const double bnorm = sqrt(h[0]);
  const double tol = std::max(rtol * bnorm, atol);
  PF_HIP(hipMemsetAsync(redg, 0, sizeof(double) * 8, st));
  double rnorm = INFINITY;
  int it = 0;
  double *state = stA, *state_next = stB;
  if ((rc = pf_dist_init_partials(ctx, state, p, s, dinv, own_w, x, r, z))) return rc;
  bool stopped = false;
  while (it < maxiter && !stopped) {
    const int burst = std::min(check_every - it % check_every, maxiter - it);
    for (int k = 0; k < burst; ++k, ++it) {
      if ((rc = pf_dist_product(ctx, halo, plan, vals, nullptr, z, w, true, nullptr, nullptr, redg + 3))) return rc;
      if ((rc = pf_dist_update(ctx, halo, plan, w, dinv, false, false, it == 0, x, r, z, p, s, state, state_next, redg, it == 0,
                               tol * tol, it)))
        return rc;
      std::swap(state, state_next);
    }
    // redg = [gamma, delta, r.r after redg[4] updates, stop flag, iteration index]
    PF_HIP(hipMemcpyAsync(h, redg, sizeof(double) * 5, hipMemcpyDeviceToHost, st));
    PF_HIP(hipMemcpyAsync(&herr, pf_halo_err_ptr(halo), sizeof(herr), hipMemcpyDeviceToHost, st));
    PF_HIP(hipStreamSynchronize(st));
    if (herr) return pf_fail(ctx, PF_E_HIP, "pf_dist_cg: a peer did not arrive at the fused exchange (timeout)");
    rnorm = sqrt(h[2]);
    if (h[3] != 0.0) {  // converged on the device after h[4] updates: the later launches of the burst were no-ops
      it = (int)h[4];
      stopped = true;
    }
    if (!(rnorm == rnorm)) break;
  }
  if (iters_host) *iters_host = it;
  if (relres_host) *relres_host = bnorm > 0 ? rnorm / bnorm : rnorm;
  if (!(rnorm <= tol)) return pf_fail(ctx, PF_E_NOCONV, "cg failed with code " + std::to_string(it));  // pyfem.py:2421-2422
  return PF_OK;
}

// ------------------------------------------------------------------------------------------
// BiCGStab for nonsymmetric matrices on the ctx pattern: the Newton tangent of NonlinearPoisson2D and any K whose
// Dirichlet conditions were applied to the rows only (enforce_symmetric_K=False) -- the systems the reference hands to
// scipy gmres / spsolve (pyfem.py:2340-2349, 2414-2420).  Right-preconditioned with the Jacobi diagonal of vals or with
// the AMG V-cycle of a symmetric positive definite companion on the same pattern (pvals: e.g. the tangent's SPD part).
// Systems of this kind are small next to the PCG's; scalars are read back every step (four syncs per iteration).
// ------------------------------------------------------------------------------------------
// out = a x + b y + c z (z may be NULL); out may alias x or y
__global__ __launch_bounds__(256) void k_lin3(int64_t n, double* out, double a, const double* x, double b, const double* y,
                                              double c, const double* z) {
  PF_VEC_LOOP(i, n) out[i] = a * x[i] + b * y[i] + (z ? c * z[i] : 0.0);
}

__global__ __launch_bounds__(256) void k_mul(int64_t n, double* __restrict__ out, const double* __restrict__ d,
                                             const double* __restrict__ x) {
  PF_VEC_LOOP(i, n) out[i] = d[i] * x[i];
}

extern "C" int pf_bicgstab(pf_ctx* ctx, const double* vals, const double* pvals, const double* b, double* x, double rtol,
                           double atol, int maxiter, int precond, int32_t* iters_host, double* relres_host) {
  if (!ctx) return pf_fail(nullptr, PF_E_INVALID, "ctx is NULL");
  PF_REQUIRE(ctx->d && vals && b && x, "pf_bicgstab: no pattern / NULL argument");
  PF_REQUIRE(precond == PF_PRECOND_NONE || precond == PF_PRECOND_JACOBI || precond == PF_PRECOND_AMG,
             "pf_bicgstab: unknown preconditioner");
  PF_REQUIRE(precond != PF_PRECOND_AMG || pvals, "pf_bicgstab: the AMG needs the symmetric positive definite companion");
  PF_HIP(hipSetDevice(ctx->device));
  const int64_t n = (int64_t)ctx->d * ctx->nnodes;
  if (maxiter <= 0) maxiter = (int)std::min<int64_t>(10 * n, INT32_MAX);
  if (ctx->cgw_cap < 9 * n) {
    pf_free(ctx->cgw);
    PF_HIP(hipMalloc(&ctx->cgw, sizeof(double) * 9 * n));
    ctx->cgw_cap = 9 * n;
  }
  double *r = ctx->cgw, *rh = r + n, *p = rh + n, *v = p + n, *sv = v + n, *tv = sv + n, *y = tv + n, *z = y + n,
         *dinv = z + n;
  hipStream_t st = ctx->stream;
  const unsigned vg = vec_grid(n);
  if (precond == PF_PRECOND_AMG) {
    const int arc = pf_amg_setup(ctx, pvals, 0);
    if (arc == PF_E_UNSUPPORTED) {
      precond = PF_PRECOND_JACOBI;
      ctx->warn = "pf_bicgstab: the companion matrix is not definite; the AMG preconditioner was refused and Jacobi is used";
    } else if (arc) return arc;
  }
  if (precond == PF_PRECOND_JACOBI) {
    const unsigned g = (unsigned)cdiv(ctx->nnodes, 256);
    if (ctx->d == 1) k_extract_diag<1><<<g, 256, 0, st>>>(ctx->nnodes, ctx->nptr, ctx->diagpos, vals, dinv, 1);
    else if (ctx->d == 3) k_extract_diag<3><<<g, 256, 0, st>>>(ctx->nnodes, ctx->nptr, ctx->diagpos, vals, dinv, 1);
    else k_extract_diag<2><<<g, 256, 0, st>>>(ctx->nnodes, ctx->nptr, ctx->diagpos, vals, dinv, 1);
    PF_LAUNCH_CHECK();
  }
  auto apply_M = [&](const double* in, double* out) -> int {
    if (precond == PF_PRECOND_AMG) return pf_amg_apply(ctx, in, out);
    if (precond == PF_PRECOND_JACOBI) k_mul<<<vg, 256, 0, st>>>(n, out, dinv, in);
    else PF_HIP(hipMemcpyAsync(out, in, sizeof(double) * n, hipMemcpyDeviceToDevice, st));
    return PF_OK;
  };
  double h[2];
  auto dots = [&](const double* a0, const double* b0, const double* a1, const double* b1) -> int {  // h[0], h[1]
    k_dot<<<vg, 256, 0, st>>>(n, a0, b0, nullptr, part(ctx, PA_PAP));
    if (a1) k_dot<<<vg, 256, 0, st>>>(n, a1, b1, nullptr, part(ctx, PA_RR));
    k_finalize<<<1, 256, 0, st>>>(a1 ? 2 : 1, part(ctx, PA_PAP), (int)vg, part(ctx, PA_RR), (int)vg, nullptr, 0,
                                  part(ctx, PA_BB), nullptr, 0);
    PF_HIP(hipMemcpyAsync(h, part(ctx, PA_BB), sizeof(double) * 2, hipMemcpyDeviceToHost, st));
    PF_HIP(hipStreamSynchronize(st));
    return PF_OK;
  };
  int rc = pf_launch_spmv(ctx, vals, x, v, nullptr);  // v = A x0
  if (rc) return rc;
  k_lin3<<<vg, 256, 0, st>>>(n, r, 1.0, b, -1.0, v, 0.0, nullptr);
  PF_HIP(hipMemcpyAsync(rh, r, sizeof(double) * n, hipMemcpyDeviceToDevice, st));
  PF_HIP(hipMemsetAsync(p, 0, sizeof(double) * n, st));
  PF_HIP(hipMemsetAsync(v, 0, sizeof(double) * n, st));
  if ((rc = dots(b, b, r, r))) return rc;
  const double bnorm = sqrt(h[0]);
  const double tol = std::max(rtol * bnorm, atol);
  double rnorm = sqrt(h[1]), rho = 1.0, alpha = 1.0, omega = 1.0;
  int it = 0;
  while (rnorm > tol && it < maxiter) {
    if ((rc = dots(rh, r, nullptr, nullptr))) return rc;
    const double rho_new = h[0];
    if (rho_new == 0.0 || !(rho_new == rho_new)) break;  // breakdown
    const double beta = (rho_new / rho) * (alpha / omega);
    k_lin3<<<vg, 256, 0, st>>>(n, p, 1.0, r, beta, p, -beta * omega, v);
    if ((rc = apply_M(p, y))) return rc;
    if ((rc = pf_launch_spmv(ctx, vals, y, v, nullptr))) return rc;
    if ((rc = dots(rh, v, nullptr, nullptr))) return rc;
    alpha = rho_new / h[0];
    k_lin3<<<vg, 256, 0, st>>>(n, sv, 1.0, r, -alpha, v, 0.0, nullptr);
    if ((rc = apply_M(sv, z))) return rc;
    if ((rc = pf_launch_spmv(ctx, vals, z, tv, nullptr))) return rc;
    if ((rc = dots(tv, sv, tv, tv))) return rc;
    omega = h[1] > 0.0 ? h[0] / h[1] : 0.0;
    k_lin3<<<vg, 256, 0, st>>>(n, x, 1.0, x, alpha, y, omega, z);
    k_lin3<<<vg, 256, 0, st>>>(n, r, 1.0, sv, -omega, tv, 0.0, nullptr);
    PF_LAUNCH_CHECK();
    if ((rc = dots(r, r, nullptr, nullptr))) return rc;
    rnorm = sqrt(h[0]);
    rho = rho_new;
    ++it;
    if (!(rnorm == rnorm) || omega == 0.0) break;
  }
  if (iters_host) *iters_host = it;
  if (relres_host) *relres_host = bnorm > 0 ? rnorm / bnorm : rnorm;
  if (!(rnorm <= tol))
    return pf_fail(ctx, PF_E_NOCONV, "bicgstab failed with code " + std::to_string(it));  // pyfem.py:2421-2422
  return PF_OK;
}

