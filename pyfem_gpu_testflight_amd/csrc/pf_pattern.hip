// Context, mesh upload and the SYMBOLIC phase (sparsity pattern + gather maps), all on device.
//
// Replaces, for the accelerated path, the reference's host-side index bookkeeping:
//   utils.create_dof (utils.py:268-298), ModelBase._compute_nz_pattern (pyfem.py:838-858) and
//   the structure half of coo_matrix(...).tocsr() (pyfem.py:930-931).
// The reference materialises 2*E*D^2 int64 COO indices (2.1 GB each at 4M quads) and lets scipy
// sort them on every assembly.  Here the pattern is derived once from node->element incidence:
//   1. count/scan/fill the node -> (element, local index) incidence lists (sorted ascending, so
//      every later summation order is fixed = deterministic assembly),
//   2. per node, merge the nodes of its incident elements into a sorted unique neighbour row
//      (this is exactly the sorted, duplicate-free column set scipy's tocsr() ends up with),
//   3. per (node, incident element, element node) store the 1-byte position of that element
//      node inside the row: the numeric phase then needs no search and no atomics.
// One thread per node; rows are short (9 for quads, ~7 for triangles) so the work is linear.
#include <hipcub/hipcub.hpp>

#include "pf_common.h"
#include <vector>

thread_local std::string g_pf_err;

static constexpr int MAXNB = 64;  // max neighbours (incl. self) of one node; pos fits a byte

// ------------------------------------------------------------------------------------------
__global__ void k_minmax_i32(const int32_t* __restrict__ v, int64_t n, int32_t* __restrict__ mm) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int32_t lo = INT32_MAX, hi = INT32_MIN;
  for (; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    int32_t x = v[i];
    lo = min(lo, x);
    hi = max(hi, x);
  }
  for (int o = 32; o > 0; o >>= 1) {
    lo = min(lo, __shfl_xor(lo, o));
    hi = max(hi, __shfl_xor(hi, o));
  }
  if ((threadIdx.x & 63) == 0) {
    atomicMin(&mm[0], lo);
    atomicMax(&mm[1], hi);
  }
}

__global__ void k_count_incidence(const int32_t* __restrict__ conn, int64_t total, int32_t* __restrict__ cnt) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < total) atomicAdd(&cnt[conn[i]], 1);
}

__global__ void k_fill_incidence(const int32_t* __restrict__ conn, int64_t total, const int32_t* __restrict__ ptr,
                                 int32_t* __restrict__ cursor, int32_t* __restrict__ n2e) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < total) {
    int32_t n = conn[i];
    int32_t s = atomicAdd(&cursor[n], 1);
    n2e[ptr[n] + s] = (int32_t)i;  // i == e*P + a
  }
}

// sort each node's incidence list ascending (lists are a handful of entries)
__global__ void k_sort_incidence(int64_t nnodes, const int32_t* __restrict__ ptr, int32_t* __restrict__ n2e) {
  int64_t n = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (n >= nnodes) return;
  int32_t b = ptr[n], e = ptr[n + 1];
  for (int32_t i = b + 1; i < e; ++i) {
    int32_t v = n2e[i];
    int32_t j = i - 1;
    while (j >= b && n2e[j] > v) {
      n2e[j + 1] = n2e[j];
      --j;
    }
    n2e[j + 1] = v;
  }
}

// sorted-unique merge of the nodes of all incident elements of node n into nb[]; returns count
// (or -1 on overflow)
template <int P>
__device__ inline int collect_row(int64_t n, const int32_t* __restrict__ conn, const int32_t* __restrict__ n2e_ptr,
                                  const int32_t* __restrict__ n2e, int32_t* nb) {
  int cnt = 0;
  for (int32_t j = n2e_ptr[n]; j < n2e_ptr[n + 1]; ++j) {
    int32_t e = n2e[j] / P;
    for (int b = 0; b < P; ++b) {
      int32_t m = conn[(int64_t)e * P + b];
      int k = 0;
      while (k < cnt && nb[k] < m) ++k;
      if (k < cnt && nb[k] == m) continue;
      if (cnt == MAXNB) return -1;
      for (int t = cnt; t > k; --t) nb[t] = nb[t - 1];
      nb[k] = m;
      ++cnt;
    }
  }
  return cnt;
}

template <int P>
__global__ void k_row_count(int64_t nnodes, const int32_t* __restrict__ conn, const int32_t* __restrict__ n2e_ptr,
                            const int32_t* __restrict__ n2e, int32_t* __restrict__ rowcnt, int32_t* __restrict__ flag) {
  int64_t n = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (n >= nnodes) return;
  int32_t nb[MAXNB];
  int c = collect_row<P>(n, conn, n2e_ptr, n2e, nb);
  if (c < 0) {
    atomicOr(flag, 1);
    c = 0;
  }
  rowcnt[n] = c;
}

template <int P>
__global__ void k_row_fill(int64_t nnodes, const int32_t* __restrict__ conn, const int32_t* __restrict__ n2e_ptr,
                           const int32_t* __restrict__ n2e, const int32_t* __restrict__ nptr,
                           int32_t* __restrict__ nidx, uint8_t* __restrict__ n2e_pos, uint8_t* __restrict__ diagpos) {
  int64_t n = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (n >= nnodes) return;
  int32_t nb[MAXNB];
  int c = collect_row<P>(n, conn, n2e_ptr, n2e, nb);
  int32_t base = nptr[n];
  for (int k = 0; k < c; ++k) {
    nidx[base + k] = nb[k];
    if (nb[k] == (int32_t)n) diagpos[n] = (uint8_t)k;
  }
  for (int32_t j = n2e_ptr[n]; j < n2e_ptr[n + 1]; ++j) {
    int32_t e = n2e[j] / P;
    for (int b = 0; b < P; ++b) {
      int32_t m = conn[(int64_t)e * P + b];
      int k = 0;
      while (nb[k] != m) ++k;
      n2e_pos[(int64_t)j * (P <= 4 ? 4 : 8) + b] = (uint8_t)k;
    }
  }
}

// Rows for elements outside the three hard-wired shapes (10-node tetrahedra, 20-node bricks, ...: the table-driven
// path of pf_tabular.hip): runtime P, up to MAXNB_TAB neighbours (a corner shared by eight 20-node bricks has 81),
// no byte-position tables -- that path finds positions by bisection.
static constexpr int MAXNB_TAB = 160;

__device__ inline int collect_row_tab(int P, int64_t n, const int32_t* __restrict__ conn,
                                      const int32_t* __restrict__ n2e_ptr, const int32_t* __restrict__ n2e, int32_t* nb) {
  int cnt = 0;
  for (int32_t j = n2e_ptr[n]; j < n2e_ptr[n + 1]; ++j) {
    const int32_t e = n2e[j] / P;
    for (int b = 0; b < P; ++b) {
      const int32_t m = conn[(int64_t)e * P + b];
      int k = 0;
      while (k < cnt && nb[k] < m) ++k;
      if (k < cnt && nb[k] == m) continue;
      if (cnt == MAXNB_TAB) return -1;
      for (int t = cnt; t > k; --t) nb[t] = nb[t - 1];
      nb[k] = m;
      ++cnt;
    }
  }
  return cnt;
}

__global__ void k_row_count_tab(int P, int64_t nnodes, const int32_t* __restrict__ conn,
                                const int32_t* __restrict__ n2e_ptr, const int32_t* __restrict__ n2e,
                                int32_t* __restrict__ rowcnt, int32_t* __restrict__ flag) {
  const int64_t n = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (n >= nnodes) return;
  int32_t nb[MAXNB_TAB];
  int c = collect_row_tab(P, n, conn, n2e_ptr, n2e, nb);
  if (c < 0) {
    atomicOr(flag, 1);
    c = 0;
  }
  rowcnt[n] = c;
}

__global__ void k_row_fill_tab(int P, int64_t nnodes, const int32_t* __restrict__ conn,
                               const int32_t* __restrict__ n2e_ptr, const int32_t* __restrict__ n2e,
                               const int32_t* __restrict__ nptr, int32_t* __restrict__ nidx,
                               uint8_t* __restrict__ diagpos) {
  const int64_t n = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (n >= nnodes) return;
  int32_t nb[MAXNB_TAB];
  const int c = collect_row_tab(P, n, conn, n2e_ptr, n2e, nb);
  const int32_t base = nptr[n];
  for (int k = 0; k < c; ++k) {
    nidx[base + k] = nb[k];
    if (nb[k] == (int32_t)n) diagpos[n] = (uint8_t)k;
  }
}

// visit-ordered element records for the assembly kernel, ONE 16-byte load per incident pair jj of node n.
// The element is ROTATED so that n is its local node 0 (a cyclic rotation of a triangle or quad keeps orientation
// and geometry, and the 1-point / 2x2 Gauss rules are invariant under it: only the order in which the quadrature
// points are visited changes).  With k counting from n around the element, (a + k) mod P being the original index:
//   {o1, o2, o3, w}   o_k = node id of rotated node k (k = 1..P-1; unused slots 0)
//   w = pos_0 | pos_1<<6 | pos_2<<12 | pos_3<<18 | a<<24 | (slot of n in its tile)<<26
// (pos_k = position of rotated node k in n's row; a = original local index of n, kept for readers that want it).
// The kernel then never selects on a: the lane's own node is always local node 0 (constant shape-function signs,
// no index shuffling of the gathered coordinates).
template <int P>
__global__ void k_build_visits(int64_t nnodes, const uint8_t* __restrict__ slot, const int32_t* __restrict__ conn,
                               const int32_t* __restrict__ n2e_ptr, const int32_t* __restrict__ n2e,
                               const uint8_t* __restrict__ n2e_pos, int4* __restrict__ vis_rec) {
  int64_t n = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (n >= nnodes) return;
  for (int32_t jj = n2e_ptr[n]; jj < n2e_ptr[n + 1]; ++jj) {
    const int32_t ea = n2e[jj];
    const int32_t e = ea / P;
    const int a = ea - e * P;
    int32_t o[3] = {0, 0, 0};
    uint32_t w = ((uint32_t)a << 24) | ((uint32_t)slot[n] << 26);
    for (int k = 0; k < P; ++k) {
      const int b = (a + k) % P;
      if (k > 0) o[k - 1] = conn[(int64_t)e * P + b];
      w |= (uint32_t)n2e_pos[(int64_t)jj * 4 + b] << (6 * k);
    }
    vis_rec[jj] = make_int4(o[0], o[1], o[2], (int32_t)w);
  }
}

// Compact 8-byte visit records (k_assemble_v4<.., REC8>): the same content as the 16-byte ones with the other nodes as
// 13-bit signed offsets from the lane's own node and 5-bit row positions,
//   d0 = off1 | off2 << 13 | (off3 & 63) << 26
//   d1 = (off3 >> 6) & 127 | pos0 << 7 | pos1 << 12 | pos2 << 17 | pos3 << 22 | slot << 27
// Half the record stream of the quad assembly (32 instead of 64 bytes per element; r01/r02: the kernel is HBM-bound and
// moved 1.23x its algorithmic bytes, a third of the excess being this stream).  Usable when every neighbour is within
// +-4095 node ids of the node (any banded numbering: +-2050 on the 2048^2 grid) and rows have at most 32 blocks;
// otherwise flag[0] is raised and the 16-byte records stay in use.
template <int P>
__global__ void k_build_rec8(int64_t nnodes, const int32_t* __restrict__ n2e_ptr, const int4* __restrict__ vis_rec,
                             uint2* __restrict__ rec8, int32_t* __restrict__ flag) {
  int64_t n = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (n >= nnodes) return;
  for (int32_t jj = n2e_ptr[n]; jj < n2e_ptr[n + 1]; ++jj) {
    const int4 r = vis_rec[jj];
    const uint32_t w = (uint32_t)r.w;
    const int32_t o[3] = {r.x, r.y, r.z};
    uint32_t off[3] = {0, 0, 0};
    bool ok = true;
    for (int k = 0; k < P - 1; ++k) {
      const int64_t d = (int64_t)o[k] - n;
      ok = ok && d >= -4096 && d <= 4095;
      off[k] = (uint32_t)d & 8191u;
    }
    uint32_t d1 = (off[2] >> 6) & 127u;
    for (int k = 0; k < P; ++k) {
      const uint32_t pos = (w >> (6 * k)) & 63u;
      ok = ok && pos < 32u;
      d1 |= pos << (7 + 5 * k);
    }
    d1 |= ((w >> 26) & 31u) << 27;
    if (!ok) { flag[0] = 1; return; }
    rec8[jj] = make_uint2(off[0] | (off[1] << 13) | ((off[2] & 63u) << 26), d1);
  }
}

// Triangles: the 16-byte record carries an unused slot (o3).  Two compact forms for the assembly kernel:
//   12 bytes  {o1, o2, w}                                           any numbering
//    8 bytes  d0 = (off1 & 0xffff) | off2 << 16                      neighbours within +-32767 node ids, rows <= 32 blocks
//             d1 = pos0 | pos1 << 5 | pos2 << 10 | slot << 15 | a << 20
// (the record stream is 28 % of what a triangle launch moves with 16-byte records, r02).  flag[0] is raised when some
// record does not fit the 8-byte form; the caller then keeps the 12-byte array.
__global__ void k_build_rec_tri(int64_t nnodes, const int32_t* __restrict__ n2e_ptr, const int4* __restrict__ vis_rec,
                                uint2* __restrict__ rec8, uint32_t* __restrict__ rec12, int32_t* __restrict__ flag) {
  int64_t n = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (n >= nnodes) return;
  for (int32_t jj = n2e_ptr[n]; jj < n2e_ptr[n + 1]; ++jj) {
    const int4 r = vis_rec[jj];
    const uint32_t w = (uint32_t)r.w;
    rec12[3 * (int64_t)jj + 0] = (uint32_t)r.x;
    rec12[3 * (int64_t)jj + 1] = (uint32_t)r.y;
    rec12[3 * (int64_t)jj + 2] = w;
    const int64_t d1n = (int64_t)r.x - n, d2n = (int64_t)r.y - n;
    bool ok = d1n >= -32768 && d1n <= 32767 && d2n >= -32768 && d2n <= 32767;
    uint32_t d1 = (((w >> 26) & 31u) << 15) | (((w >> 24) & 3u) << 20);
    for (int k = 0; k < 3; ++k) {
      const uint32_t pos = (w >> (6 * k)) & 63u;
      ok = ok && pos < 32u;
      d1 |= (pos & 31u) << (5 * k);
    }
    if (!ok) flag[0] = 1;
    rec8[jj] = make_uint2(((uint32_t)d1n & 0xffffu) | ((uint32_t)d2n << 16), d1);
  }
}

// 8-node blocks: three int4 per incident pair jj of node n:
//   {id0..id3}, {id4..id7}  (all element nodes, n included) and {pos0..3 bytes, pos4..7 bytes, a | slot<<8, 0}
__global__ void k_build_visits8(int64_t nnodes, const uint8_t* __restrict__ slot, const int32_t* __restrict__ conn,
                                const int32_t* __restrict__ n2e_ptr, const int32_t* __restrict__ n2e,
                                const uint8_t* __restrict__ n2e_pos, int4* __restrict__ vis_rec) {
  int64_t n = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (n >= nnodes) return;
  for (int32_t jj = n2e_ptr[n]; jj < n2e_ptr[n + 1]; ++jj) {
    const int32_t ea = n2e[jj];
    const int32_t e = ea >> 3;
    const int a = ea & 7;
    const int32_t* c = conn + (int64_t)e * 8;
    const uint32_t* pp = reinterpret_cast<const uint32_t*>(n2e_pos + (int64_t)jj * 8);
    vis_rec[3 * (int64_t)jj + 0] = make_int4(c[0], c[1], c[2], c[3]);
    vis_rec[3 * (int64_t)jj + 1] = make_int4(c[4], c[5], c[6], c[7]);
    vis_rec[3 * (int64_t)jj + 2] = make_int4((int32_t)pp[0], (int32_t)pp[1], a | ((int32_t)slot[n] << 8), 0);
  }
}

// position of conn[e,b] in the row of conn[e,a]  (atomic-scatter mode only)
template <int P>
__global__ void k_e2pos(int64_t nelems, const int32_t* __restrict__ conn, const int32_t* __restrict__ nptr,
                        const int32_t* __restrict__ nidx, uint8_t* __restrict__ e2pos) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;  // over e*P*P
  if (i >= nelems * P * P) return;
  int64_t e = i / (P * P);
  int a = (int)(i % (P * P)) / P, b = (int)(i % P);
  int32_t n = conn[e * P + a], m = conn[e * P + b];
  int32_t lo = nptr[n], hi = nptr[n + 1] - 1;
  while (lo < hi) {
    int32_t mid = (lo + hi) >> 1;
    if (nidx[mid] < m) lo = mid + 1; else hi = mid;
  }
  e2pos[i] = (uint8_t)(lo - nptr[n]);
}

template <int D>
__global__ void k_csr_structure(int64_t nnodes, const int32_t* __restrict__ nptr, const int32_t* __restrict__ nidx,
                                int32_t* __restrict__ indptr, int32_t* __restrict__ indices) {
  int64_t n = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (n >= nnodes) return;
  int32_t b = nptr[n], c = nptr[n + 1] - b;
  for (int k = 0; k < D; ++k) {
    int64_t rowstart = (int64_t)D * D * b + (int64_t)k * D * c;
    if (indptr) indptr[D * n + k] = (int32_t)rowstart;
    if (indices)
      for (int t = 0; t < c; ++t)
        for (int l = 0; l < D; ++l) indices[rowstart + t * D + l] = D * nidx[b + t] + l;
  }
  if (indptr && n == nnodes - 1) indptr[D * nnodes] = D * D * nptr[nnodes];
}

template <int D>
__global__ void k_vals_to_csr(int64_t nnodes, const int32_t* __restrict__ nptr, const double* __restrict__ vals,
                              double* __restrict__ csr) {
  int64_t n = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (n >= nnodes) return;
  int32_t b = nptr[n], c = nptr[n + 1] - b;
  for (int t = 0; t < c; ++t)
    for (int k = 0; k < D; ++k)
      for (int l = 0; l < D; ++l)
        csr[(int64_t)D * D * b + (int64_t)k * D * c + t * D + l] = vals[((int64_t)b + t) * D * D + k * D + l];
}

#ifndef PF_AB_TILE_SORT
#define PF_AB_TILE_SORT 1  // tools/build_variant.sh A/B: 0 = tiles walked in node order on every mesh
#endif
// ------------------------------------------------------------------------------------------
extern "C" int pf_version(void) { return 1; }

extern "C" const char* pf_last_error(pf_ctx* ctx) { return ctx ? ctx->err.c_str() : g_pf_err.c_str(); }

extern "C" const char* pf_last_warning(pf_ctx* ctx) {
  static thread_local std::string taken;
  taken.clear();
  if (ctx) taken.swap(ctx->warn);
  return taken.c_str();
}

extern "C" int pf_ctx_create(int device, pf_ctx** out) {
  pf_ctx* ctx = nullptr;
  if (!out) return pf_fail(nullptr, PF_E_INVALID, "pf_ctx_create: out is NULL");
  int ndev = 0;
  PF_HIP(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev)
    return pf_fail(nullptr, PF_E_INVALID, "pf_ctx_create: no such device " + std::to_string(device));
  PF_HIP(hipSetDevice(device));
  ctx = new pf_ctx();
  ctx->device = device;
  if (const char* e = getenv("PF_HOST_THREADS")) ctx->host_threads = std::max(1, atoi(e));
  ctx->amg_verbose = getenv("PF_AMG_VERBOSE") != nullptr;
  (void)hipDeviceGetAttribute(&ctx->ncu, hipDeviceAttributeMultiprocessorCount, device);
  hipError_t e = hipMalloc(&ctx->scal, ((size_t)PF_NPART * PF_NPART_ARRAYS + 32) * sizeof(double));  // + trash line [0,16) + AMG scalar [16]
  if (e != hipSuccess) {
    delete ctx;
    return pf_fail(nullptr, PF_E_HIP, std::string("hipMalloc: ") + hipGetErrorString(e));
  }
  *out = ctx;
  return PF_OK;
}

int pf_pattern_free(pf_ctx* ctx) {
  pf_amg_free(ctx);
  pf_free(ctx->nptr);
  pf_free(ctx->nidx);
  pf_free(ctx->nidx16);
  ctx->nidx16_state = 0;
  pf_free(ctx->n2e_ptr);
  pf_free(ctx->n2e);
  pf_free(ctx->n2e_pos);
  pf_free(ctx->diagpos);
  pf_free(ctx->e2pos);
  pf_free(ctx->vis_rec);
  pf_free(ctx->vis_rec8);
  pf_free(ctx->vis_rec12);
  pf_free(ctx->spmv_perm);
  ctx->tiles_sorted = false;
  pf_free(ctx->tile_desc);
  pf_free(ctx->tab_order);
  ctx->tab_color_ptr.clear();
  ctx->ntiles = 0;
  ctx->npt = 0;
  ctx->d = 0;
  ctx->nblocks = 0;
  return PF_OK;
}

extern "C" int pf_ctx_destroy(pf_ctx* ctx) {
  if (!ctx) return PF_OK;
  (void)hipSetDevice(ctx->device);
  pf_pattern_free(ctx);
  pf_free(ctx->X);
  pf_free(ctx->conn);
  pf_free(ctx->scal);
  pf_free(ctx->cgw);
  pf_free(ctx->cg_ap);
  ctx->cg_ap_n = 0;
  if (ctx->pin) (void)hipHostFree(ctx->pin);
  if (ctx->pin_ev) (void)hipEventDestroy(ctx->pin_ev);
  if (ctx->scan_tmp) (void)hipFree(ctx->scan_tmp);
  delete ctx;
  return PF_OK;
}

extern "C" int pf_ctx_set_stream(pf_ctx* ctx, void* s) {
  if (!ctx) return pf_fail(nullptr, PF_E_INVALID, "ctx is NULL");
  ctx->stream = (hipStream_t)s;
  return PF_OK;
}

static int mesh_set(pf_ctx* ctx, int ndims, int64_t nnodes, const double* X, int64_t nelems, int P, const int32_t* conn,
                    int on_device) {
  PF_REQUIRE(nnodes > 0 && nelems > 0 && X && conn, "pf_mesh_set: empty mesh");
  PF_REQUIRE(nnodes < (int64_t)1 << 30 && nelems * P < (int64_t)1 << 31, "pf_mesh_set: mesh too large for int32 indices");
  PF_HIP(hipSetDevice(ctx->device));
  pf_pattern_free(ctx);
  pf_free(ctx->X);
  pf_free(ctx->conn);
  ctx->nnodes = nnodes;
  ctx->nelems = nelems;
  ctx->P = P;
  ctx->ndims = ndims;
  PF_HIP(hipMalloc(&ctx->X, sizeof(double) * ndims * nnodes));
  PF_HIP(hipMalloc(&ctx->conn, sizeof(int32_t) * P * nelems));
  hipMemcpyKind kind = on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
  PF_HIP(hipMemcpyAsync(ctx->X, X, sizeof(double) * ndims * nnodes, kind, ctx->stream));
  PF_HIP(hipMemcpyAsync(ctx->conn, conn, sizeof(int32_t) * P * nelems, kind, ctx->stream));
  // conn range check (reference asserts: pyfem.py:680-681)
  int32_t* mm = nullptr;
  PF_HIP(hipMalloc(&mm, 2 * sizeof(int32_t)));
  int32_t init[2] = {INT32_MAX, INT32_MIN};
  PF_HIP(hipMemcpyAsync(mm, init, sizeof(init), hipMemcpyHostToDevice, ctx->stream));
  k_minmax_i32<<<1024, 256, 0, ctx->stream>>>(ctx->conn, nelems * P, mm);
  int32_t got[2];
  PF_HIP(hipMemcpyAsync(got, mm, sizeof(got), hipMemcpyDeviceToHost, ctx->stream));
  PF_HIP(hipStreamSynchronize(ctx->stream));
  (void)hipFree(mm);
  if (got[0] != 0 || got[1] != nnodes - 1)
    return pf_fail(ctx, PF_E_INVALID,
                   "pf_mesh_set: conn must span 0..nnodes-1 (min " + std::to_string(got[0]) + ", max " +
                       std::to_string(got[1]) + ", nnodes " + std::to_string(nnodes) + ")");
  return PF_OK;
}

extern "C" int pf_mesh_set(pf_ctx* ctx, int64_t nnodes, const double* X, int64_t nelems, int P, const int32_t* conn,
                           int on_device) {
  if (!ctx) return pf_fail(nullptr, PF_E_INVALID, "ctx is NULL");
  PF_REQUIRE(P == PF_QUAD4 || P == PF_TRI3, "pf_mesh_set: only 4-node quads and 3-node triangles are on this path");
  return mesh_set(ctx, 2, nnodes, X, nelems, P, conn, on_device);
}

extern "C" int pf_mesh_set3d(pf_ctx* ctx, int64_t nnodes, const double* X, int64_t nelems, int P, const int32_t* conn,
                             int on_device) {
  if (!ctx) return pf_fail(nullptr, PF_E_INVALID, "ctx is NULL");
  PF_REQUIRE(P >= 4 && P <= PF_TAB_MAX_NODES, "pf_mesh_set3d: 4 to 32 nodes per element (8: the hard-wired block kernels, "
                                               "anything else: the table-driven element of pf_*_tab)");
  return mesh_set(ctx, 3, nnodes, X, nelems, P, conn, on_device);
}

static int exclusive_scan_i32(pf_ctx* ctx, int32_t* in, int32_t* out, int64_t n) {
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

static int max_i32(pf_ctx* ctx, const int32_t* v, int64_t n, int32_t* out_host) {
  int32_t* mm = nullptr;
  PF_HIP(hipMalloc(&mm, 2 * sizeof(int32_t)));
  int32_t init[2] = {INT32_MAX, INT32_MIN};
  PF_HIP(hipMemcpyAsync(mm, init, sizeof(init), hipMemcpyHostToDevice, ctx->stream));
  k_minmax_i32<<<1024, 256, 0, ctx->stream>>>(v, n, mm);
  int32_t got[2];
  PF_HIP(hipMemcpyAsync(got, mm, sizeof(got), hipMemcpyDeviceToHost, ctx->stream));
  PF_HIP(hipStreamSynchronize(ctx->stream));
  (void)hipFree(mm);
  *out_host = got[1];
  return PF_OK;
}

template <int P>
static int build_rows(pf_ctx* ctx, int32_t* rowcnt, int32_t* flag) {
  const int64_t nn = ctx->nnodes;
  const int T = 128;
  k_row_count<P><<<(unsigned)cdiv(nn, T), T, 0, ctx->stream>>>(nn, ctx->conn, ctx->n2e_ptr, ctx->n2e, rowcnt, flag);
  PF_LAUNCH_CHECK();
  int32_t f = 0;
  PF_HIP(hipMemcpyAsync(&f, flag, sizeof(f), hipMemcpyDeviceToHost, ctx->stream));
  PF_HIP(hipStreamSynchronize(ctx->stream));
  if (f) return pf_fail(ctx, PF_E_UNSUPPORTED, "pf_pattern_build: a node has more than 64 neighbours");
  int rc = exclusive_scan_i32(ctx, rowcnt, ctx->nptr, nn + 1);
  if (rc) return rc;
  int32_t total = 0;
  PF_HIP(hipMemcpyAsync(&total, ctx->nptr + nn, sizeof(total), hipMemcpyDeviceToHost, ctx->stream));
  PF_HIP(hipStreamSynchronize(ctx->stream));
  ctx->nblocks = total;
  PF_HIP(hipMalloc(&ctx->nidx, sizeof(int32_t) * (size_t)total));
  constexpr int PS = P <= 4 ? 4 : 8;
  PF_HIP(hipMalloc(&ctx->n2e_pos, (size_t)ctx->nelems * P * PS));
  PF_HIP(hipMemsetAsync(ctx->n2e_pos, 0, (size_t)ctx->nelems * P * PS, ctx->stream));
  PF_HIP(hipMalloc(&ctx->diagpos, (size_t)nn));
  PF_HIP(hipMemsetAsync(ctx->diagpos, 0, (size_t)nn, ctx->stream));  // nodes no element uses have empty rows: k_row_fill never writes theirs
  k_row_fill<P><<<(unsigned)cdiv(nn, T), T, 0, ctx->stream>>>(nn, ctx->conn, ctx->n2e_ptr, ctx->n2e, ctx->nptr,
                                                              ctx->nidx, ctx->n2e_pos, ctx->diagpos);
  PF_LAUNCH_CHECK();
  return PF_OK;
}

static int build_rows_tab(pf_ctx* ctx, int32_t* rowcnt, int32_t* flag) {
  const int64_t nn = ctx->nnodes;
  const int T = 64;
  k_row_count_tab<<<(unsigned)cdiv(nn, T), T, 0, ctx->stream>>>(ctx->P, nn, ctx->conn, ctx->n2e_ptr, ctx->n2e, rowcnt, flag);
  PF_LAUNCH_CHECK();
  int32_t f = 0;
  PF_HIP(hipMemcpyAsync(&f, flag, sizeof(f), hipMemcpyDeviceToHost, ctx->stream));
  PF_HIP(hipStreamSynchronize(ctx->stream));
  if (f) return pf_fail(ctx, PF_E_UNSUPPORTED, "pf_pattern_build: a node has more than 160 neighbours");
  int rc = exclusive_scan_i32(ctx, rowcnt, ctx->nptr, nn + 1);
  if (rc) return rc;
  int32_t total = 0;
  PF_HIP(hipMemcpyAsync(&total, ctx->nptr + nn, sizeof(total), hipMemcpyDeviceToHost, ctx->stream));
  PF_HIP(hipStreamSynchronize(ctx->stream));
  ctx->nblocks = total;
  PF_HIP(hipMalloc(&ctx->nidx, sizeof(int32_t) * (size_t)total));
  PF_HIP(hipMalloc(&ctx->diagpos, (size_t)nn));
  PF_HIP(hipMemsetAsync(ctx->diagpos, 0, (size_t)nn, ctx->stream));
  k_row_fill_tab<<<(unsigned)cdiv(nn, T), T, 0, ctx->stream>>>(ctx->P, nn, ctx->conn, ctx->n2e_ptr, ctx->n2e, ctx->nptr,
                                                               ctx->nidx, ctx->diagpos);
  PF_LAUNCH_CHECK();
  return PF_OK;
}

extern "C" int pf_pattern_build(pf_ctx* ctx, int d) {
  if (!ctx) return pf_fail(nullptr, PF_E_INVALID, "ctx is NULL");
  PF_REQUIRE(ctx->conn, "pf_pattern_build: call pf_mesh_set first");
  PF_REQUIRE(ctx->ndims == 2 ? (d == 1 || d == 2) : (d == 1 || d == 3),
             "pf_pattern_build: ndof_per_node must be 1 or the space dimension");
  PF_HIP(hipSetDevice(ctx->device));
  pf_pattern_free(ctx);
  const int64_t nn = ctx->nnodes, total = ctx->nelems * ctx->P;
  const int T = 256;
  int32_t *cnt = nullptr, *cursor = nullptr, *flag = nullptr;
  PF_HIP(hipMalloc(&cnt, sizeof(int32_t) * (nn + 1)));
  PF_HIP(hipMalloc(&cursor, sizeof(int32_t) * (nn + 1)));
  PF_HIP(hipMalloc(&flag, sizeof(int32_t)));
  PF_HIP(hipMemsetAsync(cnt, 0, sizeof(int32_t) * (nn + 1), ctx->stream));
  PF_HIP(hipMemsetAsync(cursor, 0, sizeof(int32_t) * (nn + 1), ctx->stream));
  PF_HIP(hipMemsetAsync(flag, 0, sizeof(int32_t), ctx->stream));
  PF_HIP(hipMalloc(&ctx->n2e_ptr, sizeof(int32_t) * (nn + 1)));
  PF_HIP(hipMalloc(&ctx->n2e, sizeof(int32_t) * total));
  PF_HIP(hipMalloc(&ctx->nptr, sizeof(int32_t) * (nn + 1)));

  k_count_incidence<<<(unsigned)cdiv(total, T), T, 0, ctx->stream>>>(ctx->conn, total, cnt);
  PF_LAUNCH_CHECK();
  int rc = max_i32(ctx, cnt, nn, &ctx->max_valence);
  if (rc) return rc;
  rc = exclusive_scan_i32(ctx, cnt, ctx->n2e_ptr, nn + 1);
  if (rc) return rc;
  k_fill_incidence<<<(unsigned)cdiv(total, T), T, 0, ctx->stream>>>(ctx->conn, total, ctx->n2e_ptr, cursor, ctx->n2e);
  PF_LAUNCH_CHECK();
  k_sort_incidence<<<(unsigned)cdiv(nn, T), T, 0, ctx->stream>>>(nn, ctx->n2e_ptr, ctx->n2e);
  PF_LAUNCH_CHECK();
  // row counts go into cnt (reused), scanned into nptr
  PF_HIP(hipMemsetAsync(cnt, 0, sizeof(int32_t) * (nn + 1), ctx->stream));
  const bool tabular = ctx->ndims == 3 && ctx->P != 8;  // table-driven elements: rows only, no tiles / visit records
  rc = tabular ? build_rows_tab(ctx, cnt, flag)
       : (ctx->P == 8) ? build_rows<8>(ctx, cnt, flag) : (ctx->P == 4) ? build_rows<4>(ctx, cnt, flag) : build_rows<3>(ctx, cnt, flag);
  if (rc == PF_OK) rc = max_i32(ctx, cnt, nn, &ctx->max_row_blocks);
  (void)hipFree(cnt);
  (void)hipFree(cursor);
  (void)hipFree(flag);
  if (rc) {
    pf_pattern_free(ctx);
    return rc;
  }
  PF_REQUIRE((int64_t)d * d * ctx->nblocks < ((int64_t)1 << 31), "pf_pattern_build: nnz exceeds int32 (scipy CSR index type)");
  // Max nodes per assembly tile.  A tile's incident pairs must fit one wave (64 lanes): the greedy pass below stops
  // a tile at 64 pairs, so the cap only has to cover the typical valence (npt * max_valence <= 128: triangles with
  // valence 6-8 get up to 10 nodes = 60 lanes instead of 8 nodes = 48) and keep the LDS image of the tile's rows
  // (npt * max_row_blocks blocks of d*d doubles, two planes for the scalar K+R pair) within ~20 KB per wave.
  ctx->npt = 0;
  if (ctx->max_valence <= 64 && !tabular) {
    const int64_t row_bytes = (int64_t)ctx->max_row_blocks * d * d * 8 * (d == 1 ? 2 : 1);
    for (int c = 16; c >= 1; c >>= 1)
      if ((c * ctx->max_valence <= 128 && c * row_bytes <= 20 * 1024) || c == 1) { ctx->npt = c; break; }
  }
  if (ctx->npt) {
    // Tiles: runs of <= npt consecutive nodes.  A tile's rows are one contiguous segment of the value array,
    // copied out by one wave; neighbouring segments are written by different CUs, and a cache line shared by
    // two of them reaches HBM as two partial writes (tools/micro/wbench.hip: 4.0-4.4 TB/s instead of 5.5).
    // So a tile may stop up to 3 nodes early when that puts its end on a better-aligned block boundary
    // (128 bytes = 128/(8 d d) blocks) without emptying more than a tenth of the wave's lanes; interior rows of
    // quad meshes then stay aligned until the next irregular row.
    // Greedy and sequential, hence on the host: two int32 arrays down, 16 bytes per tile + 1 byte per node up.
    std::vector<int32_t> hp(nn + 1), hv(nn + 1);
    PF_HIP(hipMemcpyAsync(hp.data(), ctx->nptr, sizeof(int32_t) * (nn + 1), hipMemcpyDeviceToHost, ctx->stream));
    PF_HIP(hipMemcpyAsync(hv.data(), ctx->n2e_ptr, sizeof(int32_t) * (nn + 1), hipMemcpyDeviceToHost, ctx->stream));
    PF_HIP(hipStreamSynchronize(ctx->stream));
    int lgA = 0;  // a segment start is 128-byte aligned iff its block index is a multiple of 128 / gcd(128, 8 d d)
    for (int bytes = 8 * d * d; (bytes << lgA) % 128; ++lgA) {}
    std::vector<int32_t> desc;
    desc.reserve(4 * (size_t)(nn / ctx->npt + 16));
    std::vector<uint8_t> slot(nn);
    ctx->max_tile_blocks = 0;
    const bool fixed = pf_exp_env("PF_TILE_FIXED") != nullptr;  // ablation: npt nodes per tile, no alignment search
    for (int64_t n0 = 0; n0 < nn;) {
      int64_t emax = std::min<int64_t>(n0 + ctx->npt, nn);
      while (emax > n0 + 1 && hv[emax] - hv[n0] > 64) --emax;  // at most 64 incident pairs per tile
      // 2x2 blocks: at most 160 blocks (5 KB) per tile, so that one irregular tile (16 nodes with rows of 11 blocks)
      // does not put the whole launch on the 10-store copy-out with a twice as large LDS image
      while (d == 2 && emax > n0 + 1 && hp[emax] - hp[n0] > 160) --emax;
      int64_t e = emax;
      if (emax < nn && !fixed) {
        // stop early for alignment only while the wave stays >= 90 % as full as it could be: on valence-6 triangles
        // the aligned choice (8 nodes, 48 of 64 lanes) measured 11 % SLOWER than 10 nodes ending mid-line (r02,
        // tools/tri_tile_fixed.py); on quads the full tile is the aligned one anyway
        const int32_t full = hv[emax] - hv[n0];
        int best = -1;
        for (int64_t c = emax; c > n0 && c > emax - 4; --c) {
          if (10 * (hv[c] - hv[n0]) < 9 * full) break;
          const int score = std::min(hp[c] ? __builtin_ctz((unsigned)hp[c]) : 31, lgA);
          if (score > best) { best = score; e = c; }
        }
      }
      ctx->max_tile_blocks = std::max(ctx->max_tile_blocks, hp[e] - hp[n0]);
      for (int64_t n = n0; n < e; ++n) slot[n] = (uint8_t)(n - n0);
      int32_t deep = 0;  // bit 7 (2-D tiles): a node with more than 8 incident elements
      if (ctx->P != 8)
        for (int64_t n = n0; n < e; ++n) deep |= (hv[n + 1] - hv[n] > 8) ? 128 : 0;
      desc.push_back((int32_t)n0);
      desc.push_back(hv[n0]);
      desc.push_back(hp[n0]);
      desc.push_back((int32_t)((e - n0) | deep | ((int64_t)(hv[e] - hv[n0]) << 8) | ((int64_t)(hp[e] - hp[n0]) << 16)));
      n0 = e;
    }
    ctx->ntiles = (int64_t)desc.size() / 4;
    PF_REQUIRE(ctx->max_tile_blocks < 65536 && ctx->npt <= 32, "pf_pattern_build: tile too large for the descriptor");
    uint8_t* dslot = nullptr;
    PF_HIP(hipMalloc(&ctx->tile_desc, sizeof(int32_t) * desc.size()));
    PF_HIP(hipMalloc(&dslot, nn));
    PF_HIP(hipMemcpyAsync(ctx->tile_desc, desc.data(), sizeof(int32_t) * desc.size(), hipMemcpyHostToDevice, ctx->stream));
    PF_HIP(hipMemcpyAsync(dslot, slot.data(), nn, hipMemcpyHostToDevice, ctx->stream));
    PF_HIP(hipMalloc(&ctx->vis_rec, sizeof(int4) * (size_t)total * (ctx->P == 8 ? 3 : 1)));
    if (ctx->P == 8)
      k_build_visits8<<<(unsigned)cdiv(nn, T), T, 0, ctx->stream>>>(nn, dslot, ctx->conn, ctx->n2e_ptr, ctx->n2e, ctx->n2e_pos, (int4*)ctx->vis_rec);
    else if (ctx->P == 4)
      k_build_visits<4><<<(unsigned)cdiv(nn, T), T, 0, ctx->stream>>>(nn, dslot, ctx->conn, ctx->n2e_ptr, ctx->n2e, ctx->n2e_pos, (int4*)ctx->vis_rec);
    else
      k_build_visits<3><<<(unsigned)cdiv(nn, T), T, 0, ctx->stream>>>(nn, dslot, ctx->conn, ctx->n2e_ptr, ctx->n2e, ctx->n2e_pos, (int4*)ctx->vis_rec);
    PF_LAUNCH_CHECK();
    if (ctx->P == 4 || ctx->P == 3) {  // compact records when the numbering allows them (triangles: else 12 bytes)
      int32_t* f8 = nullptr;
      PF_HIP(hipMalloc(&ctx->vis_rec8, sizeof(uint2) * (size_t)total));
      if (ctx->P == 3) PF_HIP(hipMalloc(&ctx->vis_rec12, 12 * (size_t)total));
      PF_HIP(hipMalloc(&f8, sizeof(int32_t)));
      PF_HIP(hipMemsetAsync(f8, 0, sizeof(int32_t), ctx->stream));
      if (ctx->P == 4)
        k_build_rec8<4><<<(unsigned)cdiv(nn, T), T, 0, ctx->stream>>>(nn, ctx->n2e_ptr, (const int4*)ctx->vis_rec, (uint2*)ctx->vis_rec8, f8);
      else
        k_build_rec_tri<<<(unsigned)cdiv(nn, T), T, 0, ctx->stream>>>(nn, ctx->n2e_ptr, (const int4*)ctx->vis_rec, (uint2*)ctx->vis_rec8,
                                                                      (uint32_t*)ctx->vis_rec12, f8);
      int32_t bad = 0;
      PF_HIP(hipMemcpyAsync(&bad, f8, sizeof(bad), hipMemcpyDeviceToHost, ctx->stream));
      PF_HIP(hipStreamSynchronize(ctx->stream));
      (void)hipFree(f8);
      if (bad || pf_exp_env("PF_NO_REC8")) pf_free(ctx->vis_rec8);
      if (ctx->vis_rec8 || pf_exp_env("PF_NO_REC12")) pf_free(ctx->vis_rec12);  // one compact form is kept
#ifdef PF_AB_TRI_REC  // tools/build_variant.sh: force the triangles' record width (12 or 16)
      if (ctx->P == 3 && PF_AB_TRI_REC >= 12 && ctx->vis_rec8) {
        pf_free(ctx->vis_rec8);
        if (PF_AB_TRI_REC == 12) {
          int32_t* f8 = nullptr;
          uint2* tmp8 = nullptr;
          PF_HIP(hipMalloc(&f8, sizeof(int32_t)));
          PF_HIP(hipMalloc(&tmp8, sizeof(uint2) * (size_t)total));
          PF_HIP(hipMalloc(&ctx->vis_rec12, 12 * (size_t)total));
          k_build_rec_tri<<<(unsigned)cdiv(nn, T), T, 0, ctx->stream>>>(nn, ctx->n2e_ptr, (const int4*)ctx->vis_rec, tmp8, (uint32_t*)ctx->vis_rec12, f8);
          PF_HIP(hipStreamSynchronize(ctx->stream));
          (void)hipFree(f8);
          (void)hipFree(tmp8);
        }
      }
      if (ctx->P == 3 && PF_AB_TRI_REC == 16) pf_free(ctx->vis_rec12);
#endif
    }
    PF_HIP(hipStreamSynchronize(ctx->stream));  // the host staging vectors go out of scope
    (void)hipFree(dslot);
    // Scattered numberings (no 8-byte records: some neighbour lies far away in node ids -- refinement order, a mesher's
    // insertion order): the nodes of ONE tile are still neighbours in space, but consecutive tiles are not, and the
    // coordinate gathers of the waves that run side by side share nothing.  The order in which tiles are WALKED is free
    // (each tile still writes its own contiguous segment): sort the descriptors along a Morton curve of their first
    // node, so that the tiles an XCD's waves take together (k_assemble_v4's XCD-aware walk) are neighbours in space and
    // share its L2.  r02, torque arm refined to 2.55 M triangles in refinement order: 0.0805 -> see
    // profiles/r02_assembly_ab.md section 15.
    ctx->tiles_sorted = false;
    const bool scattered = ctx->ndims == 2 && ctx->P != 8 && !ctx->vis_rec8 && !pf_exp_env("PF_NO_TILE_SORT") && PF_AB_TILE_SORT;
    if (scattered && ctx->ntiles > 1) {
      std::vector<double> xy((size_t)nn * 2);
      PF_HIP(hipMemcpy(xy.data(), ctx->X, sizeof(double) * xy.size(), hipMemcpyDeviceToHost));
      double lo[2] = {xy[0], xy[1]}, hi[2] = {xy[0], xy[1]};
      for (int64_t i = 0; i < nn; ++i)
        for (int c = 0; c < 2; ++c) { lo[c] = std::min(lo[c], xy[2 * i + c]); hi[c] = std::max(hi[c], xy[2 * i + c]); }
      const double span = std::max(std::max(hi[0] - lo[0], hi[1] - lo[1]), 1e-300);
      auto spread = [](uint64_t v) {
        v = (v | (v << 16)) & 0x0000FFFF0000FFFFull; v = (v | (v << 8)) & 0x00FF00FF00FF00FFull;
        v = (v | (v << 4)) & 0x0F0F0F0F0F0F0F0Full; v = (v | (v << 2)) & 0x3333333333333333ull;
        return (v | (v << 1)) & 0x5555555555555555ull;
      };
      const int64_t nt = ctx->ntiles;
      std::vector<std::pair<uint64_t, int32_t>> key((size_t)nt);
      for (int64_t t = 0; t < nt; ++t) {
        const int64_t n0 = desc[4 * t];
        const uint64_t qx = (uint64_t)((xy[2 * n0] - lo[0]) / span * 65535.0), qy = (uint64_t)((xy[2 * n0 + 1] - lo[1]) / span * 65535.0);
        key[t] = {spread(qx) | (spread(qy) << 1), (int32_t)t};
      }
      std::sort(key.begin(), key.end());
      std::vector<int32_t> sorted(desc.size());
      for (int64_t t = 0; t < nt; ++t) memcpy(&sorted[4 * t], &desc[4 * (size_t)key[t].second], 4 * sizeof(int32_t));
      PF_HIP(hipMemcpy(ctx->tile_desc, sorted.data(), sizeof(int32_t) * sorted.size(), hipMemcpyHostToDevice));
      ctx->tiles_sorted = true;
      // the same walk for the product's tiles (64 nodes per wave for d = 2, 128 for d = 1: pf_solve.hip k_spmv)
      if (d <= 2) {
        const int64_t ts = d == 1 ? PF_SPMV_TILE1 : PF_SPMV_TILE2, nts = cdiv(nn, ts);  // k_spmv's tiles: pf_solve.hip asserts that its shapes match
        std::vector<std::pair<uint64_t, int32_t>> k2((size_t)nts);
        for (int64_t t = 0; t < nts; ++t) {
          const int64_t n0 = t * ts;
          const uint64_t qx = (uint64_t)((xy[2 * n0] - lo[0]) / span * 65535.0), qy = (uint64_t)((xy[2 * n0 + 1] - lo[1]) / span * 65535.0);
          k2[t] = {spread(qx) | (spread(qy) << 1), (int32_t)t};
        }
        std::sort(k2.begin(), k2.end());
        std::vector<int32_t> perm((size_t)nts);
        for (int64_t t = 0; t < nts; ++t) perm[t] = k2[t].second;
        PF_HIP(hipMalloc(&ctx->spmv_perm, sizeof(int32_t) * (size_t)nts));
        PF_HIP(hipMemcpy(ctx->spmv_perm, perm.data(), sizeof(int32_t) * (size_t)nts, hipMemcpyHostToDevice));
      }
    }
  }
  ctx->d = d;
  return PF_OK;
}

int pf_ensure_e2pos(pf_ctx* ctx) {
  if (ctx->e2pos) return PF_OK;
  const int P = ctx->P;
  PF_REQUIRE(P != 8, "the atomic scatter cross-check is a 2-D mode");
  const int64_t n = ctx->nelems * P * P;
  PF_HIP(hipMalloc(&ctx->e2pos, (size_t)n));
  if (P == 4)
    k_e2pos<4><<<(unsigned)cdiv(n, 256), 256, 0, ctx->stream>>>(ctx->nelems, ctx->conn, ctx->nptr, ctx->nidx, ctx->e2pos);
  else
    k_e2pos<3><<<(unsigned)cdiv(n, 256), 256, 0, ctx->stream>>>(ctx->nelems, ctx->conn, ctx->nptr, ctx->nidx, ctx->e2pos);
  PF_LAUNCH_CHECK();
  return PF_OK;
}

extern "C" int pf_pattern_dims(pf_ctx* ctx, int64_t* nrows, int64_t* nnz, int64_t* nblocks, int32_t* max_row_blocks) {
  if (!ctx) return pf_fail(nullptr, PF_E_INVALID, "ctx is NULL");
  PF_REQUIRE(ctx->d, "pf_pattern_dims: no pattern built");
  if (nrows) *nrows = ctx->d * ctx->nnodes;
  if (nnz) *nnz = (int64_t)ctx->d * ctx->d * ctx->nblocks;
  if (nblocks) *nblocks = ctx->nblocks;
  if (max_row_blocks) *max_row_blocks = ctx->max_row_blocks;
  return PF_OK;
}

extern "C" int pf_pattern_get_graph(pf_ctx* ctx, int32_t* nptr, int32_t* nidx) {
  if (!ctx) return pf_fail(nullptr, PF_E_INVALID, "ctx is NULL");
  PF_REQUIRE(ctx->d, "pf_pattern_get_graph: no pattern built");
  if (nptr) PF_HIP(hipMemcpyAsync(nptr, ctx->nptr, sizeof(int32_t) * (ctx->nnodes + 1), hipMemcpyDeviceToDevice, ctx->stream));
  if (nidx) PF_HIP(hipMemcpyAsync(nidx, ctx->nidx, sizeof(int32_t) * ctx->nblocks, hipMemcpyDeviceToDevice, ctx->stream));
  return PF_OK;
}

extern "C" int pf_pattern_get_tiles(pf_ctx* ctx, int64_t* ntiles, int32_t* npt, int64_t* nvisits, int32_t* tile_desc,
                                    int32_t* vis_rec, int32_t* n2e_ptr) {
  if (!ctx) return pf_fail(nullptr, PF_E_INVALID, "ctx is NULL");
  PF_REQUIRE(ctx->d && ctx->npt && ctx->tile_desc && ctx->vis_rec, "pf_pattern_get_tiles: this pattern has no assembly tiles");
  const int64_t nv = ctx->nelems * ctx->P;
  if (ntiles) *ntiles = ctx->ntiles;
  if (npt) *npt = ctx->npt;
  if (nvisits) *nvisits = nv;
  if (tile_desc) PF_HIP(hipMemcpyAsync(tile_desc, ctx->tile_desc, sizeof(int32_t) * 4 * ctx->ntiles, hipMemcpyDeviceToDevice, ctx->stream));
  if (vis_rec) PF_HIP(hipMemcpyAsync(vis_rec, ctx->vis_rec, sizeof(int32_t) * 4 * nv * (ctx->P == 8 ? 3 : 1), hipMemcpyDeviceToDevice, ctx->stream));
  if (n2e_ptr) PF_HIP(hipMemcpyAsync(n2e_ptr, ctx->n2e_ptr, sizeof(int32_t) * (ctx->nnodes + 1), hipMemcpyDeviceToDevice, ctx->stream));
  return PF_OK;
}

extern "C" int pf_pattern_tile_info(pf_ctx* ctx, int32_t* max_tile_blocks) {
  if (!ctx) return pf_fail(nullptr, PF_E_INVALID, "ctx is NULL");
  PF_REQUIRE(ctx->d && ctx->npt && ctx->tile_desc, "pf_pattern_tile_info: this pattern has no assembly tiles");
  if (max_tile_blocks) *max_tile_blocks = ctx->max_tile_blocks;
  return PF_OK;
}

extern "C" int pf_pattern_get_compact(pf_ctx* ctx, int32_t* rec_bytes, int64_t* nvisits, uint32_t* recs) {
  if (!ctx) return pf_fail(nullptr, PF_E_INVALID, "ctx is NULL");
  PF_REQUIRE(ctx->d, "pf_pattern_get_compact: no pattern built");
  const void* src = ctx->vis_rec8 ? ctx->vis_rec8 : ctx->vis_rec12;
  const int32_t bytes = ctx->vis_rec8 ? 8 : ctx->vis_rec12 ? 12 : 0;
  if (rec_bytes) *rec_bytes = bytes;
  if (nvisits) *nvisits = src ? ctx->nelems * ctx->P : 0;
  if (recs && src)
    PF_HIP(hipMemcpyAsync(recs, src, (size_t)bytes * (size_t)(ctx->nelems * ctx->P), hipMemcpyDeviceToDevice, ctx->stream));
  return PF_OK;
}

extern "C" int pf_pattern_get_csr(pf_ctx* ctx, int32_t* indptr, int32_t* indices) {
  if (!ctx) return pf_fail(nullptr, PF_E_INVALID, "ctx is NULL");
  PF_REQUIRE(ctx->d, "pf_pattern_get_csr: no pattern built");
  const int T = 128;
  unsigned g = (unsigned)cdiv(ctx->nnodes, T);
  if (ctx->d == 1)
    k_csr_structure<1><<<g, T, 0, ctx->stream>>>(ctx->nnodes, ctx->nptr, ctx->nidx, indptr, indices);
  else if (ctx->d == 3)
    k_csr_structure<3><<<g, T, 0, ctx->stream>>>(ctx->nnodes, ctx->nptr, ctx->nidx, indptr, indices);
  else
    k_csr_structure<2><<<g, T, 0, ctx->stream>>>(ctx->nnodes, ctx->nptr, ctx->nidx, indptr, indices);
  PF_LAUNCH_CHECK();
  return PF_OK;
}

extern "C" int pf_vals_to_csr(pf_ctx* ctx, const double* vals, double* csr) {
  if (!ctx) return pf_fail(nullptr, PF_E_INVALID, "ctx is NULL");
  PF_REQUIRE(ctx->d && vals && csr, "pf_vals_to_csr: no pattern / NULL argument");
  if (ctx->d == 1) {
    PF_HIP(hipMemcpyAsync(csr, vals, sizeof(double) * ctx->nblocks, hipMemcpyDeviceToDevice, ctx->stream));
    return PF_OK;
  }
  const int T = 128;
  if (ctx->d == 3)
    k_vals_to_csr<3><<<(unsigned)cdiv(ctx->nnodes, T), T, 0, ctx->stream>>>(ctx->nnodes, ctx->nptr, vals, csr);
  else
    k_vals_to_csr<2><<<(unsigned)cdiv(ctx->nnodes, T), T, 0, ctx->stream>>>(ctx->nnodes, ctx->nptr, vals, csr);
  PF_LAUNCH_CHECK();
  return PF_OK;
}
