// Internal header of libpyfem_hip.so (gfx950 only).  Public ABI: include/pyfem_hip.h
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <string>
#include <vector>

#include "pyfem_hip.h"

// reduction scratch: PF_NPART_ARRAYS arrays of PF_NPART per-block partial sums (pf_solve.hip)
#define PF_NPART 4096
#define PF_NPART_ARRAYS 8

struct pf_amg;

// Experiment / ablation knobs (PF_ABLATE, PF_GATHER_IMPL, PF_TILE_*, PF_AMG_* tuning, PF_SPMV_*) exist only in
// builds with -DPF_EXPERIMENTS (`python -m pyfem_gpu_testflight_amd.build --experiments` makes a separate
// libpyfem_hip_exp.so for tools/): an inherited environment variable can never change what the product library
// computes.  The two operational knobs (PF_HOST_THREADS, PF_AMG_VERBOSE) are read once, in pf_ctx_create.
#ifdef PF_EXPERIMENTS
static inline const char* pf_exp_env(const char* name) { return getenv(name); }
#else
static inline const char* pf_exp_env(const char*) { return nullptr; }
#endif

struct pf_ctx {
  int device = 0;
  int ncu = 0;               // compute units of the device (pf_ctx_create)
  hipStream_t stream = nullptr;
  std::string err;
  std::string warn;          // last non-fatal condition (pf_last_warning), e.g. a preconditioner that was refused
  int host_threads = 0;      // PF_HOST_THREADS at pf_ctx_create (0: hardware concurrency)
  bool amg_verbose = false;  // PF_AMG_VERBOSE at pf_ctx_create

  // ---- mesh (device copies) ----
  int64_t nnodes = 0, nelems = 0;
  int P = 0;                 // nodes per element: 4 quad, 3 tri, 8 block (3-D)
  int ndims = 2;             // 2, or 3 for the 8-node block element
  double* X = nullptr;       // (nnodes,ndims)
  int32_t* conn = nullptr;   // (nelems,P)

  // ---- symbolic pattern ----
  int d = 0;                 // dofs per node (block size)
  int64_t nblocks = 0;       // node-graph nnz
  int32_t max_row_blocks = 0;  // max neighbours of a node (incl. itself)
  int32_t max_valence = 0;     // max elements around a node
  int32_t* nptr = nullptr;   // (nnodes+1) node-graph row pointer
  int32_t* nidx = nullptr;   // (nblocks)  neighbour node ids, ascending per row
  int16_t* nidx16 = nullptr; // (nblocks)  d = 1, 2: neighbour id minus the first node of the row's SpMV tile (128 / 64
  int nidx16_state = 0;      //            rows); state 0 not tried, 1 usable, -1 some offset does not fit 16 bits
  int32_t* n2e_ptr = nullptr;  // (nnodes+1) node -> incident (element, local index) list
  int32_t* n2e = nullptr;      // (nelems*P) entries e*P + a, ascending per node
  uint8_t* n2e_pos = nullptr;  // (nelems*P*PS) PS = 4 (8 for blocks) bytes per incident pair j; byte b (< P):
                               //   position of conn[e,b] in row n of the node graph
  uint8_t* diagpos = nullptr;  // (nnodes) position of n in its own row
  void* vis_rec = nullptr;     // (nelems*P) int4 per incident pair: other node ids + packed positions/indices
                               //   (blocks: 3 x int4 = 8 node ids, 8 position bytes, local index, tile slot)
  void* vis_rec12 = nullptr;   // triangles: (nelems*3) x 12 bytes {o1, o2, w} (the 16-byte record without its unused slot)
                               //   when the 8-byte form does not fit the numbering
  void* vis_rec8 = nullptr;    // (nelems*P) uint2: the same records in 8 bytes (13-bit node offsets, 5-bit positions) when
                               //   the numbering allows it, else nullptr (pf_pattern.hip k_build_rec8)
  int npt = 0;                 // max nodes per assembly tile (incident pairs of a tile fit 64 lanes); 0: legacy kernel only
  void* tile_desc = nullptr;   // (ntiles) int4 {first node, first visit, first block, nodes | deep<<7 | visits<<8 | blocks<<16}
  int64_t ntiles = 0;          //   tiles end where the value segment is 128-byte aligned whenever possible
  int32_t max_tile_blocks = 0; // blocks of the largest tile (sets the static store count of the copy-out)
  bool tiles_sorted = false;   // descriptors sorted along a Morton curve (scattered numberings), else in node order
  int32_t* spmv_perm = nullptr; // scattered numberings: k_spmv's 64 (d = 2) / 128 (d = 1)-node tiles in Morton order, else nullptr
  uint8_t* e2pos = nullptr;    // (nelems*P*P) pos of conn[e,b] in row conn[e,a] (atomic mode; lazy)

  // table-driven elements (pf_tabular.hip): elements sorted by colour, no two of one colour share a node
  int32_t* tab_order = nullptr;          // (nelems) element ids, colour by colour
  std::vector<int64_t> tab_color_ptr;    // (ncolours+1) offsets into tab_order (host)

  pf_amg* amg = nullptr;       // AMG hierarchy (graph part built once per pattern, pf_amg.hip)

  // ---- scratch owned by the ctx ----
  double* scal = nullptr;      // PF_NPART_ARRAYS x PF_NPART per-block partial sums
  double* cgw = nullptr;       // CG work vectors (4*nrows), grown on demand
  int64_t cgw_cap = 0;
  double* cg_ap = nullptr;     // the product's output vector of pf_cg, placed by measurement (pf_solve.hip cg_place_ap)
  int64_t cg_ap_n = 0;
  void* scan_tmp = nullptr;
  size_t scan_tmp_cap = 0;
};

extern thread_local std::string g_pf_err;

static inline int pf_fail(pf_ctx* ctx, int code, const std::string& msg) {
  if (ctx) ctx->err = msg;
  g_pf_err = msg;
  return code;
}

#define PF_HIP(call)                                                                       \
  do {                                                                                     \
    hipError_t e_ = (call);                                                                \
    if (e_ != hipSuccess)                                                                  \
      return pf_fail(ctx, PF_E_HIP,                                                        \
                     std::string(#call) + " failed: " + hipGetErrorString(e_) + " (" +     \
                         __FILE__ + ":" + std::to_string(__LINE__) + ")");                 \
  } while (0)

#define PF_REQUIRE(cond, msg)                                       \
  do {                                                              \
    if (!(cond)) return pf_fail(ctx, PF_E_INVALID, std::string(msg)); \
  } while (0)

#define PF_LAUNCH_CHECK() PF_HIP(hipGetLastError())

static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

template <typename T>
static inline void pf_free(T*& p) {
  if (p) (void)hipFree(p);
  p = nullptr;
}

// peer-store halo exchange (pf_halo.hip)
struct pf_halo;
int pf_halo_exchange_impl(pf_halo* h, int plan, double* v, const double* scal_in, int ns, double* scal_out);
int32_t* pf_halo_err_ptr(pf_halo* h);
pf_ctx* pf_halo_ctx(pf_halo* h);

// internal entry points shared between translation units
int pf_pattern_free(pf_ctx* ctx);
int pf_ensure_e2pos(pf_ctx* ctx);
int pf_launch_assemble_elasticity(pf_ctx* ctx, double E, double nu, double p, const double* rho, int mode,
                                  double* Kvals);
int pf_launch_assemble_helmholtz(pf_ctx* ctx, double r0, double* Kvals, double* Rvals);
int pf_launch_spmv(pf_ctx* ctx, const double* vals, const double* x, double* y, double* dot_part,
                   const double* sm_dinv = nullptr, const double* sm_b = nullptr,  // smoother epilogues: pf_solve.hip
                   const float* vals32 = nullptr);                                 // fp32 copy of vals to use instead
void pf_amg_free(pf_ctx* ctx);
int pf_amg_setup(pf_ctx* ctx, const double* vals, int reuse);   // PF_E_UNSUPPORTED: matrix not positive definite
int pf_amg_apply(pf_ctx* ctx, const double* r, double* z);
int pf_amg_describe(pf_ctx* ctx, int32_t* nlevels, int64_t* nodes, int32_t* dofs_per_node, double* omega, int cap);
