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
  bool multi_stream = false;   // set for the duration of an AMG-preconditioned solve: several operators (fp64 K, its fp32 copy, the
                               // transfer operators) are streamed per iteration, nothing stays in the memory-side cache, and the
                               // products read their values non-temporally whatever their size (pf_launch_spmv)
  double* pin = nullptr;       // pinned host buffer (2 * PF_NPART doubles) and event for the PCG loops' residual read-back: the
  hipEvent_t pin_ev = nullptr; //   copy is queued right after the update that produces |r|^2 and awaited after the rest of the
                               //   iteration has been queued, so the device never waits for the host (pf_solve.hip)
  int dcg_par = 0;             // which of the two r.z / r.r partial-array sets the next k_dcg_update reads (pf_dist_update)
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

// ---- peer-store halo exchange (pf_halo.hip; the fused product / update kernels of pf_solve.hip read these) ----------
#define PF_HALO_MAXPLANS 8
#define PF_HALO_MAXWORLD 64   // ranks of one node; the residency argument of k_halo_xchg is made for this bound

// What the FUSED exchange of a plan needs on top of the packed-message tables (pf_halo_plan_fuse): the product kernel
// stores the rows of interface nodes straight into the neighbours' windows from its epilogue, so it has to find, for a
// node, the peers that share it and the node's position in the message to each of them; and it walks the tiles that
// hold interface nodes FIRST, so that the rows are on their way while the interior tiles are computed.
struct pf_halo_fuse {
  int d = 0;                     // entries per node of the vectors this plan moves (= the context's dofs per node)
  int tile_nodes = 0;            // nodes per k_spmv tile the tables were built for (64 at d = 2, 128 at d = 1)
  int64_t n_ifnodes = 0;
  int32_t n_iftiles = 0;         // tiles with at least one interface node: the first n_iftiles entries of tperm
  int32_t* ifslot = nullptr;     // (nnodes) index of the node among the interface nodes, or -1         [device]
  int32_t* sh_ptr = nullptr;     // (n_ifnodes+1) CSR over the sharing peers of each interface node     [device]
  int32_t* sh_peer = nullptr;    // peer rank                                                           [device]
  int32_t* sh_pos = nullptr;     // offset (doubles) of the node's d values inside the message to it    [device]
  int32_t* tperm = nullptr;      // (ntiles) tile walk: interface tiles first, the rest in the context's order [device]
  uint8_t* flags8 = nullptr;     // (d*nnodes) bit 0: this rank owns the dof, bit 1: the dof lies on an interface [device]
};

struct pf_halo_plan {
  int nnbr = 0;
  int64_t max_msg = 0;           // largest message of the plan (doubles)
  int32_t* nbr = nullptr;        // (nnbr) neighbour ranks, ascending                        [device]
  int32_t* nbr_ptr = nullptr;    // (nnbr+1) offsets into send_idx                           [device]
  int32_t* send_idx = nullptr;   // local vector index of every packed value                 [device]
  int64_t n_if = 0;              // interface entries of this rank
  int32_t* if_idx = nullptr;     // (n_if) local vector index                                [device]
  int32_t* src_ptr = nullptr;    // (n_if+1) CSR over the contributing ranks of each entry   [device]
  int32_t* src_off = nullptr;    // rank * cap + position in that rank's message; -1 = this rank's own value [device]
  std::vector<int32_t> h_nbr, h_nbr_ptr, h_send_idx, h_if_idx;  // host copies (pf_halo_plan_fuse derives its tables from them)
  pf_halo_fuse fuse;             // empty (d == 0) until pf_halo_plan_fuse
};

struct pf_halo {
  pf_ctx* ctx = nullptr;
  int world = 0, rank = 0;
  int64_t cap = 0;               // doubles per (sender, parity) message slot
  size_t win_bytes = 0;
  double* win = nullptr;         // this rank's window (fine-grained device memory)
  double** peer_win = nullptr;   // (world) device array of mapped window base pointers (own entry = win)
  std::vector<void*> mapped;     // host copies for hipIpcCloseMemHandle
  int32_t* err = nullptr;        // device flag: 1 after a wait timed out (every later wait gives up at once)
  uint32_t* done = nullptr;      // (world) workgroups that finished a peer's message in the current send
  unsigned long long* sent = nullptr;  // send workgroups that have finished READING the vector, over all exchanges so far
  unsigned long long sent_target = 0;  // its value once every send workgroup of the exchanges issued so far is through
  uint64_t seq = 0;
  // fused exchanges (rows from the product's epilogue, scalars from the update kernel): their own message slots, flags
  // and sequence numbers, so that the two protocols never meet in a slot
  uint64_t rseq = 0;             // row messages issued so far (slot parity = rseq & 1)
  uint64_t sseq = 0;             // scalar messages issued so far
  unsigned long long* rows_done = nullptr;  // interface tiles finished in the running fused product (reset by the last) [device]
  int colocated = 1;             // ranks of the job that share this rank's device (pf_halo_set_colocated; tests only)
  std::vector<int32_t> fuse_nbr; // neighbour list every fused plan of this window must share (slot-reuse argument, pf_halo.hip)
  long long timeout_ticks = 1000000000LL;  // 10 s of the 100 MHz wall clock: ranks reach their first exchange skewed by
                                           // their own setup (pattern, hierarchy, code-object load)
  pf_halo_plan plans[PF_HALO_MAXPLANS];
  int nplans = 0;
};

// Window layout (doubles; identical on every rank, so a sender needs no table of its peers):
//   flag[world] | rflag[world] | sflag[world] (uint64 each, padded to 8) | scal[2][world][NSCAL] | scal2[2][world][NSCAL]
//   | recv[2][world][cap] | recv2[2][world][cap]
// flag / scal / recv: the packed exchange (k_halo_xchg); rflag / recv2: rows stored by the fused product; sflag / scal2:
// scalars stored by the fused update.
__host__ __device__ static inline size_t halo_flag_doubles(int world) { return (size_t)((world + 7) & ~7); }
__host__ __device__ static inline size_t halo_rflag_off(int world) { return halo_flag_doubles(world); }
__host__ __device__ static inline size_t halo_sflag_off(int world) { return 2 * halo_flag_doubles(world); }
__host__ __device__ static inline size_t halo_scal_off(int world) { return 3 * halo_flag_doubles(world); }
__host__ __device__ static inline size_t halo_scal2_off(int world) { return halo_scal_off(world) + (size_t)2 * world * PF_HALO_NSCAL; }
__host__ __device__ static inline size_t halo_recv_off(int world) { return halo_scal2_off(world) + (size_t)2 * world * PF_HALO_NSCAL; }
__host__ __device__ static inline size_t halo_recv2_off(int world, int64_t cap) { return halo_recv_off(world) + (size_t)2 * world * cap; }
__host__ __device__ static inline size_t halo_win_doubles(int world, int64_t cap) { return halo_recv2_off(world, cap) + (size_t)2 * world * cap; }

// by-value kernel argument of the SEND instantiations of k_spmv (pf_solve.hip): where the rows of interface nodes go
struct pf_spmv_send {
  const int32_t* ifslot;
  const int32_t* sh_ptr;
  const int32_t* sh_peer;
  const int32_t* sh_pos;
  double* const* peer_win;
  const int32_t* nbr;            // neighbour ranks of the plan (their row flags are raised when the last interface tile is through)
  const double* stop;            // NULL, or a device flag: non-zero = the solve has converged, the whole launch is a no-op
                                 // (every rank reads the same flag value: nobody waits for rows that are not sent)
  unsigned long long* rows_done; // interface tiles of THIS launch that are through; the last one resets it
  uint64_t rseq;
  int64_t slot_off;              // halo_recv2_off + (parity * world + rank) * cap: this rank's slot in a peer's window
  int64_t rflag_off;             // halo_rflag_off + rank (uint64 units from the window base)
  int32_t n_iftiles;
  int32_t nnbr;
};

// by-value kernel argument of the kernels that RECEIVE a fused exchange (k_dcg_update, k_halo_recv_apply)
struct pf_halo_recv {
  double* win;                   // this rank's window
  double* const* peer_win;
  const int32_t* nbr;
  const int32_t* if_idx;
  const int32_t* src_ptr;
  const int32_t* src_off;
  int32_t* err;
  long long timeout_ticks;
  int64_t n_if;
  int64_t cap;
  uint64_t rseq, sseq;           // the row / scalar message to wait for (sseq = 0: no scalars in this exchange)
  int32_t world, rank, nnbr;
};

// Accesses of the fused exchange to window memory (fine-grained, possibly a peer's over xGMI).  RELAXED system-scope
// atomics: the instructions bypass the caches (sc0 sc1) but carry no fence.  An ACQUIRE load or a seq_cst fence on
// gfx950 ends in a buffer_inv that drops the whole L2 of the XCD -- harmless in a kernel that only exchanges, ruinous in
// one that streams 100 MB of vectors at the same time (first version of k_dcg_update: 1024 blocks polling with acquire
// loads and a __threadfence_system each made the iteration 1.6x SLOWER than the four-launch loop it replaced).  What
// orders things instead: the sender completes its data stores (release fence: write-back, no invalidate) before it
// stores the flag; the receiver's data loads are issued after the flag load has returned (control dependency +
// s_waitcnt) and go to memory, not to a cache.
__device__ __forceinline__ uint64_t pf_ld_sys_u64(const uint64_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ double pf_ld_sys_f64(const double* p) {
  return __longlong_as_double(__hip_atomic_load(reinterpret_cast<const long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
}
__device__ __forceinline__ void pf_st_sys_f64(double* p, double v) {
  __hip_atomic_store(reinterpret_cast<long long*>(p), __double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void pf_release_sys() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, ""); }  // write-back + wait, no invalidate
// spin until *flag >= seq or the timeout; returns true on timeout
__device__ __forceinline__ bool pf_wait_flag(const uint64_t* flag, uint64_t seq, long long timeout_ticks) {
  const long long t0 = wall_clock64();
  while (pf_ld_sys_u64(flag) < seq) {
    if (wall_clock64() - t0 > timeout_ticks) return true;
    __builtin_amdgcn_s_sleep(1);
  }
  return false;
}

int pf_halo_exchange_impl(pf_halo* h, int plan, double* v, const double* scal_in, int ns, double* scal_out);
int32_t* pf_halo_err_ptr(pf_halo* h);
pf_ctx* pf_halo_ctx(pf_halo* h);
// fused exchange, host side (pf_halo.hip).  Nothing is numbered until the launch that sends was accepted:
//   pf_halo_fused_send   fills the send block of the NEXT row message of `plan` (rseq + 1)
//   pf_halo_fused_sent   the product launch was accepted: the row message exists now
//   pf_halo_fused_recv   fills the receive block for the row message last sent (and, with scalars, for scalar message
//                        sseq + 1, which the receiving kernel itself sends before it waits)
//   pf_halo_fused_scal   the receiving launch was accepted: the scalar message exists now
int pf_halo_fused_send(pf_halo* h, int plan, pf_spmv_send* out);
void pf_halo_fused_sent(pf_halo* h, int plan);
void pf_halo_fused_recv(pf_halo* h, int plan, bool scalars, pf_halo_recv* out);
void pf_halo_fused_scal(pf_halo* h);
const pf_halo_fuse* pf_halo_fuse_of(pf_halo* h, int plan);  // nullptr if the plan was not fused
int pf_halo_world(pf_halo* h);
int pf_halo_colocated(pf_halo* h);                          // ranks of the job on this rank's device (1 in production)

// k_spmv tiles (pf_solve.hip): nodes per wave tile at d = 1 / d = 2; pf_halo_plan_fuse builds its tile walk for these
static constexpr int PF_SPMV_TILE1 = 128, PF_SPMV_TILE2 = 64;

// internal entry points shared between translation units
int pf_pattern_free(pf_ctx* ctx);
int pf_ensure_e2pos(pf_ctx* ctx);
int pf_launch_assemble_elasticity(pf_ctx* ctx, double E, double nu, double p, const double* rho, int mode,
                                  double* Kvals);
int pf_launch_assemble_helmholtz(pf_ctx* ctx, double r0, double* Kvals, double* Rvals);
int pf_launch_spmv(pf_ctx* ctx, const double* vals, const double* x, double* y, double* dot_part,
                   const double* sm_dinv = nullptr, const double* sm_b = nullptr,  // smoother epilogues: pf_solve.hip
                   const float* vals32 = nullptr,                                  // fp32 copy of vals to use instead
                   const pf_spmv_send* snd = nullptr, const int32_t* snd_tperm = nullptr,   // fused exchange (pf_halo.hip)
                   int nv = 1,   // 2: x, y, sm_b hold TWO vectors interleaved per node (scalar matrices; dot_part: 2 per wave)
                   const double* pw_diag = nullptr);  // y = (A x) ./ pw_diag, dot_part <- partials of |y|^2 (power-iteration step)
// read-back without a device stall (pf_solve.hip): pieces into the context's pinned buffer, an event behind them, awaited later
int pf_pin_copy(pf_ctx* ctx, size_t double_offset, const void* dev, size_t bytes);
int pf_pin_mark(pf_ctx* ctx);
int pf_pin_wait(pf_ctx* ctx);
int pf_spmv_waves(pf_ctx* ctx);  // waves (= dot partials) of one pf_launch_spmv launch on this pattern
int pf_halo_recv_apply(pf_halo* h, int plan, const double* x, const double* d, const double* b, double* y);
// the two launches of a distributed CG iteration and their helpers (pf_solve.hip; used by pf_dist.hip as well)
int pf_dist_product(pf_ctx* ctx, pf_halo* halo, int plan, const double* vals, const float* vals32, const double* x, double* y,
                    bool dot, const double* sm_dinv, const double* sm_b, const double* stop);
int pf_dist_update(pf_ctx* ctx, pf_halo* halo, int plan, const double* w, const double* dinv, bool keep_z, bool gamma_from_dot,
                   bool rr_from_init, double* x, double* r, double* z, double* p, double* s, const double* state_in,
                   double* state_out, double* redg, bool first, double tol2, int it);
int pf_dist_dot_gamma(pf_ctx* ctx, const double* r, const double* z, const double* own_w);
int pf_dist_init_partials(pf_ctx* ctx, const double* state0, const double* p, const double* s, const double* dinv,
                          const double* own_w, double* x, double* r, double* z);
void pf_amg_free(pf_ctx* ctx);
int pf_amg_more_power_steps(pf_ctx* ctx);  // three times the power-iteration steps from the next numeric setup on (pf_cg's rescue)
int pf_amg_setup(pf_ctx* ctx, const double* vals, int reuse);   // PF_E_UNSUPPORTED: matrix not positive definite
int pf_amg_apply(pf_ctx* ctx, const double* r, double* z, double* rz_part = nullptr, bool* rz_done = nullptr, int nv = 1);  // rz_part: partials of
                                                                    // r.z from the last sweep (one per product wave) when *rz_done
int pf_amg_describe(pf_ctx* ctx, int32_t* nlevels, int64_t* nodes, int32_t* dofs_per_node, double* omega, int cap);
