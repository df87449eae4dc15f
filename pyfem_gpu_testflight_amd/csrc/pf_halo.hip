// Peer-store halo exchange for the one-process-per-GPU partition (gfx950, xGMI) -- no counterpart in the reference
// (single process).  SURVEY 8e / BASELINE north_star: "RCCL all-reduce over xGMI for the CG dot-products and halo
// residual sum".  What a CG iteration of the element-block partition has to move is tiny and latency-bound: the
// partial row sums of the interface dofs (one node row per cut: 4 098 doubles per neighbour for the 4M-quad strips)
// and three scalars.  A collective library call per iteration costs more host and launch time than the iteration's
// kernels at 8 GPUs, so this path uses what the hardware offers for exactly that case:
//
//   * every rank owns one WINDOW in fine-grained device memory, shared with the other ranks of the node through
//     hipIpcGetMemHandle / hipIpcOpenMemHandle (set up once, handles travel over torch.distributed);
//   * the SEND workgroups of k_halo_xchg pack the interface values and STORE them straight into the neighbours' windows
//     over xGMI (xGMI is point-to-point: one hop to every peer of the node), write the rank's scalars into every peer's
//     window, fence, and raise a per-sender sequence flag with a system-scope release store;
//   * the RECEIVE workgroups of the same launch spin on the flags of their senders (system-scope acquire loads, bounded
//     by a wall-clock timeout so that the grid always drains) and on a counter of this rank's own send workgroups (the
//     vector's interface entries are overwritten only after every sender has read them), then sum what arrived IN
//     ASCENDING RANK ORDER -- every rank that shares a dof forms the same sum of the same operands, so interface copies
//     stay bitwise identical on all ranks -- and reduce the scalars the same way (a deterministic all-reduce without a
//     collective).
//   ONE launch per exchange on the solver's stream (at most 8 * 16 + 256 workgroups: senders and receivers are resident
//   together), nothing on the host: the whole CG loop (pf_dist_cg) is a sequence of kernel launches from C.
//
// Synchronisation: exchanges are numbered (seq = 1, 2, ...; SPMD, every rank issues the same sequence).  Message
// buffers are double-buffered by seq parity.  A sender may run at most one exchange ahead of a receiver (it needs
// the receiver's message of exchange k to finish k itself), so parity k is never overwritten before it was read.
//
// Window layout (doubles; identical on every rank, so a sender needs no table of its peers):
//   flag[world] (uint64) | scal[2][world][PF_HALO_NSCAL] | recv[2][world][cap]
// `cap` = the largest message any rank sends to one neighbour in any plan (agreed on the host).
#include "pf_common.h"

#define PF_HALO_MAXPLANS 8

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
  long long timeout_ticks = 1000000000LL;  // 10 s of the 100 MHz wall clock: ranks reach their first exchange skewed by
                                           // their own setup (pattern, hierarchy, code-object load)
  pf_halo_plan plans[PF_HALO_MAXPLANS];
  int nplans = 0;
};

__host__ __device__ static inline size_t halo_flag_doubles(int world) { return (size_t)((world + 7) & ~7); }
__host__ __device__ static inline size_t halo_scal_off(int world) { return halo_flag_doubles(world); }
__host__ __device__ static inline size_t halo_recv_off(int world) { return halo_scal_off(world) + (size_t)2 * world * PF_HALO_NSCAL; }

// NB workgroups per peer (NB = gridDim.y): scalars to every peer, the packed values to the neighbours of the plan; the
// workgroup that finishes a peer's message last raises that peer's flag (counted in done[peer], which it resets)
__device__ __forceinline__ void halo_send(int peer, int part, int nparts, int world, int rank, int64_t cap,
                                          double* const* __restrict__ peer_win, int nnbr, const int32_t* __restrict__ nbr,
                                          const int32_t* __restrict__ nbr_ptr, const int32_t* __restrict__ send_idx,
                                          const double* __restrict__ v, const double* __restrict__ scal_in, int ns,
                                          uint64_t seq, uint32_t* __restrict__ done, unsigned long long* __restrict__ sent) {
  // peer includes this rank: its own scalars go through its own window too
  double* w = peer_win[peer];
  const int par = (int)(seq & 1);
  if (part == 0) {
    double* scal = w + halo_scal_off(world) + ((size_t)par * world + rank) * PF_HALO_NSCAL;
    if ((int)threadIdx.x < ns) scal[threadIdx.x] = scal_in[threadIdx.x];
  }
  if (peer != rank && nnbr > 0) {
    // is this peer a neighbour of the plan?  (nbr is short: linear scan by every thread, same result)
    int k = -1;
    for (int i = 0; i < nnbr; ++i) k = (nbr[i] == peer) ? i : k;
    if (k >= 0) {
      double* dst = w + halo_recv_off(world) + ((size_t)par * world + rank) * cap;
      const int32_t b = nbr_ptr[k], e = nbr_ptr[k + 1];
      for (int32_t i = b + part * (int32_t)blockDim.x + (int32_t)threadIdx.x; i < e; i += nparts * (int32_t)blockDim.x)
        dst[i - b] = v[send_idx[i]];
    }
  }
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    bool last = true;
    if (nparts > 1) {
      const uint32_t prev = atomicAdd(&done[peer], 1u);
      last = prev + 1 == (uint32_t)nparts;
      if (last) done[peer] = 0;  // nobody else touches it before the next exchange (stream order)
      __threadfence_system();
    }
    if (last)
      __hip_atomic_store(reinterpret_cast<uint64_t*>(w) + rank, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    // this workgroup has read everything it needs from v (the barrier above is behind every thread's loads): the
    // receiving workgroups of the same launch may overwrite the interface entries once all senders have said so
    __hip_atomic_fetch_add(sent, 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// every workgroup waits for ALL ranks' messages of this exchange, then sums its share of the interface entries;
// workgroup 0 also reduces the scalars.  Sums run over the contributing ranks in ascending order.
__device__ __forceinline__ void halo_recv(int rb, int nrb, int world, int rank, int64_t cap, double* __restrict__ win,
                                          int64_t n_if, const int32_t* __restrict__ if_idx,
                                          const int32_t* __restrict__ src_ptr, const int32_t* __restrict__ src_off,
                                          double* __restrict__ v, double* __restrict__ scal_out, int ns, uint64_t seq,
                                          int32_t* __restrict__ err, long long timeout_ticks,
                                          const unsigned long long* __restrict__ sent, unsigned long long sent_target) {
  __shared__ int bad;
  if (threadIdx.x == 0) bad = *err;
  __syncthreads();
  if ((int)threadIdx.x < world && !bad) {
    const uint64_t* flag = reinterpret_cast<const uint64_t*>(win) + threadIdx.x;
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
      if (wall_clock64() - t0 > timeout_ticks) { bad = 1; break; }
      __builtin_amdgcn_s_sleep(2);
    }
  }
  if (threadIdx.x == 0 && !bad) {  // ... and for this rank's own senders (same launch) to be done reading v
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(sent, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < sent_target) {
      if (wall_clock64() - t0 > timeout_ticks) { bad = 1; break; }
      __builtin_amdgcn_s_sleep(1);
    }
  }
  __syncthreads();
  if (bad) {  // a peer never arrived: flag it and leave (the host raises; the grid drains)
    if (threadIdx.x == 0) *err = 1;
    return;
  }
  __threadfence_system();
  const int par = (int)(seq & 1);
  const double* recv = win + halo_recv_off(world) + (size_t)par * world * cap;
  for (int64_t i = rb * (int64_t)blockDim.x + threadIdx.x; i < n_if; i += (int64_t)nrb * blockDim.x) {
    const int32_t li = if_idx[i];
    const double own = v[li];
    double s = 0.0;
    for (int32_t k = src_ptr[i]; k < src_ptr[i + 1]; ++k) {
      const int32_t o = src_off[k];
      s += (o < 0) ? own : __builtin_nontemporal_load(recv + o);
    }
    v[li] = s;
  }
  if (rb == 0 && (int)threadIdx.x < ns) {
    const double* scal = win + halo_scal_off(world) + (size_t)par * world * PF_HALO_NSCAL;
    double s = 0.0;
    for (int r = 0; r < world; ++r) s += __builtin_nontemporal_load(scal + (size_t)r * PF_HALO_NSCAL + threadIdx.x);
    scal_out[threadIdx.x] = s;
  }
}

// ONE launch per exchange: the first world * nparts workgroups send, the rest receive.  The launch is far smaller than
// the device (at most 8 * 16 + 256 workgroups), so senders and receivers are resident together; a receiver waits for the
// flags of all ranks AND for this rank's own senders to have read the vector before it overwrites interface entries.
__global__ __launch_bounds__(256) void k_halo_xchg(int nparts, int world, int rank, int64_t cap, double* const* __restrict__ peer_win,
                                                   int nnbr, const int32_t* __restrict__ nbr, const int32_t* __restrict__ nbr_ptr,
                                                   const int32_t* __restrict__ send_idx, double* __restrict__ v,
                                                   const double* __restrict__ scal_in, int ns, uint64_t seq,
                                                   uint32_t* __restrict__ done, unsigned long long* __restrict__ sent,
                                                   unsigned long long sent_target, double* __restrict__ win, int64_t n_if,
                                                   const int32_t* __restrict__ if_idx, const int32_t* __restrict__ src_ptr,
                                                   const int32_t* __restrict__ src_off, double* __restrict__ scal_out,
                                                   int32_t* __restrict__ err, long long timeout_ticks) {
  const int nsend = world * nparts;
  if ((int)blockIdx.x < nsend)
    halo_send((int)blockIdx.x % world, (int)blockIdx.x / world, nparts, world, rank, cap, peer_win, nnbr, nbr, nbr_ptr, send_idx, v,
              scal_in, ns, seq, done, sent);
  else
    halo_recv((int)blockIdx.x - nsend, (int)gridDim.x - nsend, world, rank, cap, win, n_if, if_idx, src_ptr, src_off, v, scal_out, ns,
              seq, err, timeout_ticks, sent, sent_target);
}

extern "C" int pf_halo_create(pf_ctx* ctx, int world, int rank, int64_t cap_doubles, pf_halo** out) {
  if (!ctx) return pf_fail(nullptr, PF_E_INVALID, "ctx is NULL");
  PF_REQUIRE(out && world >= 1 && world <= 256 && rank >= 0 && rank < world && cap_doubles >= 0, "pf_halo_create: bad arguments");
  PF_HIP(hipSetDevice(ctx->device));
  pf_halo* h = new pf_halo();
  h->ctx = ctx;
  h->world = world;
  h->rank = rank;
  h->cap = (cap_doubles + 7) & ~(int64_t)7;
  h->win_bytes = (halo_recv_off(world) + (size_t)2 * world * h->cap) * sizeof(double);
  // fine-grained: stores from a peer GPU and the flag that follows them become visible to a running kernel here
  hipError_t e = hipExtMallocWithFlags((void**)&h->win, h->win_bytes, hipDeviceMallocFinegrained);
  if (e != hipSuccess) {
    delete h;
    return pf_fail(ctx, PF_E_HIP, std::string("pf_halo_create: hipExtMallocWithFlags(finegrained) failed: ") + hipGetErrorString(e));
  }
  PF_HIP(hipMemset(h->win, 0, h->win_bytes));
  PF_HIP(hipMalloc((void**)&h->peer_win, sizeof(double*) * world));
  h->mapped.assign(world, nullptr);
  std::vector<double*> pw(world, nullptr);
  pw[rank] = h->win;
  PF_HIP(hipMemcpy(h->peer_win, pw.data(), sizeof(double*) * world, hipMemcpyHostToDevice));
  PF_HIP(hipMalloc((void**)&h->err, sizeof(int32_t)));
  PF_HIP(hipMemset(h->err, 0, sizeof(int32_t)));
  PF_HIP(hipMalloc((void**)&h->done, sizeof(uint32_t) * world));
  PF_HIP(hipMemset(h->done, 0, sizeof(uint32_t) * world));
  PF_HIP(hipMalloc((void**)&h->sent, sizeof(unsigned long long)));
  PF_HIP(hipMemset(h->sent, 0, sizeof(unsigned long long)));
  // hipMemset of device memory may return before it ran: a peer's first flag must not be zeroed by a late memset
  PF_HIP(hipDeviceSynchronize());
  *out = h;
  return PF_OK;
}

extern "C" int pf_halo_handle(pf_halo* h, void* handle64) {
  if (!h) return pf_fail(nullptr, PF_E_INVALID, "halo is NULL");
  pf_ctx* ctx = h->ctx;
  PF_REQUIRE(handle64, "pf_halo_handle: NULL output");
  static_assert(sizeof(hipIpcMemHandle_t) == PF_HALO_HANDLE_BYTES, "IPC handle size");
  hipIpcMemHandle_t hd;
  PF_HIP(hipIpcGetMemHandle(&hd, h->win));
  memcpy(handle64, &hd, sizeof(hd));
  return PF_OK;
}

extern "C" int pf_halo_connect(pf_halo* h, int peer, const void* handle64) {
  if (!h) return pf_fail(nullptr, PF_E_INVALID, "halo is NULL");
  pf_ctx* ctx = h->ctx;
  PF_REQUIRE(handle64 && peer >= 0 && peer < h->world && peer != h->rank, "pf_halo_connect: bad peer");
  PF_HIP(hipSetDevice(ctx->device));
  hipIpcMemHandle_t hd;
  memcpy(&hd, handle64, sizeof(hd));
  void* p = nullptr;
  PF_HIP(hipIpcOpenMemHandle(&p, hd, hipIpcMemLazyEnablePeerAccess));
  h->mapped[peer] = p;
  PF_HIP(hipMemcpy(h->peer_win + peer, &p, sizeof(double*), hipMemcpyHostToDevice));
  return PF_OK;
}

template <typename T>
static int halo_upload(pf_ctx* ctx, const T* host, int64_t n, T*& dev) {
  PF_HIP(hipMalloc((void**)&dev, sizeof(T) * (size_t)std::max<int64_t>(n, 1)));
  if (n > 0) PF_HIP(hipMemcpy(dev, host, sizeof(T) * (size_t)n, hipMemcpyHostToDevice));
  return PF_OK;
}

extern "C" int pf_halo_plan_create(pf_halo* h, int nnbr, const int32_t* nbr_rank, const int32_t* nbr_ptr,
                                   const int32_t* send_idx, int64_t n_if, const int32_t* if_idx, const int32_t* src_ptr,
                                   const int32_t* src_rank, const int32_t* src_pos, int32_t* plan_id) {
  if (!h) return pf_fail(nullptr, PF_E_INVALID, "halo is NULL");
  pf_ctx* ctx = h->ctx;
  PF_REQUIRE(plan_id && nnbr >= 0 && n_if >= 0 && h->nplans < PF_HALO_MAXPLANS, "pf_halo_plan_create: bad arguments / too many plans");
  PF_REQUIRE(nnbr == 0 || (nbr_rank && nbr_ptr && send_idx), "pf_halo_plan_create: NULL neighbour tables");
  PF_REQUIRE(n_if == 0 || (if_idx && src_ptr && src_rank && src_pos), "pf_halo_plan_create: NULL interface tables");
  for (int k = 0; k < nnbr; ++k) {
    PF_REQUIRE(nbr_rank[k] >= 0 && nbr_rank[k] < h->world && nbr_rank[k] != h->rank && (k == 0 || nbr_rank[k] > nbr_rank[k - 1]),
               "pf_halo_plan_create: neighbour ranks must be ascending, in range and exclude this rank");
    PF_REQUIRE(nbr_ptr[k + 1] - nbr_ptr[k] <= h->cap, "pf_halo_plan_create: a message exceeds the window capacity");
  }
  const int64_t nsrc = n_if ? src_ptr[n_if] : 0;
  std::vector<int32_t> off((size_t)std::max<int64_t>(nsrc, 1));
  for (int64_t k = 0; k < nsrc; ++k) {
    if (src_rank[k] == h->rank) { off[k] = -1; continue; }
    PF_REQUIRE(src_rank[k] >= 0 && src_rank[k] < h->world && src_pos[k] >= 0 && src_pos[k] < h->cap,
               "pf_halo_plan_create: source outside the window");
    PF_REQUIRE((int64_t)src_rank[k] * h->cap + src_pos[k] < ((int64_t)1 << 31), "pf_halo_plan_create: window too large for int32 offsets");
    off[k] = (int32_t)((int64_t)src_rank[k] * h->cap + src_pos[k]);
  }
  PF_HIP(hipSetDevice(ctx->device));
  pf_halo_plan& P = h->plans[h->nplans];
  P.nnbr = nnbr;
  P.n_if = n_if;
  P.max_msg = 0;
  for (int k = 0; k < nnbr; ++k) P.max_msg = std::max<int64_t>(P.max_msg, nbr_ptr[k + 1] - nbr_ptr[k]);
  int rc;
  if ((rc = halo_upload(ctx, nbr_rank, nnbr, P.nbr))) return rc;
  if ((rc = halo_upload(ctx, nbr_ptr, nnbr ? nnbr + 1 : 0, P.nbr_ptr))) return rc;
  if ((rc = halo_upload(ctx, send_idx, nnbr ? nbr_ptr[nnbr] : 0, P.send_idx))) return rc;
  if ((rc = halo_upload(ctx, if_idx, n_if, P.if_idx))) return rc;
  if ((rc = halo_upload(ctx, src_ptr, n_if ? n_if + 1 : 0, P.src_ptr))) return rc;
  if ((rc = halo_upload(ctx, off.data(), nsrc, P.src_off))) return rc;
  *plan_id = h->nplans++;
  return PF_OK;
}

// internal: one exchange on the ctx stream.  plan < 0: scalars only.  v may be NULL with plan < 0.
int pf_halo_exchange_impl(pf_halo* h, int plan, double* v, const double* scal_in, int ns, double* scal_out) {
  pf_ctx* ctx = h->ctx;
  PF_REQUIRE(plan < h->nplans && ns >= 0 && ns <= PF_HALO_NSCAL && (ns == 0 || (scal_in && scal_out)), "pf_halo_exchange: bad arguments");
  PF_REQUIRE(plan < 0 || v, "pf_halo_exchange: NULL vector");
  for (int p = 0; p < h->world; ++p)
    PF_REQUIRE(p == h->rank || h->mapped[p], "pf_halo_exchange: not every peer's window is connected (pf_halo_connect)");
  static const pf_halo_plan none;
  const pf_halo_plan& P = plan >= 0 ? h->plans[plan] : none;
  const uint64_t seq = ++h->seq;
  // interface rows of a strip (4 K doubles) go through one workgroup per peer; the level-1 right-hand side of the AMG
  // cycle (~100 K doubles per peer) through up to 16
  const unsigned nparts = (unsigned)std::max<int64_t>(1, std::min<int64_t>(cdiv(P.max_msg, 8192), 16));
  const unsigned g = (unsigned)std::max<int64_t>(1, std::min<int64_t>(cdiv(P.n_if, 256), 256));
  h->sent_target += (unsigned long long)h->world * nparts;
  k_halo_xchg<<<h->world * nparts + g, 256, 0, ctx->stream>>>((int)nparts, h->world, h->rank, h->cap, h->peer_win, P.nnbr, P.nbr,
                                                              P.nbr_ptr, P.send_idx, v, scal_in, ns, seq, h->done, h->sent,
                                                              h->sent_target, h->win, P.n_if, P.if_idx, P.src_ptr, P.src_off,
                                                              scal_out, h->err, h->timeout_ticks);
  PF_LAUNCH_CHECK();
  return PF_OK;
}

extern "C" int pf_halo_exchange(pf_halo* h, int plan, double* v, const double* scal_in, int ns, double* scal_out) {
  if (!h) return pf_fail(nullptr, PF_E_INVALID, "halo is NULL");
  return pf_halo_exchange_impl(h, plan, v, scal_in, ns, scal_out);
}

// Synchronises the stream and reports whether any wait timed out (HOST output: 0 fine, 1 timed out).
extern "C" int pf_halo_status(pf_halo* h, int32_t* err_host) {
  if (!h) return pf_fail(nullptr, PF_E_INVALID, "halo is NULL");
  pf_ctx* ctx = h->ctx;
  PF_REQUIRE(err_host, "pf_halo_status: NULL output");
  PF_HIP(hipMemcpyAsync(err_host, h->err, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  PF_HIP(hipStreamSynchronize(ctx->stream));
  return PF_OK;
}

extern "C" int pf_halo_set_timeout(pf_halo* h, double seconds) {
  if (!h) return pf_fail(nullptr, PF_E_INVALID, "halo is NULL");
  h->timeout_ticks = (long long)(seconds * 1e8);
  return PF_OK;
}

int32_t* pf_halo_err_ptr(pf_halo* h) { return h->err; }
pf_ctx* pf_halo_ctx(pf_halo* h) { return h->ctx; }

extern "C" int pf_halo_destroy(pf_halo* h) {
  if (!h) return PF_OK;
  (void)hipSetDevice(h->ctx->device);
  (void)hipDeviceSynchronize();
  for (int p = 0; p < h->world; ++p)
    if (h->mapped[p]) (void)hipIpcCloseMemHandle(h->mapped[p]);
  for (int i = 0; i < h->nplans; ++i) {
    pf_halo_plan& P = h->plans[i];
    pf_free(P.nbr); pf_free(P.nbr_ptr); pf_free(P.send_idx); pf_free(P.if_idx); pf_free(P.src_ptr); pf_free(P.src_off);
  }
  pf_free(h->peer_win);
  pf_free(h->err);
  pf_free(h->done);
  pf_free(h->sent);
  if (h->win) (void)hipFree(h->win);
  delete h;
  return PF_OK;
}
