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
//   ONE launch per exchange on the solver's stream (at most PF_HALO_MAXWORLD * 16 + 256 workgroups: senders and
//   receivers are resident together), nothing on the host.
//
// Synchronisation: exchanges are numbered (seq = 1, 2, ...; SPMD, every rank issues the same sequence).  Message
// buffers are double-buffered by seq parity.  A sender may run at most one exchange ahead of a receiver (it needs
// the receiver's message of exchange k to finish k itself), so parity k is never overwritten before it was read.
//
// Window layout: pf_common.h (flags | scalar slots | message slots, each twice: packed and fused protocol).
// `cap` = the largest message any rank sends to one neighbour in any plan (agreed on the host).
//
// FUSED exchange (round 3; pf_solve.hip k_spmv<..., SEND> and k_dcg_update / k_halo_recv_apply below): the product
// kernel itself stores the rows of interface nodes into the neighbours' windows from its epilogue -- interface tiles are
// walked first, so the rows travel while the interior tiles are computed, and there is no pack launch -- and the kernel
// that consumes the product waits for the flags, sums in ascending rank order and goes on with its own work (CG update,
// smoother sweep).  A CG iteration over the partition is TWO launches.  The fused protocol has its own slots, flags and
// sequence numbers (rseq for rows, sseq for scalars).  Slot reuse: a rank starts row message j+1 only after its
// consumer of message j has finished, and that consumer waited for message j of every NEIGHBOUR, each of which had
// finished consuming j-1 before it sent j: so parity (j+1)&1 is free at every neighbour.  This needs every fused plan of
// a window to have the same neighbour list (checked in pf_halo_plan_fuse).  Scalar messages go to every rank and their
// consumers wait for every rank: the same argument with "neighbour" replaced by "rank".
#include "pf_common.h"
#include "pf_hostlogic.h"

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
// the device (at most PF_HALO_MAXWORLD * 16 + 256 = 1280 workgroups of 256 threads: five per compute unit), so senders
// and receivers are resident together; a receiver waits for the
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
  PF_REQUIRE(out && world >= 1 && world <= PF_HALO_MAXWORLD && rank >= 0 && rank < world && cap_doubles >= 0,
             "pf_halo_create: bad arguments (at most 64 ranks: one node)");
  PF_HIP(hipSetDevice(ctx->device));
  pf_halo* h = new pf_halo();
  h->ctx = ctx;
  h->world = world;
  h->rank = rank;
  h->cap = (cap_doubles + 7) & ~(int64_t)7;
  h->win_bytes = halo_win_doubles(world, h->cap) * sizeof(double);
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
  PF_HIP(hipMalloc((void**)&h->rows_done, sizeof(unsigned long long)));
  PF_HIP(hipMemset(h->rows_done, 0, sizeof(unsigned long long)));
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
  P.h_nbr.assign(nbr_rank, nbr_rank + nnbr);
  P.h_nbr_ptr.assign(nbr_ptr, nbr_ptr + (nnbr ? nnbr + 1 : 0));
  P.h_send_idx.assign(send_idx, send_idx + (nnbr ? nbr_ptr[nnbr] : 0));
  P.h_if_idx.assign(if_idx, if_idx + n_if);
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
  const uint64_t seq = h->seq + 1;
  // interface rows of a strip (4 K doubles) go through one workgroup per peer; the level-1 right-hand side of the AMG
  // cycle (~100 K doubles per peer) through up to 16
  const unsigned nparts = (unsigned)std::max<int64_t>(1, std::min<int64_t>(cdiv(P.max_msg, 8192), 16));
  const unsigned g = (unsigned)std::max<int64_t>(1, std::min<int64_t>(cdiv(P.n_if, 256), 256));
  const unsigned long long target = h->sent_target + (unsigned long long)h->world * nparts;
  k_halo_xchg<<<h->world * nparts + g, 256, 0, ctx->stream>>>((int)nparts, h->world, h->rank, h->cap, h->peer_win, P.nnbr, P.nbr,
                                                              P.nbr_ptr, P.send_idx, v, scal_in, ns, seq, h->done, h->sent,
                                                              target, h->win, P.n_if, P.if_idx, P.src_ptr, P.src_off,
                                                              scal_out, h->err, h->timeout_ticks);
  PF_LAUNCH_CHECK();
  // only a launch that was accepted advances the sequence: a failed one must not leave this rank ahead of its peers
  h->seq = seq;
  h->sent_target = target;
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

// ------------------------------------------------------------------------------------------
// fused exchange: tables (once per plan) and the per-launch argument blocks
// ------------------------------------------------------------------------------------------
extern "C" int pf_halo_plan_fuse(pf_halo* h, int plan, const double* own_w) {
  if (!h) return pf_fail(nullptr, PF_E_INVALID, "halo is NULL");
  pf_ctx* ctx = h->ctx;
  PF_REQUIRE(plan >= 0 && plan < h->nplans && own_w, "pf_halo_plan_fuse: bad plan / NULL ownership weights");
  PF_REQUIRE(ctx->d == 1 || ctx->d == 2, "pf_halo_plan_fuse: the fused product exists for 1 and 2 dofs per node (2-D models)");
  PF_HIP(hipSetDevice(ctx->device));
  pf_halo_plan& P = h->plans[plan];
  pf_halo_fuse& F = P.fuse;
  if (F.d) return PF_OK;  // already fused
  const int d = ctx->d;
  const int64_t nn = ctx->nnodes, n = nn * d;
  // every fused plan of a window must talk to the same neighbours (slot-reuse argument in the header comment)
  if (h->fuse_nbr.empty() && h->rseq == 0) h->fuse_nbr = P.h_nbr;
  PF_REQUIRE(h->fuse_nbr == P.h_nbr, "pf_halo_plan_fuse: the fused plans of one window must share their neighbour list");
  // the tables themselves are host logic without HIP (pf_hostlogic.h: built and run under ASan / UBSan by the CPU tests)
  const int tn = d == 1 ? PF_SPMV_TILE1 : PF_SPMV_TILE2;
  const int64_t nt = cdiv(nn, tn);
  std::vector<int32_t> base;
  if (ctx->spmv_perm) {
    base.resize((size_t)nt);
    PF_HIP(hipMemcpy(base.data(), ctx->spmv_perm, sizeof(int32_t) * (size_t)nt, hipMemcpyDeviceToHost));
  }
  std::vector<double> ow((size_t)n);
  PF_HIP(hipMemcpy(ow.data(), own_w, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost));
  std::vector<int32_t> ifslot, cnt, sh_peer, sh_pos, tperm;
  std::vector<uint8_t> flags;
  int32_t n_iftiles = 0;
  const std::string why = pf_host_fuse_tables(nn, d, tn, P.nnbr, P.h_nbr.data(), P.h_nbr_ptr.data(), P.h_send_idx.data(),
                                              (int64_t)P.h_if_idx.size(), P.h_if_idx.data(), base.empty() ? nullptr : base.data(),
                                              ow.data(), ifslot, cnt, sh_peer, sh_pos, tperm, n_iftiles, flags);
  PF_REQUIRE(why.empty(), "pf_halo_plan_fuse: " + why);
  if (sh_peer.empty()) { sh_peer.push_back(0); sh_pos.push_back(0); }
  const int64_t n_ifnodes = (int64_t)cnt.size() - 1;
  int rc;
  if ((rc = halo_upload(ctx, ifslot.data(), nn, F.ifslot))) return rc;
  if ((rc = halo_upload(ctx, cnt.data(), (int64_t)cnt.size(), F.sh_ptr))) return rc;
  if ((rc = halo_upload(ctx, sh_peer.data(), (int64_t)sh_peer.size(), F.sh_peer))) return rc;
  if ((rc = halo_upload(ctx, sh_pos.data(), (int64_t)sh_pos.size(), F.sh_pos))) return rc;
  if ((rc = halo_upload(ctx, tperm.data(), nt, F.tperm))) return rc;
  if ((rc = halo_upload(ctx, flags.data(), (int64_t)flags.size(), F.flags8))) return rc;
  F.n_ifnodes = n_ifnodes;
  F.n_iftiles = n_iftiles;
  F.tile_nodes = tn;
  F.d = d;
  return PF_OK;
}

// The same tables from HOST arrays, no GPU involved (outputs sized by the caller: ifslot nnodes, sh_ptr n_if/d + 1,
// sh_peer / sh_pos nbr_ptr[nnbr]/d, tperm ceil(nnodes / tile_nodes), flags8 nnodes*d; dims = {interface nodes, interface
// tiles, tiles, sharer entries}).  tests/test_dist_cpu.py replays a fused exchange over gloo with them.
extern "C" int pf_host_fuse_tables_c(int64_t nnodes, int d, int tile_nodes, int nnbr, const int32_t* nbr, const int32_t* nbr_ptr,
                                     const int32_t* send_idx, int64_t n_if, const int32_t* if_idx, const int32_t* base_perm,
                                     const double* own_w, int32_t* ifslot, int32_t* sh_ptr, int32_t* sh_peer, int32_t* sh_pos,
                                     int32_t* tperm, uint8_t* flags8, int64_t* dims) {
  std::vector<int32_t> a, b, c, e, t;
  std::vector<uint8_t> f;
  int32_t nift = 0;
  const std::string why = pf_host_fuse_tables(nnodes, d, tile_nodes, nnbr, nbr, nbr_ptr, send_idx, n_if, if_idx, base_perm, own_w, a, b,
                                              c, e, t, nift, f);
  if (!why.empty()) return pf_fail(nullptr, PF_E_INVALID, "pf_host_fuse_tables: " + why);
  if (ifslot) std::copy(a.begin(), a.end(), ifslot);
  if (sh_ptr) std::copy(b.begin(), b.end(), sh_ptr);
  if (sh_peer) std::copy(c.begin(), c.end(), sh_peer);
  if (sh_pos) std::copy(e.begin(), e.end(), sh_pos);
  if (tperm) std::copy(t.begin(), t.end(), tperm);
  if (flags8) std::copy(f.begin(), f.begin() + nnodes * d, flags8);
  if (dims) { dims[0] = (int64_t)b.size() - 1; dims[1] = nift; dims[2] = (int64_t)t.size(); dims[3] = (int64_t)c.size(); }
  return PF_OK;
}

// HOST copies of a fused plan's tables (tests replay them: tests/test_dist_cpu.py, tests/test_gpu_dist.py).  Any output
// may be NULL; dims_host = {n_ifnodes, n_iftiles, ntiles, nsharers, tile_nodes, d}.
extern "C" int pf_halo_fuse_get(pf_halo* h, int plan, int64_t* dims_host, int32_t* ifslot, int32_t* sh_ptr, int32_t* sh_peer,
                                int32_t* sh_pos, int32_t* tperm, uint8_t* flags8) {
  if (!h) return pf_fail(nullptr, PF_E_INVALID, "halo is NULL");
  pf_ctx* ctx = h->ctx;
  PF_REQUIRE(plan >= 0 && plan < h->nplans && h->plans[plan].fuse.d, "pf_halo_fuse_get: the plan is not fused");
  const pf_halo_fuse& F = h->plans[plan].fuse;
  const int64_t nt = cdiv(ctx->nnodes, F.tile_nodes);
  int32_t nsh = 0;
  PF_HIP(hipMemcpy(&nsh, F.sh_ptr + F.n_ifnodes, sizeof(int32_t), hipMemcpyDeviceToHost));
  if (dims_host) { dims_host[0] = F.n_ifnodes; dims_host[1] = F.n_iftiles; dims_host[2] = nt; dims_host[3] = nsh; dims_host[4] = F.tile_nodes; dims_host[5] = F.d; }
  if (ifslot) PF_HIP(hipMemcpy(ifslot, F.ifslot, sizeof(int32_t) * (size_t)ctx->nnodes, hipMemcpyDeviceToHost));
  if (sh_ptr) PF_HIP(hipMemcpy(sh_ptr, F.sh_ptr, sizeof(int32_t) * (size_t)(F.n_ifnodes + 1), hipMemcpyDeviceToHost));
  if (sh_peer && nsh) PF_HIP(hipMemcpy(sh_peer, F.sh_peer, sizeof(int32_t) * (size_t)nsh, hipMemcpyDeviceToHost));
  if (sh_pos && nsh) PF_HIP(hipMemcpy(sh_pos, F.sh_pos, sizeof(int32_t) * (size_t)nsh, hipMemcpyDeviceToHost));
  if (tperm) PF_HIP(hipMemcpy(tperm, F.tperm, sizeof(int32_t) * (size_t)nt, hipMemcpyDeviceToHost));
  if (flags8) PF_HIP(hipMemcpy(flags8, F.flags8, (size_t)(ctx->nnodes * F.d), hipMemcpyDeviceToHost));
  return PF_OK;
}

const pf_halo_fuse* pf_halo_fuse_of(pf_halo* h, int plan) {
  return (h && plan >= 0 && plan < h->nplans && h->plans[plan].fuse.d) ? &h->plans[plan].fuse : nullptr;
}

int pf_halo_fused_send(pf_halo* h, int plan, pf_spmv_send* out) {
  pf_ctx* ctx = h->ctx;
  PF_REQUIRE(plan >= 0 && plan < h->nplans && h->plans[plan].fuse.d, "fused exchange: the plan was not prepared (pf_halo_plan_fuse)");
  for (int p = 0; p < h->world; ++p)
    PF_REQUIRE(p == h->rank || h->mapped[p], "fused exchange: not every peer's window is connected (pf_halo_connect)");
  const pf_halo_plan& P = h->plans[plan];
  const uint64_t rseq = h->rseq + 1;
  out->ifslot = P.fuse.ifslot;
  out->sh_ptr = P.fuse.sh_ptr;
  out->sh_peer = P.fuse.sh_peer;
  out->sh_pos = P.fuse.sh_pos;
  out->peer_win = h->peer_win;
  out->nbr = P.nbr;
  out->stop = nullptr;
  out->rows_done = h->rows_done;
  out->rseq = rseq;
  out->slot_off = (int64_t)(halo_recv2_off(h->world, h->cap) + ((size_t)(rseq & 1) * h->world + h->rank) * h->cap);
  out->rflag_off = (int64_t)(halo_rflag_off(h->world) + h->rank);
  out->n_iftiles = P.fuse.n_iftiles;
  out->nnbr = P.nnbr;
  return PF_OK;
}

void pf_halo_fused_sent(pf_halo* h, int plan) {
  (void)plan;
  ++h->rseq;
}

void pf_halo_fused_recv(pf_halo* h, int plan, bool scalars, pf_halo_recv* out) {
  static const pf_halo_plan none;
  const pf_halo_plan& P = plan >= 0 ? h->plans[plan] : none;
  out->win = h->win;
  out->peer_win = h->peer_win;
  out->nbr = P.nbr;
  out->if_idx = P.if_idx;
  out->src_ptr = P.src_ptr;
  out->src_off = P.src_off;
  out->err = h->err;
  out->timeout_ticks = h->timeout_ticks;
  out->n_if = P.n_if;
  out->cap = h->cap;
  out->rseq = plan >= 0 ? h->rseq : 0;
  out->sseq = scalars ? h->sseq + 1 : 0;
  out->world = h->world;
  out->rank = h->rank;
  out->nnbr = P.nnbr;
}

void pf_halo_fused_scal(pf_halo* h) { ++h->sseq; }

// The receiving half of a fused exchange WITHOUT scalars, for the smoother sweeps of the distributed AMG cycle: the
// product left the raw rows of the interface entries in y (and in the neighbours' windows); this finishes them,
//   y[i] = x[i] + d[i] (b[i] - sum)      (d == NULL: y = b - sum;  b == NULL as well: y = sum)
// with the sum over the sharing ranks in ascending rank order -- the same bits on every sharer.
__global__ __launch_bounds__(256) void k_halo_recv_apply(pf_halo_recv rc, const double* __restrict__ x, const double* __restrict__ d,
                                                         const double* __restrict__ b, double* __restrict__ y) {
  __shared__ int bad;
  if (threadIdx.x == 0) bad = *rc.err;
  __syncthreads();
  if ((int)threadIdx.x < rc.nnbr && !bad) {
    const uint64_t* flag = reinterpret_cast<const uint64_t*>(rc.win) + halo_rflag_off(rc.world) + rc.nbr[threadIdx.x];
    if (pf_wait_flag(flag, rc.rseq, rc.timeout_ticks)) bad = 1;
  }
  __syncthreads();
  if (bad) {
    if (threadIdx.x == 0) *rc.err = 1;
    return;
  }
  const double* recv = rc.win + halo_recv2_off(rc.world, rc.cap) + (size_t)(rc.rseq & 1) * rc.world * rc.cap;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < rc.n_if; i += (int64_t)gridDim.x * blockDim.x) {
    const int32_t li = rc.if_idx[i];
    const double own = y[li];
    double s = 0.0;
    for (int32_t k = rc.src_ptr[i]; k < rc.src_ptr[i + 1]; ++k) {
      const int32_t o = rc.src_off[k];
      s += (o < 0) ? own : pf_ld_sys_f64(recv + o);
    }
    if (b) s = b[li] - s;
    if (d) s = x[li] + d[li] * s;
    y[li] = s;
  }
}

int pf_halo_recv_apply(pf_halo* h, int plan, const double* x, const double* d, const double* b, double* y) {
  pf_ctx* ctx = h->ctx;
  const pf_halo_plan& P = h->plans[plan];
  if (P.n_if == 0) return PF_OK;  // a rank without neighbours (one-rank runs): nothing to finish, no launch
  pf_halo_recv rc;
  pf_halo_fused_recv(h, plan, false, &rc);
  const unsigned g = (unsigned)std::min<int64_t>(cdiv(P.n_if, 256), 64);
  k_halo_recv_apply<<<g, 256, 0, ctx->stream>>>(rc, x, d, b, y);
  PF_LAUNCH_CHECK();
  return PF_OK;
}

int pf_halo_colocated(pf_halo* h) { return h->colocated; }
int pf_halo_world(pf_halo* h) { return h->world; }

// How many ranks of the job run on THIS rank's device.  One in production (one process per GPU); the tests rehearse
// several ranks on one GPU, where kernels that wait for a peer must leave room for that peer's kernels to run.
extern "C" int pf_halo_set_colocated(pf_halo* h, int nranks) {
  if (!h) return pf_fail(nullptr, PF_E_INVALID, "halo is NULL");
  pf_ctx* ctx = h->ctx;
  PF_REQUIRE(nranks >= 1 && nranks <= h->world, "pf_halo_set_colocated: between 1 and the world size");
  h->colocated = nranks;
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
    pf_free(P.fuse.ifslot); pf_free(P.fuse.sh_ptr); pf_free(P.fuse.sh_peer); pf_free(P.fuse.sh_pos); pf_free(P.fuse.tperm);
    pf_free(P.fuse.flags8);
  }
  pf_free(h->peer_win);
  pf_free(h->err);
  pf_free(h->done);
  pf_free(h->sent);
  pf_free(h->rows_done);
  if (h->win) (void)hipFree(h->win);
  delete h;
  return PF_OK;
}
