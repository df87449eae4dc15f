// Measurement helpers of bench.py and tools/ (include/pyfem_hip_bench.h): HIP-event timing of the library's own kernel
// families on the library's stream, and the in-process HBM stream probe.  Not on the product path.
#include "pf_common.h"
#include "pyfem_hip_bench.h"

// ------------------------------------------------------------------------------------------
extern "C" int pf_time_kernel(pf_ctx* ctx, int which, int reps, const double* in0, const double* in1, double* out0,
                              double E, double nu, double p, double* ms_host) {
  if (!ctx) return pf_fail(nullptr, PF_E_INVALID, "ctx is NULL");
  PF_REQUIRE(ctx->d && reps > 0 && ms_host, "pf_time_kernel: bad arguments");
  hipEvent_t e0, e1;
  PF_HIP(hipEventCreate(&e0));
  PF_HIP(hipEventCreate(&e1));
  int rc = PF_OK;
  // pass 0 = warm-up ON THE TIMED KERNEL: after the device has run another kernel (or idled) the first ~100 launches are up
  // to 10 % slower (profiles/r03_spmv_placement.md, r02_launch_series.json); one warm-up launch measured that ramp
  for (int pass = 0; pass < 2 && rc == PF_OK; ++pass) {
    const int nrep = pass ? reps : std::max(reps, 100);
    PF_HIP(hipEventRecord(e0, ctx->stream));
    for (int i = 0; i < nrep && rc == PF_OK; ++i) {
      if (which == PF_K_ASSEMBLE_ELASTICITY && ctx->ndims == 3) rc = pf_assemble_elasticity3d(ctx, E, nu, p, in0, out0);
      else if (which == PF_K_ASSEMBLE_ELASTICITY) rc = pf_launch_assemble_elasticity(ctx, E, nu, p, in0, PF_SCATTER_GATHER, out0);
      else if (which == PF_K_SPMV) rc = pf_launch_spmv(ctx, in0, in1, out0, nullptr);
      else if (which == PF_K_ASSEMBLE_HELMHOLTZ && ctx->ndims == 2) rc = pf_launch_assemble_helmholtz(ctx, E, out0, nullptr);
      else rc = pf_fail(ctx, PF_E_INVALID, "pf_time_kernel: unknown kernel");
    }
    PF_HIP(hipEventRecord(e1, ctx->stream));
    PF_HIP(hipEventSynchronize(e1));
    float ms = 0;
    PF_HIP(hipEventElapsedTime(&ms, e0, e1));
    *ms_host = (double)ms / nrep;
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return rc;
}

// ------------------------------------------------------------------------------------------
// HBM ceiling probe (measurement helper of bench.py): what this device sustains in this process for a pure write,
// a pure read and the assembly's 1:3 read:write mix, with the access shape the kernels of this library use
// (persistent waves, 16 bytes per lane, 1 KiB per wave access).
// ------------------------------------------------------------------------------------------
typedef double pf_probe_f64x2 __attribute__((ext_vector_type(2)));
// NT: 0 plain accesses, 1 non-temporal loads and stores, 2 non-temporal stores only (what the assembly kernels do)
template <int MODE, int NT>
__global__ __launch_bounds__(256) void k_hbm_probe(pf_probe_f64x2* __restrict__ a, const pf_probe_f64x2* __restrict__ b, int64_t n2,
                                                   double* __restrict__ sink) {
  // Every WAVE streams whole 64 KiB chunks (dealt round-robin to the waves), 16 bytes per lane and access, four accesses
  // in flight: the walk that gets the most out of the memory system (tools/micro/fillpat.hip, r02: a chip-wide store
  // stream reaches 4.3 TB/s in 1 KiB pieces scattered by a grid-stride loop, 5.5 TB/s in the 4.6 KB pieces of the quad
  // assembly's tile segments, 6.2 TB/s in 64 KiB runs per wave; hipMemset 6.5).
  constexpr int64_t CH2 = 4096;  // 16-byte pieces per chunk = 64 KiB
  constexpr int U = 4;
  const int lane = threadIdx.x & 63;
  const int64_t wave = blockIdx.x * (int64_t)(blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
  double s = 0.0;
  for (int64_t c0 = wave * CH2; c0 < n2; c0 += nwaves * CH2) {
    const int64_t c1 = min(n2, c0 + CH2);
    for (int64_t i0 = c0 + lane; i0 < c1; i0 += 64 * U) {
      pf_probe_f64x2 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t i = min(i0 + 64 * u, c1 - 1);
        if constexpr (MODE == PF_HBM_FILL) v[u] = pf_probe_f64x2{(double)u, 1.0};
        else if constexpr (NT == 1) v[u] = __builtin_nontemporal_load(b + i);
        else v[u] = b[i];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t i = i0 + 64 * u;
        if (i >= c1) continue;
        if constexpr (MODE == PF_HBM_READ) {
          s += v[u].x + v[u].y;
        } else {
          constexpr int NW = MODE == PF_HBM_MIX ? 3 : 1;
#pragma unroll
          for (int k = 0; k < NW; ++k) {
            const pf_probe_f64x2 w = {v[u].x + k, v[u].y - k};
            // MIX: the three written runs of a chunk lie next to each other (one 192 KiB run per 64 KiB read), not in three
            // arrays n2 apart -- four streams whose relative placement then decides the result
            pf_probe_f64x2* dst = (MODE == PF_HBM_MIX) ? a + 3 * c0 + k * (c1 - c0) + (i - c0) : a + i;
            if constexpr (NT != 0) __builtin_nontemporal_store(w, dst);
            else *dst = w;
          }
        }
      }
    }
  }
  if (MODE == PF_HBM_READ && s == 0.12345) sink[0] = s;  // keeps the loads alive
}

extern "C" int pf_hbm_probe(pf_ctx* ctx, int mode, int nt, int64_t nbytes, int reps, double* ms_host) {
  if (!ctx) return pf_fail(nullptr, PF_E_INVALID, "ctx is NULL");
  PF_REQUIRE(mode >= PF_HBM_FILL && mode <= PF_HBM_MIX && nbytes >= 4096 && reps > 0 && ms_host, "pf_hbm_probe: bad arguments");
  PF_HIP(hipSetDevice(ctx->device));
  // FILL: a = nbytes; READ: b = nbytes; MIX: b = nbytes/4 read, a = 3 nbytes/4 written
  const int64_t n2 = (mode == PF_HBM_MIX ? nbytes / 4 : nbytes) / 16;
  pf_probe_f64x2 *a = nullptr, *b = nullptr;
  if (mode != PF_HBM_READ) PF_HIP(hipMalloc(&a, (size_t)n2 * 16 * (mode == PF_HBM_MIX ? 3 : 1)));
  if (mode != PF_HBM_FILL) {
    PF_HIP(hipMalloc(&b, (size_t)n2 * 16));
    PF_HIP(hipMemsetAsync(b, 0, (size_t)n2 * 16, ctx->stream));
  }
  hipEvent_t e0, e1;
  PF_HIP(hipEventCreate(&e0));
  PF_HIP(hipEventCreate(&e1));
  const unsigned grid = 256 * 6;  // 24 waves per CU
  auto launch = [&]() {
#define PF_PROBE(M_)                                                                                            \
    if (mode == M_) {                                                                                           \
      if (nt == 1) k_hbm_probe<M_, 1><<<grid, 256, 0, ctx->stream>>>(a, b, n2, ctx->scal);                      \
      else if (nt == 2) k_hbm_probe<M_, 2><<<grid, 256, 0, ctx->stream>>>(a, b, n2, ctx->scal);                 \
      else k_hbm_probe<M_, 0><<<grid, 256, 0, ctx->stream>>>(a, b, n2, ctx->scal);                              \
    }
    PF_PROBE(PF_HBM_FILL) PF_PROBE(PF_HBM_READ) PF_PROBE(PF_HBM_MIX)
#undef PF_PROBE
  };
  for (int i = 0; i < 3; ++i) launch();  // warm-up
  PF_HIP(hipEventRecord(e0, ctx->stream));
  for (int i = 0; i < reps; ++i) launch();
  PF_HIP(hipEventRecord(e1, ctx->stream));
  PF_HIP(hipEventSynchronize(e1));
  PF_LAUNCH_CHECK();
  float ms = 0;
  PF_HIP(hipEventElapsedTime(&ms, e0, e1));
  *ms_host = (double)ms / reps;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if (a) (void)hipFree(a);
  if (b) (void)hipFree(b);
  return PF_OK;
}
