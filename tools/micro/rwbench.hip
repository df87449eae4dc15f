// Mixed read/write microbenchmark shaped like the assembly kernel's traffic: per 4608-byte output segment
// one 1 KiB record read (16 B/lane stream) and optionally four 16-byte gathers per lane from a 67 MB table.
// Shows what HBM gives this read/write mix when nothing else (quadrature, LDS) is in the way.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int GATHER>
__global__ __launch_bounds__(64) void k_rw(double2* __restrict__ out, const int4* __restrict__ rec, const double2* __restrict__ X,
                                           long nseg, int segpieces, int strided, long nnodes) {
  const int lane = threadIdx.x;
  const long stride = strided ? gridDim.x : 1;
  const long t0 = strided ? blockIdx.x : nseg * blockIdx.x / gridDim.x;
  const long t1 = strided ? nseg : nseg * (blockIdx.x + 1) / gridDim.x;
  if (t0 >= t1) return;
  int4 r = rec[t0 * 64 + lane];
  for (long t = t0; t < t1; t += stride) {
    const long tn = min(t + stride, t1 - 1);
    const int4 rn = rec[tn * 64 + lane];   // next tile's record in flight during this tile's stores
    double2 v = make_double2((double)r.x, (double)r.w);
    if (GATHER) {
      const long n = t * 16 + (lane >> 2);
      const double2 a = X[min(n, nnodes - 1)], b = X[min(n + 1, nnodes - 1)], c = X[min(n + 2049, nnodes - 1)], d = X[min(n + 2050, nnodes - 1)];
      v.x += a.x + b.x + c.x + d.x; v.y += a.y + b.y + c.y + d.y;
    }
    double2* dd = out + t * (long)segpieces;
#pragma unroll
    for (int u = 0; u < 5; ++u) dd[min(lane + 64 * u, segpieces - 1)] = v;
    r = rn;
  }
}

int main() {
  const long nseg = 262144; const int segp = 288;
  const long nnodes = 2049L * 2049L;
  double2 *out, *X; int4* rec;
  CK(hipMalloc(&out, (nseg * segp + 1024) * 16));
  CK(hipMalloc(&rec, nseg * 64 * 16));
  CK(hipMalloc(&X, nnodes * 16));
  CK(hipMemset(rec, 0, nseg * 64 * 16)); CK(hipMemset(X, 0, nnodes * 16));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time = [&](auto launch, const char* name, double bytes) {
    for (int i = 0; i < 3; ++i) launch();
    CK(hipEventRecord(e0));
    for (int i = 0; i < 20; ++i) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-50s %8.1f us  %6.2f TB/s (compulsory bytes)\n", name, ms / 20 * 1e3, bytes / (ms / 20 * 1e-3) / 1e12);
  };
  char name[128];
  const double wb = (double)nseg * segp * 16, rb = (double)nseg * 1024, xb = (double)nnodes * 16;
  for (int wpc : {16, 24, 32}) for (int strided : {1, 0}) {
    snprintf(name, sizeof name, "write + record stream, wpc %d, %s", wpc, strided ? "strided" : "contig");
    time([&] { k_rw<0><<<256 * wpc, 64>>>(out, rec, X, nseg, segp, strided, nnodes); }, name, wb + rb);
    snprintf(name, sizeof name, "write + records + gathers, wpc %d, %s", wpc, strided ? "strided" : "contig");
    time([&] { k_rw<1><<<256 * wpc, 64>>>(out, rec, X, nseg, segp, strided, nnodes); }, name, wb + rb + xb);
  }
  return 0;
}
