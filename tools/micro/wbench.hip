// Store-pattern microbenchmark for the assembly copy-out: persistent one-wave blocks write consecutive
// segments of `segpieces` 16-byte pieces (5 x 64-lane stores per segment, clamped duplicates past the end)
// into a 1.2 GB array.  Compares against a flat grid-stride fill.   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(64) void k_seg(double2* __restrict__ out, long nseg, int segpieces, int offpieces, int strided, int nodup) {
  const int lane = threadIdx.x;
  const long stride = strided ? gridDim.x : 1;
  const long t0 = strided ? blockIdx.x : nseg * blockIdx.x / gridDim.x;
  const long t1 = strided ? nseg : nseg * (blockIdx.x + 1) / gridDim.x;
  for (long t = t0; t < t1; t += stride) {
    double2* d = out + offpieces + t * (long)segpieces;
    const double2 v = make_double2((double)t, (double)lane);
#pragma unroll
    for (int u = 0; u < 5; ++u) {
      const int q = lane + 64 * u;
      if (nodup) { if (q < segpieces) d[q] = v; }
      else d[min(q, segpieces - 1)] = v;
    }
  }
}
__global__ void k_fill(double2* __restrict__ out, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    out[i] = make_double2(1.0, 2.0);
}

int main() {
  const long npieces = 75522050;  // 1.208 GB
  double2* out;
  CK(hipMalloc(&out, (npieces + 1024) * 16));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time = [&](auto launch, const char* name) {
    for (int i = 0; i < 3; ++i) launch();
    CK(hipEventRecord(e0));
    for (int i = 0; i < 20; ++i) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-60s %8.1f us  %6.2f TB/s\n", name, ms / 20 * 1e3, npieces * 16.0 / (ms / 20 * 1e-3) / 1e12);
  };
  time([&] { k_fill<<<256 * 32, 256>>>(out, npieces); }, "flat fill, 256x32 blocks of 256");
  char name[128];
  for (int segp : {288, 290, 320}) for (int off : {0, 1}) for (int wpc : {16, 24, 32}) for (int strided : {1, 0}) for (int nodup : {0, 1}) {
    if (segp == 320 && nodup) continue;
    const long nseg = npieces / segp;
    snprintf(name, sizeof name, "seg %d pieces, off %d, wpc %d, %s, %s", segp, off, wpc, strided ? "strided" : "contig", nodup ? "predicated" : "clamped-dup");
    time([&] { k_seg<<<256 * wpc, 64>>>(out, nseg, segp, off, strided, nodup); }, name);
  }
  return 0;
}
