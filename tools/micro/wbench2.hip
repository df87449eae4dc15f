// Which write patterns reach the 6.2-6.6 TB/s of a torch fill?  (context for the assembly copy-out)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// one block per chunk of `per` pieces per thread, non-persistent, blocks in address order
template <int PER>
__global__ void k_np(double2* __restrict__ out, long n) {
  const long base = (long)blockIdx.x * blockDim.x * PER;
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const long i = base + u * blockDim.x + threadIdx.x;
    if (i < n) out[i] = make_double2(1.0, (double)u);
  }
}
// persistent, block-strided chunks of PER*blockDim pieces
template <int PER>
__global__ void k_ps(double2* __restrict__ out, long n) {
  const long chunk = (long)blockDim.x * PER;
  for (long base = blockIdx.x * chunk; base < n; base += gridDim.x * chunk) {
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const long i = base + u * blockDim.x + threadIdx.x;
      if (i < n) out[i] = make_double2(1.0, (double)u);
    }
  }
}
int main() {
  const long n = 75522050;
  double2* out;
  CK(hipMalloc(&out, (n + 4096) * 16));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time = [&](auto launch, const char* name) {
    for (int i = 0; i < 3; ++i) launch();
    CK(hipEventRecord(e0));
    for (int i = 0; i < 20; ++i) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-64s %8.1f us  %6.2f TB/s\n", name, ms / 20 * 1e3, n * 16.0 / (ms / 20 * 1e-3) / 1e12);
  };
  time([&] { CK(hipMemsetAsync(out, 0, n * 16, 0)); }, "hipMemsetAsync");
  time([&] { k_np<1><<<(unsigned)((n + 255) / 256), 256>>>(out, n); }, "non-persistent, 256 thr x 1 piece (4 KB per block)");
  time([&] { k_np<4><<<(unsigned)((n + 1023) / 1024), 256>>>(out, n); }, "non-persistent, 256 thr x 4 pieces (16 KB per block)");
  time([&] { k_np<5><<<(unsigned)((n + 319) / 320), 64>>>(out, n); }, "non-persistent, 64 thr x 5 pieces (5 KB per block)");
  time([&] { k_np<8><<<(unsigned)((n + 2047) / 2048), 256>>>(out, n); }, "non-persistent, 256 thr x 8 pieces (32 KB per block)");
  time([&] { k_np<4><<<(unsigned)((n + 4095) / 4096), 1024>>>(out, n); }, "non-persistent, 1024 thr x 4 pieces (64 KB per block)");
  for (int wpc : {8, 16, 32}) {
    char nm[128];
    snprintf(nm, sizeof nm, "persistent, %d x 64 thr x 5 pieces per CU (5 KB chunks)", wpc);
    time([&] { k_ps<5><<<256 * wpc, 64>>>(out, n); }, nm);
    snprintf(nm, sizeof nm, "persistent, %d x 256 thr x 4 pieces per CU (16 KB chunks)", wpc / 4);
    time([&] { k_ps<4><<<256 * wpc / 4, 256>>>(out, n); }, nm);
    snprintf(nm, sizeof nm, "persistent, %d x 256 thr x 16 pieces per CU (64 KB chunks)", wpc / 4);
    time([&] { k_ps<16><<<256 * wpc / 4, 256>>>(out, n); }, nm);
  }
  return 0;
}
