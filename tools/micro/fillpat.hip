// Write-ceiling probe: how the order in which a chip-wide store stream walks memory changes its throughput.
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/bin/fillpat tools/micro/fillpat.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// mode 0: grid-stride, 16 B per lane (every wave writes 1 KiB pieces all over the buffer)
// mode 1: each WORKGROUP owns one contiguous chunk of `chunk` bytes at a time, chunks dealt round-robin to workgroups
// mode 2: each WAVE owns a contiguous chunk
__global__ __launch_bounds__(256) void k_fill(double2* a, int64_t n2, int mode, int64_t chunk2) {
  const double2 v = make_double2(1.0, 2.0);
  if (mode == 0) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n2; i += (int64_t)gridDim.x * blockDim.x) a[i] = v;
  } else if (mode == 1) {
    for (int64_t c = blockIdx.x; c * chunk2 < n2; c += gridDim.x)
      for (int64_t i = c * chunk2 + threadIdx.x; i < min(n2, (c + 1) * chunk2); i += blockDim.x) a[i] = v;
  } else {
    const int lane = threadIdx.x & 63;
    const int64_t w = blockIdx.x * 4 + (threadIdx.x >> 6), nw = (int64_t)gridDim.x * 4;
    for (int64_t c = w; c * chunk2 < n2; c += nw)
      for (int64_t i = c * chunk2 + lane; i < min(n2, (c + 1) * chunk2); i += 64) a[i] = v;
  }
}

int main() {
  const int64_t bytes = 1208352800;  // the K value array of the 4M-quad mesh
  const int64_t n2 = bytes / 16;
  double2* a;
  CK(hipMalloc(&a, bytes));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  struct { int mode; int64_t chunk; int grid; const char* name; } cases[] = {
      {0, 0, 2048, "grid-stride, 2048 x 256"}, {0, 0, 8192, "grid-stride, 8192 x 256"}, {0, 0, 1024, "grid-stride, 1024 x 256"},
      {2, 4608, 1536, "wave-contiguous 4608 B (a quad tile segment), 6144 waves"}, {2, 4608, 1024, "wave-contiguous 4608 B, 4096 waves"},
      {2, 65536, 1536, "wave-contiguous 64 KiB"}, {1, 16384, 2048, "workgroup-contiguous 16 KiB"}, {1, 65536, 2048, "workgroup-contiguous 64 KiB"},
      {1, 1 << 20, 2048, "workgroup-contiguous 1 MiB"}, {1, 1 << 20, 512, "workgroup-contiguous 1 MiB, 512 workgroups"}};
  for (auto& c : cases) {
    for (int i = 0; i < 3; ++i) k_fill<<<c.grid, 256>>>(a, n2, c.mode, c.chunk / 16);
    CK(hipEventRecord(e0));
    for (int i = 0; i < 20; ++i) k_fill<<<c.grid, 256>>>(a, n2, c.mode, c.chunk / 16);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-60s %.1f us  %.2f TB/s\n", c.name, 1e3 * ms / 20, bytes / (ms / 20 * 1e-3) / 1e12);
  }
  for (int i = 0; i < 3; ++i) CK(hipMemsetAsync(a, 0, bytes));
  CK(hipEventRecord(e0));
  for (int i = 0; i < 20; ++i) CK(hipMemsetAsync(a, 0, bytes));
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("%-60s %.1f us  %.2f TB/s\n", "hipMemsetAsync", 1e3 * ms / 20, bytes / (ms / 20 * 1e-3) / 1e12);
  return 0;
}
