// Feasibility probe for the peer-store halo exchange (pf_halo.hip): two PROCESSES on one node share a device window
// through hipIpcGetMemHandle / hipIpcOpenMemHandle and ping-pong a message + flag written by kernels straight into
// the peer's memory.  Run:  ipc_probe [dev0 dev1]   (defaults 0 0: both ranks on the same GPU, what a 1-GPU box can
// show; on a multi-GPU node give two devices to cross xGMI).  Prints which allocation kinds can be shared and the
// round-trip latency per exchange.
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/bin/ipc_probe tools/micro/ipc_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/wait.h>
#include <unistd.h>

#include <chrono>

#define CK(call)                                                                              \
  do {                                                                                        \
    hipError_t e_ = (call);                                                                   \
    if (e_ != hipSuccess) {                                                                   \
      fprintf(stderr, "[rank %d] %s failed: %s\n", g_rank, #call, hipGetErrorString(e_));     \
      return 2;                                                                               \
    }                                                                                         \
  } while (0)

static int g_rank = 0;

struct Window {          // layout inside the shared allocation
  uint64_t flag[2];      // flag[0]: written by rank 0's kernels into rank 1's window and vice versa (use index = sender)
  uint64_t pad[14];
  double buf[2][4096];   // parity x payload
};

__global__ void k_send(Window* remote, int sender, const double* src, int n, uint64_t seq) {
  double* dst = remote->buf[seq & 1];
  for (int i = threadIdx.x; i < n; i += blockDim.x) dst[i] = src[i] + (double)seq;
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(&remote->flag[sender], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ void k_wait(Window* mine, int sender, double* dst, int n, uint64_t seq, int* err, long long timeout_ticks) {
  __shared__ int bad;
  if (threadIdx.x == 0) {
    bad = *err;  // after the first timeout every later wait gives up at once: the probe ends within seconds
    const long long t0 = wall_clock64();
    while (!bad && __hip_atomic_load(&mine->flag[sender], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
      if (wall_clock64() - t0 > timeout_ticks) { bad = 1; break; }
      __builtin_amdgcn_s_sleep(4);
    }
  }
  __syncthreads();
  if (bad) { if (threadIdx.x == 0) *err = 1; return; }
  const double* src = mine->buf[seq & 1];
  for (int i = threadIdx.x; i < n; i += blockDim.x) dst[i] = __builtin_nontemporal_load(src + i);
}

static int run(int rank, int dev, int kind, int rd, int wr) {
  g_rank = rank;
  CK(hipSetDevice(dev));
  Window* win = nullptr;
  const char* kname = kind == 0 ? "hipMalloc (coarse-grained)" : kind == 1 ? "hipExtMallocWithFlags(Finegrained)" : "hipExtMallocWithFlags(Uncached)";
  hipError_t e = kind == 0 ? hipMalloc((void**)&win, sizeof(Window))
               : hipExtMallocWithFlags((void**)&win, sizeof(Window), kind == 1 ? hipDeviceMallocFinegrained : hipDeviceMallocUncached);
  if (e != hipSuccess) { fprintf(stderr, "[rank %d] %s: alloc failed: %s\n", rank, kname, hipGetErrorString(e)); return 3; }
  CK(hipMemset(win, 0, sizeof(Window)));
  hipIpcMemHandle_t mine, theirs;
  e = hipIpcGetMemHandle(&mine, win);
  if (e != hipSuccess) { fprintf(stderr, "[rank %d] %s: hipIpcGetMemHandle failed: %s\n", rank, kname, hipGetErrorString(e)); return 4; }
  if (write(wr, &mine, sizeof(mine)) != (ssize_t)sizeof(mine)) return 5;
  if (read(rd, &theirs, sizeof(theirs)) != (ssize_t)sizeof(theirs)) return 5;
  Window* peer = nullptr;
  e = hipIpcOpenMemHandle((void**)&peer, theirs, hipIpcMemLazyEnablePeerAccess);
  if (e != hipSuccess) { fprintf(stderr, "[rank %d] %s: hipIpcOpenMemHandle failed: %s\n", rank, kname, hipGetErrorString(e)); return 6; }
  const int n = 4096;
  double *src, *dst;
  int* err;
  CK(hipMalloc((void**)&src, n * 8));
  CK(hipMalloc((void**)&dst, n * 8));
  CK(hipMalloc((void**)&err, 4));
  CK(hipMemset(err, 0, 4));
  double* h = (double*)malloc(n * 8);
  for (int i = 0; i < n; ++i) h[i] = 1000.0 * rank + i;
  CK(hipMemcpy(src, h, n * 8, hipMemcpyHostToDevice));
  char tok = 1;  // both windows are mapped on both sides before anybody writes
  if (write(wr, &tok, 1) != 1 || read(rd, &tok, 1) != 1) return 5;
  const long long timeout = 100000000LL * 3;  // 3 s of the 100 MHz wall clock
  const int iters = 2000;
  auto t0 = std::chrono::steady_clock::now();
  for (int it = 1; it <= iters; ++it) {
    // rank 0: send, then wait for the echo; rank 1: wait, then send (a dependent round trip per iteration)
    if (rank == 0) {
      k_send<<<1, 256>>>(peer, 0, src, n, (uint64_t)it);
      k_wait<<<1, 256>>>(win, 1, dst, n, (uint64_t)it, err, timeout);
    } else {
      k_wait<<<1, 256>>>(win, 0, dst, n, (uint64_t)it, err, timeout);
      k_send<<<1, 256>>>(peer, 1, src, n, (uint64_t)it);
    }
  }
  CK(hipDeviceSynchronize());
  const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / iters;
  int herr = 0;
  CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(h, dst, n * 8, hipMemcpyDeviceToHost));
  int wrong = 0;
  for (int i = 0; i < n; ++i) wrong += h[i] != 1000.0 * (1 - rank) + i + iters;
  printf("[rank %d dev %d] %-36s timeout=%d wrong=%d  %.2f us per round trip (%d-double messages)\n", rank, dev, kname, herr, wrong, us, n);
  fflush(stdout);
  if (write(wr, &tok, 1) != 1 || read(rd, &tok, 1) != 1) return 5;  // nobody unmaps before both are done
  (void)hipIpcCloseMemHandle(peer);
  (void)hipFree(win);
  return (herr || wrong) ? 7 : 0;
}

int main(int argc, char** argv) {
  const int dev0 = argc > 2 ? atoi(argv[1]) : 0, dev1 = argc > 2 ? atoi(argv[2]) : 0;
  int worst = 0;
  for (int kind = 0; kind < 3; ++kind) {
    int p01[2], p10[2];
    if (pipe(p01) || pipe(p10)) return 1;
    const pid_t pid = fork();  // BEFORE any HIP call in this process
    if (pid == 0) {
      close(p01[1]); close(p10[0]);
      _exit(run(1, dev1, kind, p01[0], p10[1]));
    }
    // the parent must not touch HIP before the next fork either: run rank 0 in a second child
    const pid_t pid0 = fork();
    if (pid0 == 0) {
      close(p01[0]); close(p10[1]);
      _exit(run(0, dev0, kind, p10[0], p01[1]));
    }
    close(p01[0]); close(p01[1]); close(p10[0]); close(p10[1]);
    int st0 = 0, st1 = 0;
    waitpid(pid0, &st0, 0);
    waitpid(pid, &st1, 0);
    const int r0 = WIFEXITED(st0) ? WEXITSTATUS(st0) : 99, r1 = WIFEXITED(st1) ? WEXITSTATUS(st1) : 99;
    printf("kind %d: rank0 rc=%d rank1 rc=%d\n", kind, r0, r1);
    fflush(stdout);
    if (r0 > worst) worst = r0;
    if (r1 > worst) worst = r1;
  }
  return 0;  // a kind that cannot be shared is a finding, not a failure of the probe
}
