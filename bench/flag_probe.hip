// Scratch probe: round-trip latency of one tiny kernel, completion seen (a) through hipEventRecord + hipEventQuery spin,
// (b) through a sequence number the kernel stores in host-mapped memory.   hipcc --offload-arch=gfx950 -O2 bench/flag_probe.hip -o bench/flag_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
__global__ void k_flag(volatile uint32_t* flag, uint32_t seq, uint64_t* out) {
  out[threadIdx.x] = seq + threadIdx.x;
  __syncthreads();
  if (threadIdx.x == 0) { __threadfence_system(); *flag = seq; }
}
__global__ void k_done(volatile uint32_t* flag, uint32_t seq) { __threadfence_system(); *flag = seq; }
__global__ void k_done_nofence(volatile uint32_t* flag, uint32_t seq) { *flag = seq; }
__global__ void k_plain(uint64_t* out, uint32_t seq) { out[threadIdx.x] = seq + threadIdx.x; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  uint8_t* hm; hipHostMalloc((void**)&hm, 65536, hipHostMallocDefault);
  volatile uint32_t* flag = (volatile uint32_t*)hm; uint64_t* out = (uint64_t*)(hm + 4096);
  hipEvent_t ev; hipEventCreateWithFlags(&ev, hipEventDisableTiming);
  const int N = 20000;
  for (int rep = 0; rep < 2; rep++) {
    double t0 = now();
    for (int i = 1; i <= N; i++) {
      hipLaunchKernelGGL(k_plain, dim3(1), dim3(64), 0, st, out, (uint32_t)i);
      hipEventRecord(ev, st);
      while (hipEventQuery(ev) == hipErrorNotReady) {}
      if (out[0] != (uint64_t)i) { printf("bad\n"); return 1; }
    }
    printf("event path: %.2f us per round trip\n", (now() - t0) / N * 1e6);
    *flag = 0;
    t0 = now();
    for (int i = 1; i <= N; i++) {
      hipLaunchKernelGGL(k_flag, dim3(1), dim3(64), 0, st, flag, (uint32_t)(i + rep * N), out);
      while (*flag != (uint32_t)(i + rep * N)) {}
      if (out[0] != (uint64_t)(i + rep * N)) { printf("bad2\n"); return 1; }
    }
    printf("flag  path: %.2f us per round trip\n", (now() - t0) / N * 1e6);
    t0 = now();
    for (int i = 1; i <= N; i++) {
      uint32_t sq = (uint32_t)(i + rep * N + 1000000);
      hipLaunchKernelGGL(k_plain, dim3(1), dim3(64), 0, st, out, sq);
      hipLaunchKernelGGL(k_done, dim3(1), dim3(1), 0, st, flag, sq);
      while (*flag != sq) {}
      if (out[0] != (uint64_t)sq) { printf("bad3\n"); return 1; }
    }
    printf("flag kernel appended: %.2f us per round trip\n", (now() - t0) / N * 1e6);
    t0 = now();
    for (int i = 1; i <= N; i++) {
      uint32_t sq = (uint32_t)(i + rep * N + 2000000);
      hipLaunchKernelGGL(k_plain, dim3(1), dim3(64), 0, st, out, sq);
      hipLaunchKernelGGL(k_done_nofence, dim3(1), dim3(1), 0, st, flag, sq);
      while (*flag != sq) {}
      if (out[0] != (uint64_t)sq || out[63] != (uint64_t)sq + 63) { printf("bad4\n"); return 1; }
    }
    printf("flag kernel appended, no fence: %.2f us per round trip\n", (now() - t0) / N * 1e6);
  }
  hipStreamSynchronize(st);
  return 0;
}
