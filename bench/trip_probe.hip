// What one Fiat-Shamir round trip costs beyond its kernel, and what launching the NEXT round's kernel ahead of its challenge buys
// ("declined on estimates, not measurements"). Standalone: hipcc --offload-arch=gfx950 -O3 -o trip_probe trip_probe.hip
//   A  the product's mechanism: launch (challenge in the kernel arguments) -> kernel -> flag in the host page -> the proving thread sees it
//   B  the kernel of round j+1 is enqueued while round j runs and spins on a doorbell word in the host page; the proving thread writes the
//      challenge + doorbell when it has it. Launch and dispatch latency are off the chain; what stays is one PCIe read of the doorbell.
// `work` = dependent 64-bit multiply-adds per thread (about 30 ns each): the kernel body a trip carries (7-9 us for k_cubic_bind2_eval). Each body is
// stamped with the shader clock it ran at (clock64 against the 100 MHz wall clock): a kernel that follows an idle gap runs at the same 2.2-2.4 GHz
// as one in a back-to-back queue, i.e. the fixed cost of a trip is launch + dispatch + completion, not a clock that has to ramp up.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t body(uint64_t x, int work) {
  for (int i = 0; i < work; i++) x = x * 6364136223846793005ull + 1442695040888963407ull;
  return x;
}
__device__ __forceinline__ void done(volatile uint32_t* flag, uint32_t* counter, uint32_t seq) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence_system();
    if (gridDim.x > 1) {
      if (atomicAdd(counter, 1u) != gridDim.x - 1) return;
      __threadfence_system();
      *counter = 0;
    }
    *flag = seq;
  }
}
// stamps[2*seq], [2*seq+1]: shader cycles (clock64) and 100 MHz ticks (wall_clock64) the body took in workgroup 0 -> the clock the body ran at
__global__ void __launch_bounds__(256) k_args(volatile uint32_t* flag, uint32_t* counter, uint32_t seq, uint64_t challenge, uint64_t* sink, int work, long long* stamps) {
  const long long c0 = clock64(), w0 = wall_clock64();
  uint64_t x = body(challenge + threadIdx.x, work);
  if (x == 42) *sink = x;
  if (stamps && blockIdx.x == 0 && threadIdx.x == 0) { stamps[2 * (seq & 4095)] = clock64() - c0; stamps[2 * (seq & 4095) + 1] = wall_clock64() - w0; }
  done(flag, counter, seq);
}
__global__ void __launch_bounds__(256) k_bell(volatile uint32_t* flag, uint32_t* counter, uint32_t seq, const uint32_t* bell, const uint64_t* challenge, uint64_t* sink,
                                              int work) {
  __shared__ uint64_t ch;
  if (threadIdx.x == 0) {
    const long long t0 = wall_clock64();  // 100 MHz
    bool ok = true;
    while (__hip_atomic_load(bell, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
      if (wall_clock64() - t0 > 200000000ll) { ok = false; break; }  // 2 s: never hang the box
      __builtin_amdgcn_s_sleep(2);
    }
    ch = ok ? __hip_atomic_load(challenge, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0xdeadull;
  }
  __syncthreads();
  uint64_t x = body(ch + threadIdx.x, work);
  if (x == 42) *sink = x;
  done(flag, counter, seq);
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static bool wait_flag(volatile uint32_t* flag, uint32_t seq) {
  const double t0 = now_us();
  while (*flag != seq) {
    if (now_us() - t0 > 3e6) { fprintf(stderr, "flag %u never arrived (have %u)\n", seq, *flag); return false; }
  }
  return true;
}
#include <algorithm>
#include <vector>
static double median_mhz(const long long* stamps, int n) {  // stamps live in host-mapped memory
  std::vector<double> v;
  for (int i = 1; i < std::min(n, 4096); i++) if (stamps[2 * i + 1] > 0) v.push_back((double)stamps[2 * i] / (double)stamps[2 * i + 1] * 100.0);
  if (v.empty()) return 0;
  std::sort(v.begin(), v.end());
  return v[v.size() / 2];
}
int main(int argc, char** argv) {
  const int trips = argc > 1 ? atoi(argv[1]) : 2000;
  uint8_t* page = nullptr;
  CHK(hipHostMalloc((void**)&page, 4096, hipHostMallocMapped));
  volatile uint32_t* flag = (volatile uint32_t*)page;          // device -> host
  uint32_t* bell = (uint32_t*)(page + 256);                     // host -> device
  uint64_t* challenge = (uint64_t*)(page + 512);
  uint32_t* counter; uint64_t* sink;
  CHK(hipMalloc((void**)&counter, 64)); CHK(hipMalloc((void**)&sink, 64)); CHK(hipMemset(counter, 0, 64));
  hipStream_t st; CHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  long long* stamps = nullptr; CHK(hipHostMalloc((void**)&stamps, 4096 * 16, hipHostMallocMapped)); memset(stamps, 0, 4096 * 16);
  const int works[3] = {8, 400, 900};
  const int grids[2] = {1, 16};
  for (int gi = 0; gi < 2; gi++)
    for (int wi = 0; wi < 3; wi++) {
      const int work = works[wi], grid = grids[gi];
      double tA = 0, tB = 0, mhzA = 0, mhzBB = 0;
      for (int rep = 0; rep < 2; rep++) {  // rep 0 warms up
        *flag = 0; *bell = 0; CHK(hipStreamSynchronize(st));
        // A: launch with the challenge in the arguments, wait for the flag
        double t0 = now_us();
        for (uint32_t j = 1; j <= (uint32_t)trips; j++) {
          hipLaunchKernelGGL(k_args, dim3(grid), dim3(256), 0, st, flag, counter, j, (uint64_t)j * 77, sink, work, stamps);
          if (!wait_flag(flag, j)) return 1;
        }
        tA = (now_us() - t0) / trips;
        CHK(hipStreamSynchronize(st));
        mhzA = median_mhz(stamps, trips);
        // B: one launch ahead, doorbell in the host page
        *flag = 0; *bell = 0;
        hipLaunchKernelGGL(k_bell, dim3(grid), dim3(256), 0, st, flag, counter, 1u, bell, challenge, sink, work);
        t0 = now_us();
        for (uint32_t j = 1; j <= (uint32_t)trips; j++) {
          *challenge = (uint64_t)j * 77;
          __atomic_store_n(bell, j, __ATOMIC_RELEASE);  // the challenge of trip j is known: ring
          if (j < (uint32_t)trips) hipLaunchKernelGGL(k_bell, dim3(grid), dim3(256), 0, st, flag, counter, j + 1, bell, challenge, sink, work);  // the next trip's kernel, ahead
          if (!wait_flag(flag, j)) return 1;
        }
        tB = (now_us() - t0) / trips;
        CHK(hipStreamSynchronize(st));
      }
      // the kernel alone (events over back-to-back launches)
      hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
      CHK(hipEventRecord(e0, st));
      for (int j = 0; j < 200; j++) hipLaunchKernelGGL(k_args, dim3(grid), dim3(256), 0, st, flag, counter, (uint32_t)j, 1ull, sink, work, stamps);
      CHK(hipEventRecord(e1, st)); CHK(hipEventSynchronize(e1));
      float ms = 0; CHK(hipEventElapsedTime(&ms, e0, e1));
      mhzBB = median_mhz(stamps, 200);
      printf("grid %2d x 256, body %3d mul-adds: back-to-back launch %.2f us (body at %.0f MHz) | A launch-per-trip %.2f us/trip (body at %.0f MHz) | B launched ahead + doorbell %.2f us/trip\n",
             grid, work, ms * 1e3 / 200, mhzBB, tA, mhzA, tB);
    }
  return 0;
}
