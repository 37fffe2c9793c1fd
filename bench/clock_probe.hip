// bench/clock_probe.hip — what shader clock do latency-bound kernels see? clock64() counts shader cycles (s_memtime),
// wall_clock64() a constant 100 MHz counter; their ratio during a dependent-ALU loop gives the effective sclk. Measured for
// a lone wavefront, for a lone wavefront while a background kernel keeps N CUs busy, and for a full-chip launch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void k_probe(unsigned long long* out, int iters) {
  unsigned long long c0 = clock64(), w0 = wall_clock64();
  unsigned long long x = threadIdx.x + 1;
  for (int i = 0; i < iters; i++) x = x * 6364136223846793005ULL + 1442695040888963407ULL;  // dependent 64-bit mads
  unsigned long long c1 = clock64(), w1 = wall_clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; out[2] = x; }
}
__global__ void k_busy(volatile int* stop, unsigned long long* sink) {
  unsigned long long x = threadIdx.x + blockIdx.x;
  unsigned long long t0 = wall_clock64();
  while (!*stop && wall_clock64() - t0 < 300000000ULL) {  // at most 3 s
    for (int i = 0; i < 4096; i++) x = x * 6364136223846793005ULL + 1442695040888963407ULL;
  }
  if (x == 42) *sink = x;
}
int main() {
  unsigned long long *d, h[3];
  hipMalloc(&d, 64);
  int* stop; hipHostMalloc((void**)&stop, 64, hipHostMallocCoherent | hipHostMallocMapped); *stop = 0;
  unsigned long long* sink; hipMalloc(&sink, 8);
  hipStream_t s1, s2; hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
  auto run = [&](const char* what, int blocks, int threads, int iters) {
    for (int rep = 0; rep < 3; rep++) {
      hipLaunchKernelGGL(k_probe, dim3(blocks), dim3(threads), 0, s1, d, iters);
      hipStreamSynchronize(s1);
      hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
      printf("%-46s iters %7d: %9llu shader cycles in %8.1f us -> %.0f MHz, %.1f cycles per dependent 64-bit mad\n", what, iters, h[0], h[1] * 0.01, h[0] / (h[1] * 0.01),
             (double)h[0] / iters);
    }
  };
  run("lone wavefront, cold", 1, 64, 2000);
  run("lone wavefront", 1, 64, 200000);
  run("full chip (2048 x 256)", 2048, 256, 200000);
  run("lone wavefront right after full chip", 1, 64, 2000);
  for (int nb : {8, 64, 192}) {
    *stop = 0;
    hipLaunchKernelGGL(k_busy, dim3(nb), dim3(256), 0, s2, (volatile int*)stop, sink);
    char buf[96]; snprintf(buf, sizeof buf, "lone wavefront + %d busy workgroups", nb);
    run(buf, 1, 64, 2000);
    run(buf, 1, 64, 200000);
    *stop = 1;
    hipStreamSynchronize(s2);
  }
  return 0;
}
