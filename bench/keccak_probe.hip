// bench/keccak_probe.hip — how long does ONE Keccak-f[1600] take on the GPU when nothing else can run beside it?
//
// Question behind it: could the Fiat–Shamir transcript of the group-free batched sum-check rounds
// (sumcheck.rs:254-424: three scalars absorbed, one challenge drawn per round = ~3 STROBE permutations) live on the
// device, so that a round needs no host round trip? A round trip costs ~10 us beyond its kernel; the
// device transcript wins only if 3 dependent permutations take well under that. Two formulations, both measured as a
// chain of dependent permutations executed by ONE wavefront (that is the situation of a transcript: one state, serial):
//   lanes25 : lane (x + 5 y) holds state word A[x][y]; theta / rho-pi / chi exchange through __shfl (ds_bpermute)
//   serial  : one lane holds all 25 words in registers (the compiler's best straight-line code)
// Build: hipcc --offload-arch=gfx950 -O3 bench/keccak_probe.hip -o bench/keccak_probe ; run: bench/keccak_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>

static const uint64_t RC[24] = {0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x000000000000808bULL, 0x0000000080000001ULL,
                                0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
                                0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL,
                                0x000000000000800aULL, 0x800000008000000aULL, 0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
static const int RHO[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};  // [x + 5 y]
__constant__ uint64_t dRC[24];
__constant__ int dRHO[25];

static inline uint64_t rotl(uint64_t v, int n) { return n ? (v << n) | (v >> (64 - n)) : v; }
static void keccak_host(uint64_t A[25]) {
  for (int r = 0; r < 24; r++) {
    uint64_t C[5], B[25];
    for (int x = 0; x < 5; x++) C[x] = A[x] ^ A[x + 5] ^ A[x + 10] ^ A[x + 15] ^ A[x + 20];
    for (int x = 0; x < 5; x++) { uint64_t D = C[(x + 4) % 5] ^ rotl(C[(x + 1) % 5], 1); for (int y = 0; y < 5; y++) A[x + 5 * y] ^= D; }
    for (int x = 0; x < 5; x++) for (int y = 0; y < 5; y++) B[y + 5 * ((2 * x + 3 * y) % 5)] = rotl(A[x + 5 * y], RHO[x + 5 * y]);
    for (int x = 0; x < 5; x++) for (int y = 0; y < 5; y++) A[x + 5 * y] = B[x + 5 * y] ^ (~B[(x + 1) % 5 + 5 * y] & B[(x + 2) % 5 + 5 * y]);
    A[0] ^= RC[r];
  }
}

__device__ __forceinline__ uint64_t shfl64(uint64_t v, int src) {
  uint32_t lo = __shfl((int)(uint32_t)v, src), hi = __shfl((int)(uint32_t)(v >> 32), src);
  return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t drotl(uint64_t v, int n) { return (v << (n & 63)) | (v >> ((64 - n) & 63)); }
// one wavefront, lanes 0..24 carry the state
__global__ void __launch_bounds__(64) k_lanes25(uint64_t* state, int nperm, long long* cycles) {
  const int lane = threadIdx.x, L = lane < 25 ? lane : 0, x = L % 5, y = L / 5;
  uint64_t a = state[L];
  const int rho = dRHO[L];
  // pi: B[X][Y] = rot(A[x][y]) with X = y, Y = 2x + 3y  =>  this lane (X = x, Y = y) takes the word of lane (x', y') with y' = X, 2x' + 3y' = Y
  const int ys = x, xs = ((y - 3 * ys) % 5 + 5) * 3 % 5;  // 2 x' = Y - 3 y'  =>  x' = 3 (Y - 3 y') mod 5   (3 = 2^-1 mod 5)
  const int pi_src = xs + 5 * ys;
  const int c1 = (x + 1) % 5 + 5 * y, c2 = (x + 2) % 5 + 5 * y;
  long long t0 = wall_clock64();
  for (int p = 0; p < nperm; p++) {
#pragma unroll 1
    for (int r = 0; r < 24; r++) {
      uint64_t c = a ^ shfl64(a, x + 5 * ((y + 1) % 5)) ^ shfl64(a, x + 5 * ((y + 2) % 5)) ^ shfl64(a, x + 5 * ((y + 3) % 5)) ^ shfl64(a, x + 5 * ((y + 4) % 5));
      uint64_t d = shfl64(c, (x + 4) % 5) ^ drotl(shfl64(c, (x + 1) % 5), 1);
      uint64_t e = drotl(a ^ d, rho);
      uint64_t b = shfl64(e, pi_src);
      a = b ^ (~shfl64(b, c1) & shfl64(b, c2));
      if (lane == 0) a ^= dRC[r];
    }
  }
  long long t1 = wall_clock64();
  if (lane < 25) state[lane] = a;
  if (lane == 0) *cycles = t1 - t0;
}
__global__ void __launch_bounds__(64) k_serial(uint64_t* state, int nperm, long long* cycles) {
  if (threadIdx.x != 0) return;
  uint64_t A[25];
  for (int i = 0; i < 25; i++) A[i] = state[i];
  long long t0 = wall_clock64();
  for (int p = 0; p < nperm; p++) {
#pragma unroll 1
    for (int r = 0; r < 24; r++) {
      uint64_t C[5], B[25];
#pragma unroll
      for (int x = 0; x < 5; x++) C[x] = A[x] ^ A[x + 5] ^ A[x + 10] ^ A[x + 15] ^ A[x + 20];
#pragma unroll
      for (int x = 0; x < 5; x++) {
        uint64_t D = C[(x + 4) % 5] ^ drotl(C[(x + 1) % 5], 1);
#pragma unroll
        for (int y = 0; y < 5; y++) A[x + 5 * y] ^= D;
      }
      const int rho[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
#pragma unroll
      for (int x = 0; x < 5; x++)
#pragma unroll
        for (int y = 0; y < 5; y++) B[y + 5 * ((2 * x + 3 * y) % 5)] = drotl(A[x + 5 * y], rho[x + 5 * y]);
#pragma unroll
      for (int x = 0; x < 5; x++)
#pragma unroll
        for (int y = 0; y < 5; y++) A[x + 5 * y] = B[x + 5 * y] ^ (~B[(x + 1) % 5 + 5 * y] & B[(x + 2) % 5 + 5 * y]);
      A[0] ^= dRC[r];
    }
  }
  long long t1 = wall_clock64();
  for (int i = 0; i < 25; i++) state[i] = A[i];
  *cycles = t1 - t0;
}

int main() {
  hipMemcpyToSymbol(HIP_SYMBOL(dRC), RC, sizeof RC);
  hipMemcpyToSymbol(HIP_SYMBOL(dRHO), RHO, sizeof RHO);
  int clk_khz = 0;
  hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeWallClockRate, 0);
  uint64_t h[25], ref[25], *d;
  long long* dc;
  hipMalloc(&d, 200); hipMalloc(&dc, 8);
  const int N = 2000;
  for (int variant = 0; variant < 2; variant++) {
    for (int i = 0; i < 25; i++) h[i] = 0x0123456789abcdefULL * (i + 1);
    memcpy(ref, h, 200);
    for (int p = 0; p < N; p++) keccak_host(ref);
    for (int rep = 0; rep < 2; rep++) {  // second run: warm instruction cache
      hipMemcpy(d, h, 200, hipMemcpyHostToDevice);
      if (variant == 0) k_lanes25<<<1, 64>>>(d, N, dc); else k_serial<<<1, 64>>>(d, N, dc);
      hipDeviceSynchronize();
    }
    uint64_t got[25]; long long cyc = 0;
    hipMemcpy(got, d, 200, hipMemcpyDeviceToHost); hipMemcpy(&cyc, dc, 8, hipMemcpyDeviceToHost);
    double us = (double)cyc / (clk_khz ? clk_khz : 100000) * 1e3 / N;
    printf("%-8s  %s  %.3f us per Keccak-f[1600] (one wavefront, %d dependent permutations; wall clock %d kHz)  -> 3 per sum-check round = %.1f us\n",
           variant == 0 ? "lanes25" : "serial", memcmp(got, ref, 200) == 0 ? "matches host" : "MISMATCH", us, N, clk_khz, 3 * us);
  }
  return 0;
}
