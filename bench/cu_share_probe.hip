// Scratch probe: does a compute-saturated workgroup on one CU slow a latency-bound wave on a NEIGHBOURING CU?
// busy: persistent 1024-thread workgroups at 128 VGPRs (a CU holds exactly one, like k_msm_rows_bg), dense v_mad_u64_u32 work,
//       each records where it landed (XCC_ID, SE_ID, CU_ID). lone: one wave per workgroup, a dependent v_mad_u64_u32 chain timed
//       with s_memrealtime, recording its own location. Output: chain time by (same CU pair as a busy workgroup / other).
// hipcc --offload-arch=gfx950 -O3 bench/cu_share_probe.hip -o bench/cu_share_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t where() {
  uint32_t hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  return ((xcc & 0xf) << 16) | (hw & 0xffff);  // hw: WAVE[3:0] SIMD[5:4] PIPE[7:6] CU[11:8] SH[12] SE[15:13]
}
__global__ void __launch_bounds__(1024) busy(uint32_t* loc, volatile int* stop, uint64_t iters) {
  if (threadIdx.x == 0) loc[blockIdx.x] = where();
  uint64_t a = threadIdx.x + 1, b = blockIdx.x + 3, c = 7, d = 11;
  for (uint64_t i = 0; i < iters; i++) {
#pragma unroll 16
    for (int k = 0; k < 64; k++) { a = a * b + c; c = c * d + a; b = b * a + d; d = d * c + b; }
    if ((i & 63) == 0 && *stop) break;
  }
  asm volatile("v_mov_b32 v127, %0" ::"v"((uint32_t)a) : "v127");  // 128 VGPRs: four waves fill a SIMD
  if (a + b + c + d == 0x1234567) loc[0] = 0;
}
__global__ void __launch_bounds__(64) lone(uint32_t* loc, uint32_t* ticks, int n) {
  uint64_t t0, t1;
  uint64_t a = threadIdx.x + 1, b = 0x9e3779b97f4a7c15ull;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
  for (int i = 0; i < n; i++) { a = a * b + i; a = a * a + b; a = a * b + 1; a = a * a + i; }
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) : "v"((uint32_t)a));
  if (threadIdx.x == 0) { loc[blockIdx.x] = where(); ticks[blockIdx.x] = (uint32_t)(t1 - t0); }
  if (a == 0x1234567) loc[0] = 0;
}
int main(int argc, char** argv) {
  int nbusy = argc > 1 ? atoi(argv[1]) : 128, nlone = 2048;
  uint32_t *bl, *ll, *lt; int* stop;
  CK(hipMalloc(&bl, 4 * 1024)); CK(hipMalloc(&ll, 4 * nlone)); CK(hipMalloc(&lt, 4 * nlone)); CK(hipHostMalloc(&stop, 4)); *stop = 0;
  hipStream_t s_bg, s_fg; int lo, hi;
  CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  CK(hipStreamCreateWithPriority(&s_bg, hipStreamNonBlocking, lo)); CK(hipStreamCreateWithPriority(&s_fg, hipStreamNonBlocking, hi));
  std::vector<uint32_t> hl(nlone), ht(nlone), hb(1024);
  auto run_lone = [&](const char* tag, const std::map<uint32_t, int>* busy_cus) {
    double sum[3] = {0, 0, 0}; int cnt[3] = {0, 0, 0};
    for (int rep = 0; rep < 8; rep++) {  // 64 single-wave workgroups at a time: they land on whatever CUs have room
      hipLaunchKernelGGL(lone, dim3(64), dim3(64), 0, s_fg, ll, lt, 2000);
      CK(hipStreamSynchronize(s_fg));
      CK(hipMemcpy(hl.data(), ll, 4 * 64, hipMemcpyDeviceToHost)); CK(hipMemcpy(ht.data(), lt, 4 * 64, hipMemcpyDeviceToHost));
      for (int i = 0; i < 64; i++) {
        uint32_t cu = (hl[i] >> 8) & 0xff | (hl[i] & 0xf0000);  // (XCC, SE, SH, CU)
        int cls = 0;
        if (busy_cus) cls = busy_cus->count(cu) ? 2 : (busy_cus->count(cu ^ 0x1) ? 1 : 0);  // same CU / pair mate (CU_ID ^ 1) / neither
        sum[cls] += ht[i] * 0.01; cnt[cls]++;  // s_memrealtime: 100 MHz
      }
    }
    printf("%-22s chain time us: no busy neighbour %.1f (n=%d) | pair mate busy %.1f (n=%d) | same CU %.1f (n=%d)\n", tag, cnt[0] ? sum[0] / cnt[0] : 0, cnt[0],
           cnt[1] ? sum[1] / cnt[1] : 0, cnt[1], cnt[2] ? sum[2] / cnt[2] : 0, cnt[2]);
  };
  run_lone("alone", nullptr);
  hipLaunchKernelGGL(busy, dim3(nbusy), dim3(1024), 0, s_bg, bl, stop, (uint64_t)1 << 40);
  struct timespec ts = {0, 3000000}; nanosleep(&ts, nullptr);
  CK(hipMemcpy(hb.data(), bl, 4 * nbusy, hipMemcpyDeviceToHost));
  std::map<uint32_t, int> cus, per_xcc, cuid;
  for (int i = 0; i < nbusy; i++) { cus[(hb[i] >> 8) & 0xff | (hb[i] & 0xf0000)]++; per_xcc[hb[i] >> 16]++; cuid[(hb[i] >> 8) & 0xf]++; }
  printf("busy: %d workgroups on %zu distinct CUs; per XCC:", nbusy, cus.size());
  for (auto& kv : per_xcc) printf(" %u:%d", kv.first, kv.second);
  printf("; CU_ID histogram:");
  for (auto& kv : cuid) printf(" %u:%d", kv.first, kv.second);
  printf("\n");
  run_lone("with busy workgroups", &cus);
  *stop = 1;
  CK(hipStreamSynchronize(s_bg));
  return 0;
}
