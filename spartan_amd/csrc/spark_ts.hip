// spartan_amd: AddrTimestamps::new (src/sparse_mlpoly.rs:221-254) on the device.
//
// The reference walks the operations of all address lists in order and keeps one counter per memory cell:
// read_ts[op] = how many earlier operations touched the same cell, audit_ts[cell] = how many touched it in total.
// That sequential scan is a stable sort in disguise: sort (address, position) pairs by address, and the rank of an
// operation inside its run of equal addresses is its read timestamp; the run length is the cell's audit timestamp.
// Radix sort and the run-start scan come from rocPRIM; the rest is three streaming kernels.
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/functional.hpp>

#include "internal.hpp"

__global__ void __launch_bounds__(256) k_ts_iota(uint32_t* __restrict__ pos, size_t n) { SP_FG_PRIO();
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) pos[i] = (uint32_t)i;
}
// start[q] = q where a run of equal keys begins, 0 elsewhere (an inclusive max-scan then carries the run start forward)
__global__ void __launch_bounds__(256) k_ts_run_heads(const uint32_t* __restrict__ key, size_t n, uint32_t* __restrict__ start) { SP_FG_PRIO();
  for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (size_t)gridDim.x * blockDim.x)
    start[q] = (q == 0 || key[q] != key[q - 1]) ? (uint32_t)q : 0u;
}
// read_ts of the operation at sorted position q is q - start[q]; the last operation of a run also fixes the cell's audit_ts.
// ts_dst[list][i] and audit_dst[cell] are F_q tables (DensePolynomial::from_usize, dense_mlpoly.rs:274-280).
struct TsDst {
  Fq* p[8];
};
__global__ void __launch_bounds__(256) k_ts_scatter(const uint32_t* __restrict__ key, const uint32_t* __restrict__ pos,
                                                    const uint32_t* __restrict__ start, size_t n, size_t per_list, TsDst dst,
                                                    Fq* __restrict__ audit) { SP_FG_PRIO();
  for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (size_t)gridDim.x * blockDim.x) {
    uint32_t rank = (uint32_t)q - start[q], p = pos[q];
    st_fq(dst.p[p / per_list] + p % per_list, fq_from_u64(rank));
    if (q + 1 == n || key[q + 1] != key[q]) st_fq(audit + key[q], fq_from_u64((uint64_t)rank + 1));
  }
}
__global__ void __launch_bounds__(256) k_ts_zero(Fq* __restrict__ t, size_t n) { SP_FG_PRIO();
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) st_fq(t + i, fq_zero());
}

extern "C" int32_t sp_addr_timestamps(sp_ctx* c, sp_index* const* addr, size_t nlists, size_t cells, sp_table* ts_dst, const size_t* ts_off,
                                      sp_table* audit_dst, size_t audit_off) {
  if (!c || !addr || !ts_dst || !ts_off || !audit_dst || nlists == 0 || nlists > 8 || cells == 0 || cells > 0xffffffffULL) return SP_EINVAL;
  size_t per = addr[0] ? addr[0]->n : 0;
  if (per == 0 || audit_off + cells > audit_dst->cap) return SP_EINVAL;
  for (size_t k = 0; k < nlists; k++)
    if (!addr[k] || addr[k]->n != per || ts_off[k] + per > ts_dst->cap) return SP_EINVAL;
  size_t n = per * nlists;
  if (n > 0xffffffffULL) return SP_EINVAL;
  HIPCHK(hipSetDevice(c->dev));
  int bits = 1;
  while (((uint64_t)1 << bits) < cells) bits++;  // addresses are < cells (checked by the caller when it builds the lists)
  size_t tmp_sort = 0, tmp_scan = 0;
  HIPCHK(rocprim::radix_sort_pairs(nullptr, tmp_sort, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const uint32_t*)nullptr, (uint32_t*)nullptr, n, 0u,
                                   (unsigned)bits, c->stream));
  HIPCHK(rocprim::inclusive_scan(nullptr, tmp_scan, (const uint32_t*)nullptr, (uint32_t*)nullptr, n, rocprim::maximum<uint32_t>(), c->stream));
  size_t tmp = tmp_sort > tmp_scan ? tmp_sort : tmp_scan;
  size_t al = (4 * n + 255) & ~(size_t)255;
  // scratch2: [keys_in][keys_out][pos_in][pos_out][start][library temp]
  HIPCHK(hipStreamSynchronize(c->stream));
  SPCHK(ensure(&c->scratch2, &c->scratch2_cap, 5 * al + tmp + 256));
  uint8_t* base = (uint8_t*)c->scratch2;
  uint32_t *kin = (uint32_t*)base, *kout = (uint32_t*)(base + al), *pin = (uint32_t*)(base + 2 * al), *pout = (uint32_t*)(base + 3 * al),
           *start = (uint32_t*)(base + 4 * al);
  void* lib = base + 5 * al;
  for (size_t k = 0; k < nlists; k++) HIPCHK(hipMemcpyAsync(kin + k * per, addr[k]->d, 4 * per, hipMemcpyDeviceToDevice, c->stream));
  TsDst dst;
  for (size_t k = 0; k < 8; k++) dst.p[k] = k < nlists ? ts_dst->d + ts_off[k] : nullptr;
  {
    ProfScope ps(c, PF_SPARK, 4.0 * 10 * (double)n + 32.0 * (double)(n + cells));
    hipLaunchKernelGGL(k_ts_iota, dim3((unsigned)grid_for(n)), dim3(256), 0, c->stream, pin, n);
    HIPCHK(rocprim::radix_sort_pairs(lib, tmp_sort, (const uint32_t*)kin, kout, (const uint32_t*)pin, pout, n, 0u, (unsigned)bits, c->stream));
    hipLaunchKernelGGL(k_ts_run_heads, dim3((unsigned)grid_for(n)), dim3(256), 0, c->stream, (const uint32_t*)kout, n, start);
    HIPCHK(rocprim::inclusive_scan(lib, tmp_scan, (const uint32_t*)start, start, n, rocprim::maximum<uint32_t>(), c->stream));
    hipLaunchKernelGGL(k_ts_zero, dim3((unsigned)grid_for(cells)), dim3(256), 0, c->stream, audit_dst->d + audit_off, cells);
    hipLaunchKernelGGL(k_ts_scatter, dim3((unsigned)grid_for(n)), dim3(256), 0, c->stream, (const uint32_t*)kout, (const uint32_t*)pout,
                       (const uint32_t*)start, n, per, dst, audit_dst->d + audit_off);
  }
  HIPCHK(hipStreamSynchronize(c->stream));  // scratch2 is shared with other calls
  return hipGetLastError() == hipSuccess ? SP_OK : SP_EHIP;
}
