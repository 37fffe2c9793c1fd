// spartan_amd: HIP kernels (gfx950) and the C ABI of include/spartan_hip.h.
// One context = one GPU = one stream. No CPU fallback: every entry point needs a device.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/spartan_hip.h"
#include "curve.hpp"
#include "field.hpp"
#include "msm.hpp"

using namespace sp;

// ------------------------------------------------------------------------------------------------ host structs
enum ProfFamily {
  PF_GENS_TABLE = 0,
  PF_MSM_ROWS,
  PF_MSM_REDUCE,
  PF_EQ_EXPAND,
  PF_SC_EVAL,
  PF_SC_BIND,
  PF_SC_BIND_EVAL,
  PF_VECMAT,
  PF_DOT,
  PF_REDUCE,
  PF_MISC,
  PF_COUNT
};
static const char* kProfNames[PF_COUNT] = {"gens_table_build", "msm_rows_fixed", "msm_reduce_compress", "eq_expand", "sumcheck_eval",
                                            "table_bind", "sumcheck_bind_eval", "vecmat", "dot", "fq_reduce", "misc"};

struct ProfRec {
  hipEvent_t e0, e1;
  int fam;
};

struct sp_ctx {
  int dev;
  hipStream_t stream;
  // scratch
  void* scratch;
  size_t scratch_cap;
  void* scratch2;
  size_t scratch2_cap;
  uint8_t* pinned;  // host pinned staging
  size_t pinned_cap;
  void* dstage;  // device staging for small host inputs
  size_t dstage_cap;
  // profiling
  int prof_on;
  std::vector<ProfRec> pending;
  std::vector<hipEvent_t> free_events;
  double prof_ms[PF_COUNT];
  uint64_t prof_n[PF_COUNT];
  double prof_bytes[PF_COUNT];
};
struct sp_gens {
  sp_ctx* ctx;
  size_t n;
  Niels* table;  // [n][32][128]
};
struct sp_table {
  sp_ctx* ctx;
  Fq* d;
  size_t cap, len;
};

#define HIPCHK(x)                                                                                   \
  do {                                                                                              \
    hipError_t e_ = (x);                                                                            \
    if (e_ != hipSuccess) {                                                                         \
      fprintf(stderr, "spartan_hip: %s failed: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
      return e_ == hipErrorOutOfMemory ? SP_ENOMEM : SP_EHIP;                                       \
    }                                                                                               \
  } while (0)
#define SPCHK(x)              \
  do {                        \
    int32_t r_ = (x);         \
    if (r_ != SP_OK) return r_; \
  } while (0)

static int32_t ensure(void** p, size_t* cap, size_t need) {
  if (*cap >= need) return SP_OK;
  if (*p) HIPCHK(hipFree(*p));
  *p = nullptr;
  *cap = 0;
  size_t want = need + need / 4 + 4096;
  HIPCHK(hipMalloc(p, want));
  *cap = want;
  return SP_OK;
}
static int32_t ensure_pinned(sp_ctx* c, size_t need) {
  if (c->pinned_cap >= need) return SP_OK;
  HIPCHK(hipStreamSynchronize(c->stream));  // an async copy may still read the old buffer
  if (c->pinned) HIPCHK(hipHostFree(c->pinned));
  c->pinned = nullptr;
  c->pinned_cap = 0;
  size_t want = need * 2 + 4096;
  HIPCHK(hipHostMalloc((void**)&c->pinned, want, hipHostMallocDefault));
  c->pinned_cap = want;
  return SP_OK;
}

struct ProfScope {
  sp_ctx* c;
  int fam;
  hipEvent_t e0, e1;
  bool on;
  ProfScope(sp_ctx* c_, int fam_, double bytes) : c(c_), fam(fam_), on(c_->prof_on != 0) {
    if (!on) return;
    auto get = [&]() {
      hipEvent_t e;
      if (!c->free_events.empty()) {
        e = c->free_events.back();
        c->free_events.pop_back();
      } else {
        (void)hipEventCreate(&e);
      }
      return e;
    };
    e0 = get();
    e1 = get();
    c->prof_bytes[fam] += bytes;
    (void)hipEventRecord(e0, c->stream);
  }
  ~ProfScope() {
    if (!on) return;
    (void)hipEventRecord(e1, c->stream);
    c->pending.push_back(ProfRec{e0, e1, fam});
  }
};
static void prof_drain(sp_ctx* c) {
  if (c->pending.empty()) return;
  (void)hipStreamSynchronize(c->stream);
  for (auto& r : c->pending) {
    float ms = 0;
    (void)hipEventElapsedTime(&ms, r.e0, r.e1);
    c->prof_ms[r.fam] += ms;
    c->prof_n[r.fam] += 1;
    c->free_events.push_back(r.e0);
    c->free_events.push_back(r.e1);
  }
  c->pending.clear();
}

// ------------------------------------------------------------------------------------------------ device helpers
__device__ __forceinline__ Fq ld_fq(const Fq* p) {
  const ulonglong2* q = reinterpret_cast<const ulonglong2*>(p);
  ulonglong2 a = q[0], b = q[1];
  return Fq{{a.x, a.y, b.x, b.y}};
}
__device__ __forceinline__ void st_fq(Fq* p, const Fq& v) {
  ulonglong2* q = reinterpret_cast<ulonglong2*>(p);
  q[0] = make_ulonglong2(v.l[0], v.l[1]);
  q[1] = make_ulonglong2(v.l[2], v.l[3]);
}

// block-wide sum of K Fq values per thread; result valid in thread 0. blockDim.x == 256.
template <int K>
__device__ __forceinline__ void block_sum_fq(Fq (&v)[K], Fq* smem /*256*/) {
  int t = threadIdx.x;
#pragma unroll
  for (int k = 0; k < K; k++) {
    __syncthreads();
    smem[t] = v[k];
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if (t < s) smem[t] = fq_add(smem[t], smem[t + s]);
      __syncthreads();
    }
    v[k] = smem[0];
  }
}

// ------------------------------------------------------------------------------------------------ generators
// stage 1: decode / map points. mode 0: compressed in (32 B); mode 1: uniform in (64 B) -> also writes compressed.
__global__ void k_points_load(const uint8_t* __restrict__ in, int mode, size_t n, Pt* __restrict__ pts, uint8_t* __restrict__ comp_out,
                              int* __restrict__ bad) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Pt p;
  if (mode == 0) {
    uint8_t b[32];
    for (int k = 0; k < 32; k++) b[k] = in[32 * i + k];
    if (!pt_decompress(b, &p)) {
      atomicExch(bad, 1);
      p = pt_identity();
    }
  } else {
    uint8_t b[64];
    for (int k = 0; k < 64; k++) b[k] = in[64 * i + k];
    p = pt_from_uniform_bytes(b);
    if (comp_out) {
      uint8_t c[32];
      pt_compress(p, c);
      for (int k = 0; k < 32; k++) comp_out[32 * i + k] = c[k];
    }
  }
  pts[i] = p;
}
// stage 2: one thread per (point, window): entries k * 2^(8w) * P, k = 1..128, affine Niels.
__global__ void k_table_build(const Pt* __restrict__ pts, size_t n, Niels* __restrict__ table) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * MSM_NWIN) return;
  size_t pt = t / MSM_NWIN;
  int w = (int)(t % MSM_NWIN);
  Pt base = pts[pt];
  for (int k = 0; k < MSM_WBITS * w; k++) base = pt_dbl(base);
  Pt acc = base;
  for (int m = 1; m <= MSM_TENT; m++) {
    Fp zinv = fp_invert(acc.Z);
    table[msm_tidx(pt, w, m)] = pt_to_niels(acc, zinv);
    if (m < MSM_TENT) acc = pt_add(acc, base);
  }
}

// ------------------------------------------------------------------------------------------------ MSM
// thread <-> (row, strip): accumulates sum_{j in strip} Z[row][j] * P[col(j)] into one extended point.
// Lanes run fastest over rows so a wave shares the generator (and its 12 KiB window sub-table) whenever
// rows >= 64: table gathers then hit L1/L2, while the scalar load (32 B per 32 additions) is the strided one.
__global__ void __launch_bounds__(256) k_msm_rows(const Fq* __restrict__ Z, size_t z_row_stride, size_t rows, size_t cols, size_t strip,
                                                  size_t nstrips, const Niels* __restrict__ table, size_t g_off,
                                                  const uint32_t* __restrict__ idx, const Fq* __restrict__ blinds, size_t h_idx,
                                                  Pt* __restrict__ partial) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= rows * nstrips) return;
  size_t row = t % rows, s = t / rows;
  Pt acc = pt_identity();
  size_t j0 = s * strip, j1 = j0 + strip;
  if (j1 > cols) j1 = cols;
  for (size_t j = j0; j < j1; j++) {
    Fq sc = ld_fq(Z + row * z_row_stride + j);
    size_t pt = idx ? (size_t)idx[j] : g_off + j;
    msm_accumulate(acc, sc, table, pt);
  }
  if (blinds && s == 0) msm_accumulate(acc, ld_fq(blinds + row), table, h_idx);
  partial[row * nstrips + s] = acc;
}
// one block per row: sum the row's strip partials, compress.
__global__ void __launch_bounds__(256) k_msm_reduce(const Pt* __restrict__ partial, size_t nstrips, uint8_t* __restrict__ out) {
  __shared__ Pt sm[256];
  size_t row = blockIdx.x;
  int t = threadIdx.x;
  Pt acc = pt_identity();
  bool any = false;
  for (size_t s = t; s < nstrips; s += 256) {
    Pt p = partial[row * nstrips + s];
    acc = any ? pt_add(acc, p) : p;
    any = true;
  }
  sm[t] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (t < s && (size_t)(t + s) < nstrips) sm[t] = pt_add(sm[t], sm[t + s]);
    __syncthreads();
  }
  if (t == 0) {
    uint8_t c[32];
    pt_compress(sm[0], c);
    for (int k = 0; k < 32; k++) out[32 * row + k] = c[k];
  }
}

// ------------------------------------------------------------------------------------------------ F_q streaming kernels
// chi table: thread computes 2^LOWB consecutive entries. r[0] <-> most significant index bit.
constexpr int EQ_LOWB = 4;
__device__ __forceinline__ Fq eq_prefix(const Fq* __restrict__ r, size_t ell, int lowb, size_t hi) {
  // product over the (ell - lowb) high bits of index `hi` (hi = index >> lowb)
  Fq acc = fq_one();
  int nh = (int)ell - lowb;
  for (int k = 0; k < nh; k++) {
    Fq rk = ld_fq(r + k);
    bool bit = (hi >> (nh - 1 - k)) & 1;
    acc = fq_mul(acc, bit ? rk : fq_sub(fq_one(), rk));
  }
  return acc;
}
__device__ __forceinline__ void eq_expand_low(Fq (&v)[1 << EQ_LOWB], const Fq* __restrict__ r, size_t ell, int lowb, const Fq& prefix) {
  v[0] = prefix;
  int size = 1;
  for (int k = 0; k < lowb; k++) {
    Fq rk = ld_fq(r + (ell - lowb + k));
    for (int i = size - 1; i >= 0; i--) {
      Fq hi = fq_mul(v[i], rk);
      v[2 * i + 1] = hi;
      v[2 * i] = fq_sub(v[i], hi);
    }
    size *= 2;
  }
}
__global__ void __launch_bounds__(256) k_eq_expand(const Fq* __restrict__ r, size_t ell, int lowb, Fq* __restrict__ out) {
  size_t nthreads = (size_t)1 << (ell - lowb);
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nthreads) return;
  Fq v[1 << EQ_LOWB];
  eq_expand_low(v, r, ell, lowb, eq_prefix(r, ell, lowb, t));
  int cnt = 1 << lowb;
  for (int i = 0; i < cnt; i++) st_fq(out + (t << lowb) + i, v[i]);
}
// <Z, chi(r)> without materialising chi; per-block partials.
__global__ void __launch_bounds__(256) k_evaluate(const Fq* __restrict__ Z, const Fq* __restrict__ r, size_t ell, int lowb,
                                                  Fq* __restrict__ partials) {
  __shared__ Fq sm[256];
  size_t nthreads = (size_t)1 << (ell - lowb);
  Fq acc[1] = {fq_zero()};
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < nthreads; t += (size_t)gridDim.x * blockDim.x) {
    Fq v[1 << EQ_LOWB];
    eq_expand_low(v, r, ell, lowb, eq_prefix(r, ell, lowb, t));
    int cnt = 1 << lowb;
    for (int i = 0; i < cnt; i++) acc[0] = fq_add(acc[0], fq_mul(v[i], ld_fq(Z + (t << lowb) + i)));
  }
  block_sum_fq<1>(acc, sm);
  if (threadIdx.x == 0) st_fq(partials + blockIdx.x, acc[0]);
}

struct Tabs4 {
  Fq* p[4];
};
// sum-check evaluations at t = 0, 2, 3 of the line through (T[i], T[i+half]); per-block partial sums.
template <int KIND>
__device__ __forceinline__ void sc_point(const Fq& a0, const Fq& a1, const Fq& b0, const Fq& b1, const Fq& c0, const Fq& c1, const Fq& d0,
                                         const Fq& d1, Fq& e0, Fq& e2, Fq& e3) {
  Fq a2 = fq_sub(fq_dbl(a1), a0), b2 = fq_sub(fq_dbl(b1), b0);
  if (KIND == 0) {
    e0 = fq_add(e0, fq_mul(a0, b0));
    e2 = fq_add(e2, fq_mul(a2, b2));
    return;
  }
  Fq a3 = fq_sub(fq_add(a2, a1), a0), b3 = fq_sub(fq_add(b2, b1), b0);
  Fq c2 = fq_sub(fq_dbl(c1), c0), c3 = fq_sub(fq_add(c2, c1), c0);
  if (KIND == 1) {
    e0 = fq_add(e0, fq_mul(fq_mul(a0, b0), c0));
    e2 = fq_add(e2, fq_mul(fq_mul(a2, b2), c2));
    e3 = fq_add(e3, fq_mul(fq_mul(a3, b3), c3));
    return;
  }
  Fq d2 = fq_sub(fq_dbl(d1), d0), d3 = fq_sub(fq_add(d2, d1), d0);
  e0 = fq_add(e0, fq_mul(a0, fq_sub(fq_mul(b0, c0), d0)));
  e2 = fq_add(e2, fq_mul(a2, fq_sub(fq_mul(b2, c2), d2)));
  e3 = fq_add(e3, fq_mul(a3, fq_sub(fq_mul(b3, c3), d3)));
}
template <int KIND>
__global__ void __launch_bounds__(256) k_sc_eval(Tabs4 T, size_t half, Fq* __restrict__ partials) {
  __shared__ Fq sm[256];
  Fq e[3] = {fq_zero(), fq_zero(), fq_zero()};
  Fq z = fq_zero();
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < half; i += (size_t)gridDim.x * blockDim.x) {
    Fq a0 = ld_fq(T.p[0] + i), a1 = ld_fq(T.p[0] + half + i), b0 = ld_fq(T.p[1] + i), b1 = ld_fq(T.p[1] + half + i);
    Fq c0 = z, c1 = z, d0 = z, d1 = z;
    if (KIND >= 1) { c0 = ld_fq(T.p[2] + i); c1 = ld_fq(T.p[2] + half + i); }
    if (KIND == 2) { d0 = ld_fq(T.p[3] + i); d1 = ld_fq(T.p[3] + half + i); }
    sc_point<KIND>(a0, a1, b0, b1, c0, c1, d0, d1, e[0], e[1], e[2]);
  }
  block_sum_fq<3>(e, sm);
  if (threadIdx.x == 0) {
    st_fq(partials + 3 * blockIdx.x + 0, e[0]);
    st_fq(partials + 3 * blockIdx.x + 1, e[1]);
    st_fq(partials + 3 * blockIdx.x + 2, e[2]);
  }
}
// fused bind(r) + evaluate next round. quarter = len/4. Thread i < quarter reads T[i], T[i+q], T[i+2q], T[i+3q],
// writes the bound values T'[i] = T[i] + r (T[i+2q]-T[i]) and T'[i+q], and evaluates the round on (T'[i], T'[i+q]).
template <int KIND>
__global__ void __launch_bounds__(256) k_sc_bind_eval(Tabs4 T, size_t quarter, Fq r, Fq* __restrict__ partials) {
  __shared__ Fq sm[256];
  constexpr int NT = KIND == 0 ? 2 : (KIND == 1 ? 3 : 4);
  Fq e[3] = {fq_zero(), fq_zero(), fq_zero()};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < quarter; i += (size_t)gridDim.x * blockDim.x) {
    Fq lo[4], hi[4];
#pragma unroll
    for (int k = 0; k < NT; k++) {
      Fq x0 = ld_fq(T.p[k] + i), x1 = ld_fq(T.p[k] + quarter + i), x2 = ld_fq(T.p[k] + 2 * quarter + i), x3 = ld_fq(T.p[k] + 3 * quarter + i);
      lo[k] = fq_add(x0, fq_mul(r, fq_sub(x2, x0)));
      hi[k] = fq_add(x1, fq_mul(r, fq_sub(x3, x1)));
      st_fq(T.p[k] + i, lo[k]);
      st_fq(T.p[k] + quarter + i, hi[k]);
    }
    Fq z = fq_zero();
    sc_point<KIND>(lo[0], hi[0], lo[1], hi[1], NT > 2 ? lo[2] : z, NT > 2 ? hi[2] : z, NT > 3 ? lo[3] : z, NT > 3 ? hi[3] : z, e[0], e[1], e[2]);
  }
  block_sum_fq<3>(e, sm);
  if (threadIdx.x == 0) {
    st_fq(partials + 3 * blockIdx.x + 0, e[0]);
    st_fq(partials + 3 * blockIdx.x + 1, e[1]);
    st_fq(partials + 3 * blockIdx.x + 2, e[2]);
  }
}
__global__ void __launch_bounds__(256) k_bind_top(Tabs4 T, int ntabs, size_t half, Fq r) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < half; i += (size_t)gridDim.x * blockDim.x) {
    for (int k = 0; k < ntabs; k++) {
      Fq x0 = ld_fq(T.p[k] + i), x1 = ld_fq(T.p[k] + half + i);
      st_fq(T.p[k] + i, fq_add(x0, fq_mul(r, fq_sub(x1, x0))));
    }
  }
}
// partials[nblk][K] -> out[K] ; single block
__global__ void __launch_bounds__(256) k_reduce_partials(const Fq* __restrict__ partials, size_t nblk, int K, Fq* __restrict__ out) {
  __shared__ Fq sm[256];
  for (int k = 0; k < K; k++) {
    Fq acc[1] = {fq_zero()};
    for (size_t b = threadIdx.x; b < nblk; b += 256) acc[0] = fq_add(acc[0], ld_fq(partials + b * K + k));
    block_sum_fq<1>(acc, sm);
    if (threadIdx.x == 0) st_fq(out + k, acc[0]);
  }
}
// out[i] (+)= sum_{j in chunk} L[j] * Z[j*R + i] ; grid (R/256, nchunks) ; partial[chunk][i]
__global__ void __launch_bounds__(256) k_vecmat(const Fq* __restrict__ L, size_t Lsz, const Fq* __restrict__ Z, size_t R, size_t jchunk,
                                                Fq* __restrict__ partial) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= R) return;
  size_t j0 = (size_t)blockIdx.y * jchunk, j1 = j0 + jchunk;
  if (j1 > Lsz) j1 = Lsz;
  Fq acc = fq_zero();
  for (size_t j = j0; j < j1; j++) acc = fq_add(acc, fq_mul(ld_fq(L + j), ld_fq(Z + j * R + i)));
  st_fq(partial + (size_t)blockIdx.y * R + i, acc);
}
__global__ void __launch_bounds__(256) k_colsum(const Fq* __restrict__ partial, size_t nchunks, size_t R, Fq* __restrict__ out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= R) return;
  Fq acc = fq_zero();
  for (size_t c = 0; c < nchunks; c++) acc = fq_add(acc, ld_fq(partial + c * R + i));
  st_fq(out + i, acc);
}
__global__ void __launch_bounds__(256) k_dot(const Fq* __restrict__ a, const Fq* __restrict__ b, size_t n, Fq* __restrict__ partials) {
  __shared__ Fq sm[256];
  Fq acc[1] = {fq_zero()};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    acc[0] = fq_add(acc[0], fq_mul(ld_fq(a + i), ld_fq(b + i)));
  block_sum_fq<1>(acc, sm);
  if (threadIdx.x == 0) st_fq(partials + blockIdx.x, acc[0]);
}
__global__ void k_gather_heads(Tabs4 T, int ntabs, Fq* __restrict__ out) {
  int k = threadIdx.x;
  if (k < ntabs) st_fq(out + k, ld_fq(T.p[k]));
}

// ------------------------------------------------------------------------------------------------ C ABI
static size_t grid_for(size_t work, size_t maxblocks = 2048) {
  size_t b = (work + 255) / 256;
  if (b < 1) b = 1;
  return b > maxblocks ? maxblocks : b;
}
static bool is_pow2(size_t x) { return x && !(x & (x - 1)); }
static size_t ilog2(size_t x) {
  size_t l = 0;
  while (((size_t)1 << l) < x) l++;
  return l;
}

extern "C" {

const char* sp_strerror(int32_t s) {
  switch (s) {
    case SP_OK: return "ok";
    case SP_EINVAL: return "invalid argument";
    case SP_ENOMEM: return "out of device memory";
    case SP_EHIP: return "HIP runtime error or no gfx950 device";
    case SP_EPOINT: return "invalid ristretto255 encoding";
    default: return "unknown";
  }
}
const char* sp_version(void) { return "spartan_amd 0.1 (gfx950)"; }

int32_t sp_ctx_create(int device_id, sp_ctx** out) {
  if (!out) return SP_EINVAL;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0) {
    fprintf(stderr, "spartan_hip: no HIP device available (this library has no CPU fallback)\n");
    return SP_EHIP;
  }
  if (device_id < 0 || device_id >= ndev) return SP_EINVAL;
  HIPCHK(hipSetDevice(device_id));
  sp_ctx* c = new (std::nothrow) sp_ctx();
  if (!c) return SP_ENOMEM;
  c->dev = device_id;
  c->scratch = c->scratch2 = c->dstage = nullptr;
  c->scratch_cap = c->scratch2_cap = c->dstage_cap = 0;
  c->pinned = nullptr;
  c->pinned_cap = 0;
  c->prof_on = 0;
  memset(c->prof_ms, 0, sizeof c->prof_ms);
  memset(c->prof_n, 0, sizeof c->prof_n);
  memset(c->prof_bytes, 0, sizeof c->prof_bytes);
  HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  *out = c;
  return SP_OK;
}
void sp_ctx_destroy(sp_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->dev);
  prof_drain(c);
  for (auto e : c->free_events) (void)hipEventDestroy(e);
  if (c->scratch) (void)hipFree(c->scratch);
  if (c->scratch2) (void)hipFree(c->scratch2);
  if (c->dstage) (void)hipFree(c->dstage);
  if (c->pinned) (void)hipHostFree(c->pinned);
  (void)hipStreamDestroy(c->stream);
  delete c;
}
int32_t sp_prof_enable(sp_ctx* c, int on) {
  if (!c) return SP_EINVAL;
  prof_drain(c);
  c->prof_on = on;
  return SP_OK;
}
int32_t sp_prof_reset(sp_ctx* c) {
  if (!c) return SP_EINVAL;
  prof_drain(c);
  memset(c->prof_ms, 0, sizeof c->prof_ms);
  memset(c->prof_n, 0, sizeof c->prof_n);
  memset(c->prof_bytes, 0, sizeof c->prof_bytes);
  return SP_OK;
}
int32_t sp_prof_read(sp_ctx* c, const char** names, double* total_ms, uint64_t* launches, double* alg_bytes, int cap) {
  if (!c) return SP_EINVAL;
  prof_drain(c);
  for (int i = 0; i < PF_COUNT && i < cap; i++) {
    if (names) names[i] = kProfNames[i];
    if (total_ms) total_ms[i] = c->prof_ms[i];
    if (launches) launches[i] = c->prof_n[i];
    if (alg_bytes) alg_bytes[i] = c->prof_bytes[i];
  }
  return PF_COUNT;
}

// copy small host data to the device staging buffer at byte offset off
static int32_t stage_in(sp_ctx* c, size_t off, const void* src, size_t bytes) {
  SPCHK(ensure_pinned(c, off + bytes));
  memcpy(c->pinned + off, src, bytes);
  HIPCHK(hipMemcpyAsync((uint8_t*)c->dstage + off, c->pinned + off, bytes, hipMemcpyHostToDevice, c->stream));
  return SP_OK;
}
static int32_t ensure_dstage(sp_ctx* c, size_t need) {
  if (c->dstage_cap >= need) return SP_OK;
  HIPCHK(hipStreamSynchronize(c->stream));
  return ensure(&c->dstage, &c->dstage_cap, need);
}
// device -> host through pinned memory, synchronous
static int32_t fetch_out(sp_ctx* c, const void* dsrc, void* hdst, size_t bytes) {
  SPCHK(ensure_pinned(c, bytes));
  HIPCHK(hipMemcpyAsync(c->pinned, dsrc, bytes, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  memcpy(hdst, c->pinned, bytes);
  return SP_OK;
}

static int32_t gens_build(sp_ctx* c, const uint8_t* in, int mode, size_t n, uint8_t* comp_out, sp_gens** out) {
  if (!c || !in || !out || n == 0) return SP_EINVAL;
  HIPCHK(hipSetDevice(c->dev));
  size_t in_bytes = (mode == 0 ? 32 : 64) * n;
  // scratch layout: [in bytes][pad][Pt n][comp 32n][bad int]
  size_t off_pts = (in_bytes + 255) & ~(size_t)255;
  size_t off_comp = off_pts + n * sizeof(Pt);
  size_t off_bad = off_comp + ((32 * n + 255) & ~(size_t)255);
  HIPCHK(hipStreamSynchronize(c->stream));
  SPCHK(ensure(&c->scratch, &c->scratch_cap, off_bad + 256));
  uint8_t* base = (uint8_t*)c->scratch;
  HIPCHK(hipMemcpyAsync(base, in, in_bytes, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemsetAsync(base + off_bad, 0, 4, c->stream));
  Niels* table = nullptr;
  HIPCHK(hipMalloc((void**)&table, n * MSM_PT_ENTRIES * sizeof(Niels)));
  {
    ProfScope ps(c, PF_GENS_TABLE, (double)n * MSM_PT_ENTRIES * sizeof(Niels));
    hipLaunchKernelGGL(k_points_load, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, c->stream, base, mode, n, (Pt*)(base + off_pts),
                       (mode == 1 && comp_out) ? base + off_comp : (uint8_t*)nullptr, (int*)(base + off_bad));
    size_t nt = n * MSM_NWIN;
    hipLaunchKernelGGL(k_table_build, dim3((unsigned)((nt + 63) / 64)), dim3(64), 0, c->stream, (const Pt*)(base + off_pts), n, table);
  }
  int bad = 0;
  int32_t rc = fetch_out(c, base + off_bad, &bad, 4);
  if (rc == SP_OK && mode == 1 && comp_out) rc = fetch_out(c, base + off_comp, comp_out, 32 * n);
  if (rc == SP_OK && hipGetLastError() != hipSuccess) rc = SP_EHIP;
  if (rc != SP_OK || bad) {
    (void)hipFree(table);
    return rc != SP_OK ? rc : SP_EPOINT;
  }
  sp_gens* g = new (std::nothrow) sp_gens();
  if (!g) { (void)hipFree(table); return SP_ENOMEM; }
  g->ctx = c;
  g->n = n;
  g->table = table;
  *out = g;
  return SP_OK;
}
int32_t sp_gens_upload(sp_ctx* c, const uint8_t* compressed, size_t n, sp_gens** out) { return gens_build(c, compressed, 0, n, nullptr, out); }
int32_t sp_gens_from_uniform(sp_ctx* c, const uint8_t* uniform, size_t n, uint8_t* compressed_out, sp_gens** out) {
  return gens_build(c, uniform, 1, n, compressed_out, out);
}
size_t sp_gens_len(const sp_gens* g) { return g ? g->n : 0; }
void sp_gens_free(sp_gens* g) {
  if (!g) return;
  (void)hipSetDevice(g->ctx->dev);
  (void)hipStreamSynchronize(g->ctx->stream);
  (void)hipFree(g->table);
  delete g;
}

// core: Z on device (row stride in elements), optional idx (device), optional blinds (device)
static int32_t msm_launch(sp_ctx* c, const sp_gens* g, const Fq* dZ, size_t z_stride, size_t rows, size_t cols, size_t g_off,
                          const uint32_t* didx, const Fq* dblinds, size_t h_idx, uint8_t* out_host) {
  size_t total = rows * cols;
  size_t strip = total / 131072;
  if (strip < 1) strip = 1;
  if (strip > cols) strip = cols;
  size_t nstrips = (cols + strip - 1) / strip;
  size_t part_bytes = rows * nstrips * sizeof(Pt);
  size_t off_out = (part_bytes + 255) & ~(size_t)255;
  SPCHK(ensure(&c->scratch, &c->scratch_cap, off_out + 32 * rows));
  Pt* partial = (Pt*)c->scratch;
  uint8_t* dout = (uint8_t*)c->scratch + off_out;
  size_t nthreads = rows * nstrips;
  {
    ProfScope ps(c, PF_MSM_ROWS, 32.0 * (double)total + 32.0 * (double)rows);
    hipLaunchKernelGGL(k_msm_rows, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, c->stream, dZ, z_stride, rows, cols, strip, nstrips,
                       (const Niels*)g->table, g_off, didx, dblinds, h_idx, partial);
  }
  {
    ProfScope ps(c, PF_MSM_REDUCE, (double)part_bytes);
    hipLaunchKernelGGL(k_msm_reduce, dim3((unsigned)rows), dim3(256), 0, c->stream, (const Pt*)partial, nstrips, dout);
  }
  SPCHK(fetch_out(c, dout, out_host, 32 * rows));
  if (hipGetLastError() != hipSuccess) return SP_EHIP;
  return SP_OK;
}

int32_t sp_commit_rows_dev(sp_ctx* c, const sp_gens* g, size_t g_off, size_t h_idx, const sp_table* Z, size_t z_off, size_t rows, size_t cols,
                           const uint64_t* blinds, uint8_t* out) {
  if (!c || !g || !Z || !out || rows == 0 || cols == 0) return SP_EINVAL;
  if (g_off + cols > g->n || (blinds && h_idx >= g->n) || z_off + rows * cols > Z->cap) return SP_EINVAL;
  HIPCHK(hipSetDevice(c->dev));
  const Fq* dbl = nullptr;
  if (blinds) {
    SPCHK(ensure_dstage(c, 32 * rows));
    SPCHK(stage_in(c, 0, blinds, 32 * rows));
    dbl = (const Fq*)c->dstage;
  }
  return msm_launch(c, g, Z->d + z_off, cols, rows, cols, g_off, nullptr, dbl, h_idx, out);
}
int32_t sp_commit_rows(sp_ctx* c, const sp_gens* g, size_t g_off, size_t h_idx, const uint64_t* Z, size_t rows, size_t cols,
                       const uint64_t* blinds, uint8_t* out) {
  if (!c || !g || !Z || !out || rows == 0 || cols == 0) return SP_EINVAL;
  if (g_off + cols > g->n || (blinds && h_idx >= g->n)) return SP_EINVAL;
  HIPCHK(hipSetDevice(c->dev));
  size_t zb = 32 * rows * cols;
  HIPCHK(hipStreamSynchronize(c->stream));
  SPCHK(ensure(&c->scratch2, &c->scratch2_cap, zb + 32 * rows));
  HIPCHK(hipMemcpyAsync(c->scratch2, Z, zb, hipMemcpyHostToDevice, c->stream));
  const Fq* dbl = nullptr;
  if (blinds) {
    HIPCHK(hipMemcpyAsync((uint8_t*)c->scratch2 + zb, blinds, 32 * rows, hipMemcpyHostToDevice, c->stream));
    dbl = (const Fq*)((uint8_t*)c->scratch2 + zb);
  }
  return msm_launch(c, g, (const Fq*)c->scratch2, cols, rows, cols, g_off, nullptr, dbl, h_idx, out);
}
int32_t sp_msm_indexed(sp_ctx* c, const sp_gens* g, const uint32_t* idx, size_t cols, const uint64_t* S, size_t rows, uint8_t* out) {
  if (!c || !g || !idx || !S || !out || rows == 0 || cols == 0) return SP_EINVAL;
  for (size_t j = 0; j < cols; j++)
    if (idx[j] >= g->n) return SP_EINVAL;
  HIPCHK(hipSetDevice(c->dev));
  size_t sb = 32 * rows * cols, ib = (4 * cols + 31) & ~(size_t)31;
  SPCHK(ensure_dstage(c, sb + ib));
  SPCHK(stage_in(c, 0, S, sb));
  SPCHK(stage_in(c, sb, idx, 4 * cols));
  return msm_launch(c, g, (const Fq*)c->dstage, cols, rows, cols, 0, (const uint32_t*)((uint8_t*)c->dstage + sb), nullptr, 0, out);
}

// ---- tables
int32_t sp_table_alloc(sp_ctx* c, size_t len, sp_table** out) {
  if (!c || !out || len == 0) return SP_EINVAL;
  HIPCHK(hipSetDevice(c->dev));
  sp_table* t = new (std::nothrow) sp_table();
  if (!t) return SP_ENOMEM;
  t->ctx = c;
  t->cap = t->len = len;
  t->d = nullptr;
  hipError_t e = hipMalloc((void**)&t->d, 32 * len);
  if (e != hipSuccess) { delete t; return e == hipErrorOutOfMemory ? SP_ENOMEM : SP_EHIP; }
  e = hipMemsetAsync(t->d, 0, 32 * len, c->stream);
  if (e != hipSuccess) { (void)hipFree(t->d); delete t; return SP_EHIP; }
  *out = t;
  return SP_OK;
}
int32_t sp_table_write(sp_ctx* c, sp_table* t, size_t off, const uint64_t* Z, size_t len) {
  if (!c || !t || !Z || off + len > t->cap) return SP_EINVAL;
  HIPCHK(hipSetDevice(c->dev));
  HIPCHK(hipMemcpyAsync(t->d + off, Z, 32 * len, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));  // caller may reuse Z immediately
  return SP_OK;
}
int32_t sp_table_upload(sp_ctx* c, const uint64_t* Z, size_t len, sp_table** out) {
  if (!Z) return SP_EINVAL;
  SPCHK(sp_table_alloc(c, len, out));
  int32_t rc = sp_table_write(c, *out, 0, Z, len);
  if (rc != SP_OK) { sp_table_free(*out); *out = nullptr; }
  return rc;
}
int32_t sp_table_download(sp_ctx* c, const sp_table* t, size_t off, size_t len, uint64_t* out) {
  if (!c || !t || !out || off + len > t->cap) return SP_EINVAL;
  HIPCHK(hipSetDevice(c->dev));
  HIPCHK(hipMemcpyAsync(out, t->d + off, 32 * len, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return SP_OK;
}
int32_t sp_table_clone(sp_ctx* c, const sp_table* t, sp_table** out) {
  if (!t) return SP_EINVAL;
  SPCHK(sp_table_alloc(c, t->cap, out));
  (*out)->len = t->len;
  HIPCHK(hipMemcpyAsync((*out)->d, t->d, 32 * t->cap, hipMemcpyDeviceToDevice, c->stream));
  return SP_OK;
}
int32_t sp_table_copy(sp_ctx* c, sp_table* dst, size_t dst_off, const sp_table* src, size_t src_off, size_t len) {
  if (!c || !dst || !src || dst_off + len > dst->cap || src_off + len > src->cap) return SP_EINVAL;
  HIPCHK(hipSetDevice(c->dev));
  HIPCHK(hipMemcpyAsync(dst->d + dst_off, src->d + src_off, 32 * len, hipMemcpyDeviceToDevice, c->stream));
  return SP_OK;
}
size_t sp_table_len(const sp_table* t) { return t ? t->len : 0; }
void sp_table_free(sp_table* t) {
  if (!t) return;
  (void)hipSetDevice(t->ctx->dev);
  (void)hipStreamSynchronize(t->ctx->stream);
  (void)hipFree(t->d);
  delete t;
}

int32_t sp_eq_expand(sp_ctx* c, const uint64_t* r, size_t ell, sp_table** out) {
  if (!c || !r || !out || ell == 0 || ell > 40) return SP_EINVAL;
  HIPCHK(hipSetDevice(c->dev));
  SPCHK(ensure_dstage(c, 32 * ell));
  SPCHK(stage_in(c, 0, r, 32 * ell));
  size_t len = (size_t)1 << ell;
  SPCHK(sp_table_alloc(c, len, out));
  int lowb = ell < (size_t)EQ_LOWB ? (int)ell : EQ_LOWB;
  size_t nthreads = len >> lowb;
  {
    ProfScope ps(c, PF_EQ_EXPAND, 32.0 * (double)len);
    hipLaunchKernelGGL(k_eq_expand, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, c->stream, (const Fq*)c->dstage, ell, lowb, (*out)->d);
  }
  HIPCHK(hipStreamSynchronize(c->stream));  // dstage reusable
  return hipGetLastError() == hipSuccess ? SP_OK : SP_EHIP;
}

static int32_t tabs_check(sp_ctx* c, sp_table* const* tabs, size_t ntabs, size_t need, Tabs4* T, size_t* len) {
  if (!c || !tabs || ntabs != need) return SP_EINVAL;
  size_t l = tabs[0] ? tabs[0]->len : 0;
  for (size_t k = 0; k < ntabs; k++) {
    if (!tabs[k] || tabs[k]->len != l) return SP_EINVAL;
    T->p[k] = tabs[k]->d;
  }
  for (size_t k = ntabs; k < 4; k++) T->p[k] = nullptr;
  if (l < 2 || !is_pow2(l)) return SP_EINVAL;
  *len = l;
  return SP_OK;
}
static int32_t reduce_and_fetch(sp_ctx* c, Fq* partials, size_t nblk, int K, uint64_t* out) {
  Fq* dres = partials + nblk * K;
  {
    ProfScope ps(c, PF_REDUCE, 32.0 * (double)(nblk * K));
    hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(256), 0, c->stream, (const Fq*)partials, nblk, K, dres);
  }
  SPCHK(fetch_out(c, dres, out, 32 * K));
  return hipGetLastError() == hipSuccess ? SP_OK : SP_EHIP;
}
int32_t sp_sumcheck_eval(sp_ctx* c, int kind, sp_table* const* tabs, size_t ntabs, uint64_t* out_evals) {
  if (kind < 0 || kind > 2 || !out_evals) return SP_EINVAL;
  Tabs4 T;
  size_t len;
  SPCHK(tabs_check(c, tabs, ntabs, kind == 0 ? 2 : (kind == 1 ? 3 : 4), &T, &len));
  HIPCHK(hipSetDevice(c->dev));
  size_t half = len / 2, nblk = grid_for(half, 1024);
  SPCHK(ensure(&c->scratch, &c->scratch_cap, 32 * (nblk * 3 + 3)));
  Fq* partials = (Fq*)c->scratch;
  {
    ProfScope ps(c, PF_SC_EVAL, 32.0 * (double)len * (double)ntabs);
    if (kind == 0) hipLaunchKernelGGL(k_sc_eval<0>, dim3((unsigned)nblk), dim3(256), 0, c->stream, T, half, partials);
    if (kind == 1) hipLaunchKernelGGL(k_sc_eval<1>, dim3((unsigned)nblk), dim3(256), 0, c->stream, T, half, partials);
    if (kind == 2) hipLaunchKernelGGL(k_sc_eval<2>, dim3((unsigned)nblk), dim3(256), 0, c->stream, T, half, partials);
  }
  uint64_t e[12];
  SPCHK(reduce_and_fetch(c, partials, nblk, 3, e));
  memcpy(out_evals, e, 32);
  memcpy(out_evals + 4, e + 4, 32);
  if (kind != 0) memcpy(out_evals + 8, e + 8, 32);
  return SP_OK;
}
int32_t sp_table_bind_top(sp_ctx* c, sp_table* const* tabs, size_t ntabs, const uint64_t r[4]) {
  if (!c || !tabs || !r || ntabs == 0) return SP_EINVAL;
  HIPCHK(hipSetDevice(c->dev));
  Fq rr;
  memcpy(rr.l, r, 32);
  for (size_t k0 = 0; k0 < ntabs; k0 += 4) {
    size_t nk = ntabs - k0 < 4 ? ntabs - k0 : 4;
    Tabs4 T = {{nullptr, nullptr, nullptr, nullptr}};
    size_t len = tabs[k0] ? tabs[k0]->len : 0;
    for (size_t k = 0; k < nk; k++) {
      if (!tabs[k0 + k] || tabs[k0 + k]->len != len) return SP_EINVAL;
      T.p[k] = tabs[k0 + k]->d;
    }
    if (len < 2 || !is_pow2(len)) return SP_EINVAL;
    size_t half = len / 2;
    {
      ProfScope ps(c, PF_SC_BIND, 48.0 * (double)len * (double)nk);
      hipLaunchKernelGGL(k_bind_top, dim3((unsigned)grid_for(half)), dim3(256), 0, c->stream, T, (int)nk, half, rr);
    }
    for (size_t k = 0; k < nk; k++) tabs[k0 + k]->len = half;
  }
  return hipGetLastError() == hipSuccess ? SP_OK : SP_EHIP;
}
int32_t sp_sumcheck_bind_eval(sp_ctx* c, int kind, sp_table* const* tabs, size_t ntabs, const uint64_t r[4], uint64_t* out_evals) {
  if (kind < 0 || kind > 2 || !out_evals || !r) return SP_EINVAL;
  Tabs4 T;
  size_t len;
  SPCHK(tabs_check(c, tabs, ntabs, kind == 0 ? 2 : (kind == 1 ? 3 : 4), &T, &len));
  if (len < 4) return SP_EINVAL;
  HIPCHK(hipSetDevice(c->dev));
  Fq rr;
  memcpy(rr.l, r, 32);
  size_t quarter = len / 4, nblk = grid_for(quarter, 1024);
  SPCHK(ensure(&c->scratch, &c->scratch_cap, 32 * (nblk * 3 + 3)));
  Fq* partials = (Fq*)c->scratch;
  {
    ProfScope ps(c, PF_SC_BIND_EVAL, 48.0 * (double)len * (double)ntabs);
    if (kind == 0) hipLaunchKernelGGL(k_sc_bind_eval<0>, dim3((unsigned)nblk), dim3(256), 0, c->stream, T, quarter, rr, partials);
    if (kind == 1) hipLaunchKernelGGL(k_sc_bind_eval<1>, dim3((unsigned)nblk), dim3(256), 0, c->stream, T, quarter, rr, partials);
    if (kind == 2) hipLaunchKernelGGL(k_sc_bind_eval<2>, dim3((unsigned)nblk), dim3(256), 0, c->stream, T, quarter, rr, partials);
  }
  for (size_t k = 0; k < ntabs; k++) tabs[k]->len = len / 2;
  uint64_t e[12];
  SPCHK(reduce_and_fetch(c, partials, nblk, 3, e));
  memcpy(out_evals, e, 32);
  memcpy(out_evals + 4, e + 4, 32);
  if (kind != 0) memcpy(out_evals + 8, e + 8, 32);
  return SP_OK;
}
int32_t sp_vecmat(sp_ctx* c, const uint64_t* L, size_t Lsz, const sp_table* Z, uint64_t* out) {
  if (!c || !L || !Z || !out || Lsz == 0 || Z->len % Lsz) return SP_EINVAL;
  HIPCHK(hipSetDevice(c->dev));
  size_t R = Z->len / Lsz;
  SPCHK(ensure_dstage(c, 32 * Lsz));
  SPCHK(stage_in(c, 0, L, 32 * Lsz));
  size_t nchunks = Lsz < 64 ? 1 : 64;
  while (nchunks > 1 && (R / 256 + 1) * nchunks > 4096) nchunks /= 2;
  size_t jchunk = (Lsz + nchunks - 1) / nchunks;
  SPCHK(ensure(&c->scratch, &c->scratch_cap, 32 * (nchunks * R + R)));
  Fq* partial = (Fq*)c->scratch;
  Fq* dres = partial + nchunks * R;
  {
    ProfScope ps(c, PF_VECMAT, 32.0 * (double)Z->len + 32.0 * (double)R);
    hipLaunchKernelGGL(k_vecmat, dim3((unsigned)((R + 255) / 256), (unsigned)nchunks), dim3(256), 0, c->stream, (const Fq*)c->dstage, Lsz,
                       (const Fq*)Z->d, R, jchunk, partial);
    hipLaunchKernelGGL(k_colsum, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, c->stream, (const Fq*)partial, nchunks, R, dres);
  }
  SPCHK(fetch_out(c, dres, out, 32 * R));
  return hipGetLastError() == hipSuccess ? SP_OK : SP_EHIP;
}
int32_t sp_dot(sp_ctx* c, const sp_table* a, size_t a_off, const sp_table* b, size_t b_off, size_t n, uint64_t out[4]) {
  if (!c || !a || !b || !out || n == 0 || a_off + n > a->cap || b_off + n > b->cap) return SP_EINVAL;
  HIPCHK(hipSetDevice(c->dev));
  size_t nblk = grid_for(n, 1024);
  SPCHK(ensure(&c->scratch, &c->scratch_cap, 32 * (nblk + 1)));
  Fq* partials = (Fq*)c->scratch;
  {
    ProfScope ps(c, PF_DOT, 64.0 * (double)n);
    hipLaunchKernelGGL(k_dot, dim3((unsigned)nblk), dim3(256), 0, c->stream, (const Fq*)(a->d + a_off), (const Fq*)(b->d + b_off), n, partials);
  }
  return reduce_and_fetch(c, partials, nblk, 1, out);
}
int32_t sp_evaluate(sp_ctx* c, const sp_table* Z, const uint64_t* r, size_t ell, uint64_t out[4]) {
  if (!c || !Z || !r || !out || ell == 0 || ell > 40 || Z->len != ((size_t)1 << ell)) return SP_EINVAL;
  HIPCHK(hipSetDevice(c->dev));
  SPCHK(ensure_dstage(c, 32 * ell));
  SPCHK(stage_in(c, 0, r, 32 * ell));
  int lowb = ell < (size_t)EQ_LOWB ? (int)ell : EQ_LOWB;
  size_t nthreads = Z->len >> lowb, nblk = grid_for(nthreads, 1024);
  SPCHK(ensure(&c->scratch, &c->scratch_cap, 32 * (nblk + 1)));
  Fq* partials = (Fq*)c->scratch;
  {
    ProfScope ps(c, PF_DOT, 32.0 * (double)Z->len);
    hipLaunchKernelGGL(k_evaluate, dim3((unsigned)nblk), dim3(256), 0, c->stream, (const Fq*)Z->d, (const Fq*)c->dstage, ell, lowb, partials);
  }
  return reduce_and_fetch(c, partials, nblk, 1, out);
}
int32_t sp_table_heads(sp_ctx* c, sp_table* const* tabs, size_t ntabs, uint64_t* out) {
  if (!c || !tabs || !out || ntabs == 0) return SP_EINVAL;
  HIPCHK(hipSetDevice(c->dev));
  SPCHK(ensure(&c->scratch, &c->scratch_cap, 32 * 4));
  for (size_t k0 = 0; k0 < ntabs; k0 += 4) {
    size_t nk = ntabs - k0 < 4 ? ntabs - k0 : 4;
    Tabs4 T = {{nullptr, nullptr, nullptr, nullptr}};
    for (size_t k = 0; k < nk; k++) {
      if (!tabs[k0 + k]) return SP_EINVAL;
      T.p[k] = tabs[k0 + k]->d;
    }
    hipLaunchKernelGGL(k_gather_heads, dim3(1), dim3(64), 0, c->stream, T, (int)nk, (Fq*)c->scratch);
    SPCHK(fetch_out(c, c->scratch, out + 4 * k0, 32 * nk));
  }
  return hipGetLastError() == hipSuccess ? SP_OK : SP_EHIP;
}

}  // extern "C"
