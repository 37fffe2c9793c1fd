// spartan_amd: context, generators (window tables), fixed-base MSM, device tables.
#include <sched.h>
#include <cctype>
#include "internal.hpp"
#include <atomic>
#include <list>
#include <mutex>

const char* kProfNames[PF_COUNT] = {"gens_table_build", "msm_rows_fixed", "msm_windows_fixed", "msm_reduce_pass", "msm_reduce_compress", "eq_expand", "sumcheck_eval",
                                    "table_bind", "sumcheck_bind_eval", "vecmat", "dot", "fq_reduce", "sparse", "ipa", "spark", "misc"};

int32_t ensure(void** p, size_t* cap, size_t need) {
  if (*cap >= need) return SP_OK;
  if (*p) HIPCHK(hipFree(*p));
  *p = nullptr;
  *cap = 0;
  size_t want = need + need / 4 + 4096;
  HIPCHK(hipMalloc(p, want));
  *cap = want;
  return SP_OK;
}
static size_t pool_class(size_t bytes) {
  size_t k = 4096;
  while (k < bytes) k <<= 1;
  return k;
}
int32_t pool_alloc(sp_ctx* c, size_t bytes, void** out) {
  size_t k = pool_class(bytes);
  auto it = c->pool.find(k);
  if (it != c->pool.end() && !it->second.empty()) {
    *out = it->second.back();
    it->second.pop_back();
    return SP_OK;
  }
  hipError_t e = hipMalloc(out, k);
  if (e != hipSuccess) {  // give cached buffers back to the driver and retry once
    (void)hipStreamSynchronize(c->stream);
    for (auto& kv : c->pool) { for (void* p : kv.second) (void)hipFree(p); kv.second.clear(); }
    e = hipMalloc(out, k);
    if (e != hipSuccess) return e == hipErrorOutOfMemory ? SP_ENOMEM : SP_EHIP;
  }
  c->pool_bytes += k;
  return SP_OK;
}
void pool_release(sp_ctx* c, void* p, size_t bytes) {
  if (p) c->pool[pool_class(bytes)].push_back(p);
}
int32_t ensure_pinned(sp_ctx* c, size_t need) {
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
// A staging pair of its own for calls that queue a host vector and return without waiting (sp_vecmat_dev): the buffers of
// stage_in are rewritten by the next call, which may come before this copy has run. The event guards the pair's own reuse.
int32_t vm_stage(sp_ctx* c, const void* src, size_t bytes, const void** dev) {
  if (c->vm_ev) HIPCHK(hipEventSynchronize(c->vm_ev));  // the previous copy out of vm_pinned (long done in practice)
  else HIPCHK(hipEventCreateWithFlags(&c->vm_ev, hipEventDisableTiming));
  if (c->vm_cap < bytes) {
    HIPCHK(hipStreamSynchronize(c->stream));
    if (c->vm_pinned) HIPCHK(hipHostFree(c->vm_pinned));
    if (c->vm_dstage) HIPCHK(hipFree(c->vm_dstage));
    c->vm_pinned = c->vm_dstage = nullptr;
    c->vm_cap = 0;
    size_t want = bytes * 2 + 4096;
    HIPCHK(hipHostMalloc((void**)&c->vm_pinned, want, hipHostMallocDefault));
    HIPCHK(hipMalloc((void**)&c->vm_dstage, want));
    c->vm_cap = want;
  }
  memcpy(c->vm_pinned, src, bytes);
  HIPCHK(hipMemcpyAsync(c->vm_dstage, c->vm_pinned, bytes, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipEventRecord(c->vm_ev, c->stream));
  *dev = c->vm_dstage;
  return SP_OK;
}
void prof_drain(sp_ctx* c) {
  if (c->pending.empty()) return;
  (void)hipStreamSynchronize(c->stream);
  (void)hipStreamSynchronize(c->stream_bg);
  for (auto& r : c->pending) {
    float ms = 0;
    (void)hipEventElapsedTime(&ms, r.e0, r.e1);
    c->prof_ms[r.fam] += ms;
    c->prof_n[r.fam] += 1;
    if (r.shape) {
      ProfShape& ps = c->prof_shapes[std::make_pair(r.fam, r.shape)];
      ps.ms += ms; ps.n += 1; ps.bytes += r.bytes; ps.ops += r.ops;
      // where the launch lies on the context's clock (sp_prof_read_spans): launches of one family overlap — a background launch under a
      // foreground one, a launch queued while its predecessor still holds the CUs' LDS — so their durations do not add up to busy time
      float t0 = 0;
      if (c->prof_epoch && hipEventElapsedTime(&t0, c->prof_epoch, r.e0) == hipSuccess && c->prof_spans.size() < 65536)
      {
        unsigned long long tiles = 0;
        if (r.issued && hipMemcpy(&tiles, r.issued, 8, hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); tiles = 0; }
        c->prof_spans.push_back(ProfSpan{r.fam, r.shape, (double)t0, (double)t0 + (double)ms, 64.0 * (double)tiles});
      }
      else (void)hipGetLastError();
    }
    else if (r.bytes >= 64e6) {  // the throughput-sized launches of the other families (>= 64 MB of algorithmic traffic) as one "shape" of their own:
      ProfShape& ps = c->prof_shapes[std::make_pair(r.fam, PROF_SHAPE_BIG)];  // bench.py reports them apart from the launch-sized ones
      ps.ms += ms; ps.n += 1; ps.bytes += r.bytes; ps.ops += r.ops;
    }
    c->free_events.push_back(r.e0);
    c->free_events.push_back(r.e1);
  }
  c->pending.clear();
}
// copy small host data to the device staging buffer at byte offset off
int32_t stage_in(sp_ctx* c, size_t off, const void* src, size_t bytes) {
  SPCHK(ensure_pinned(c, off + bytes));
  memcpy(c->pinned + off, src, bytes);
  HIPCHK(hipMemcpyAsync((uint8_t*)c->dstage + off, c->pinned + off, bytes, hipMemcpyHostToDevice, c->stream));
  return SP_OK;
}
void* stage_small(sp_ctx* c, size_t off, const void* src, size_t bytes) {
  memcpy(c->hmap + off, src, bytes);
  return c->hmap + off;
}
__global__ void k_done(volatile uint32_t* flag, uint32_t seq) { SP_FG_PRIO();
  __threadfence_system();
  *flag = seq;
}
// Wait until everything queued on the main stream has run. The stream is in-order, so the flag kernel runs after the
// kernels (and copies) before it have completed, and their results in host memory precede the flag on the way to the host.
uint32_t sync_post(sp_ctx* c) {
  ahead_cancel(c);
  uint32_t seq = ++c->done_seq;
  hipLaunchKernelGGL(k_done, dim3(1), dim3(1), 0, c->stream, c->done_flag, seq);
  return seq;
}
int32_t sync_wait(sp_ctx* c, uint32_t seq) {
  for (uint64_t spins = 1;; spins++) {
    if (*c->done_flag == seq) { c->sync_epoch++; return SP_OK; }
    if ((spins & 0xFFFFF) == 0) {  // every ~ms: a faulted queue never delivers the flag
      hipError_t e = hipStreamQuery(c->stream.s);  // the raw handle: the trip after this one may be waiting for its bell behind it
      if (e == hipSuccess) {
        if (*c->done_flag != seq) return SP_EHIP;
        c->sync_epoch++;
        return SP_OK;
      }
      if (e != hipErrorNotReady) {
        fprintf(stderr, "spartan_hip: stream failed: %s\n", hipGetErrorString(e));
        return SP_EHIP;
      }
    }
  }
}
int32_t sync_spin(sp_ctx* c) { return sync_wait(c, sync_post(c)); }
DoneSig sig_make(sp_ctx* c, size_t total_workgroups) {
  ahead_cancel(c);
  if (!c->done_counter) return sig_none();
  return DoneSig{c->done_flag, c->done_counter, ++c->done_seq, (uint32_t)total_workgroups, c->ktime};
}
// wait for a trip whose last kernel was launched with `sig` (falls back to the flag kernel when the signal is off)
int32_t sig_wait(sp_ctx* c, const DoneSig& sig) { return sig.flag ? sync_wait(c, sig.seq) : sync_spin(c); }
int32_t fetch_small(sp_ctx* c, void* hdst, size_t bytes) {
  SPCHK(sync_spin(c));
  memcpy(hdst, hres(c), bytes);
  return SP_OK;
}
int32_t ensure_dstage(sp_ctx* c, size_t need) {
  if (c->dstage_cap >= need) return SP_OK;
  HIPCHK(hipStreamSynchronize(c->stream));
  return ensure(&c->dstage, &c->dstage_cap, need);
}
// device -> host through pinned memory, synchronous
int32_t fetch_out(sp_ctx* c, const void* dsrc, void* hdst, size_t bytes) {
  SPCHK(ensure_pinned(c, bytes));
  HIPCHK(hipMemcpyAsync(c->pinned, dsrc, bytes, hipMemcpyDeviceToHost, c->stream));
  SPCHK(sync_spin(c));
  memcpy(hdst, c->pinned, bytes);
  return SP_OK;
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
// stage 2: one thread per (point, window, chunk of up to 128 magnitudes): entries k * 2^(c w) * P in affine Niels form.
template <class E>  // E = Niels (one entry per 128-byte line: the gathered wide-window tables) or NielsP (packed: the streamed LDS-form tables)
__global__ void k_table_build(const Pt* __restrict__ pts, size_t n, E* __restrict__ table, MsmGeom geom) {
  // chunks of a point run over its windows in order; a wide window (msm.hpp, mixed widths) is two narrow windows' worth of chunks
  const int chunk = geom.tent < 128 ? geom.tent : 128, nchunk = geom.tent / chunk;
  const size_t per_pt = (size_t)(geom.nwin + geom.nwide) * nchunk;
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * per_pt) return;
  size_t pt = t / per_pt;
  const int slot = (int)((t % per_pt) / nchunk), n0 = msm_n0(geom);   // slot: narrow-window-sized piece of the point's table
  const int w = slot < n0 ? slot : n0 + (slot - n0) / 2;
  int ck = (int)(t % nchunk) + (slot >= n0 && ((slot - n0) & 1) ? nchunk : 0);
  Pt base = pts[pt];
  for (int k = 0; k < msm_bitpos(geom, w); k++) base = pt_dbl(base);
  // first multiple of this chunk: (ck * chunk + 1) * base by double-and-add
  int m0 = ck * chunk + 1;
  Pt acc = pt_identity();
  for (int b = 15; b >= 0; b--) {
    acc = pt_dbl(acc);
    if ((m0 >> b) & 1) acc = pt_add(acc, base);
  }
  for (int m = m0; m < m0 + chunk; m++) {
    Fp zinv = fp_invert(acc.Z);
    Niels e = pt_to_niels(acc, zinv);
    E& dst = table[msm_tidx(geom, pt, w, m)];
    dst.yp = e.yp; dst.ym = e.ym; dst.t2d = e.t2d;
    if (m + 1 < m0 + chunk) acc = pt_add(acc, base);
  }
}

// ------------------------------------------------------------------------------------------------ MSM
// thread <-> (row, strip): accumulates sum_{j in strip} Z[row][j] * P[col(j)] into one extended point.
// Lanes run fastest over rows so a wave shares the generator (and its 12 KiB window sub-table) whenever
// rows >= 64: table gathers then hit L1/L2, while the scalar load (32 B per 32 additions) is the strided one.
template <bool PF2>
__device__ __forceinline__ void msm_rows_tile(size_t lb, unsigned tid, const Fq* __restrict__ Z, size_t z_row_stride, size_t rows, size_t cols,
                                              size_t strip, size_t nstrips, const Niels* __restrict__ table, size_t g_off,
                                              const uint32_t* __restrict__ idx, const Fq* __restrict__ blinds, size_t h_idx, Pt* __restrict__ partial,
                                              int xcd_map, const MsmGeom& geom) {
  size_t row, s;
  if (xcd_map) {
    // XCD-aware tile order (block b runs on XCD b % 8, each XCD has its own L2): all row-blocks of a column strip are
    // given to the SAME XCD, back to back, so the strip's window tables are fetched into one L2 instead of eight.
    size_t rb_count = rows / 256, xcd = lb % 8, k = lb / 8;
    s = xcd + 8 * (k / rb_count);
    if (s >= nstrips) return;
    row = (k % rb_count) * 256 + tid;
  } else {
    size_t t = lb * 256 + tid;
    if (t >= rows * nstrips) return;
    row = t % rows;
    s = t / rows;
  }
  Pt acc = pt_identity();
  size_t j0 = s * strip, j1 = j0 + strip;
  if (j1 > cols) j1 = cols;
  for (size_t j = j0; j < j1; j++) {
    Fq sc = ld_fq(Z + row * z_row_stride + j);
    size_t pt = idx ? (size_t)idx[j] : g_off + j;
    msm_accumulate_t<PF2>(acc, sc, table, pt, geom);
  }
  if (blinds && s == 0) msm_accumulate_t<PF2>(acc, ld_fq(blinds + row), table, h_idx, geom);
  partial[row * nstrips + s] = acc;
}
template <bool PF2>
__global__ void __launch_bounds__(256) k_msm_rows(const Fq* __restrict__ Z, size_t z_row_stride, size_t rows, size_t cols, size_t strip,
                                                  size_t nstrips, const Niels* __restrict__ table, size_t g_off,
                                                  const uint32_t* __restrict__ idx, const Fq* __restrict__ blinds, size_t h_idx,
                                                  Pt* __restrict__ partial, int xcd_map, MsmGeom geom) {
  msm_rows_tile<PF2>(blockIdx.x, threadIdx.x, Z, z_row_stride, rows, cols, strip, nstrips, table, g_off, idx, blinds, h_idx, partial, xcd_map, geom);
}
// Background form: persistent 1024-thread workgroups, launched on fewer workgroups than the chip has CUs. At 127 VGPRs a
// CU holds exactly one of them (16 waves, 508 of 512 registers per lane), so the CUs left over cannot receive a second
// MSM workgroup and the main stream keeps a reserve of idle CUs for its latency-bound kernels — the partition a CU mask
// would give, which this platform does not honour.
__global__ void __launch_bounds__(1024) k_msm_rows_bg(const Fq* __restrict__ Z, size_t z_row_stride, size_t rows, size_t cols, size_t strip,
                                                      size_t nstrips, const Niels* __restrict__ table, size_t g_off, Pt* __restrict__ partial,
                                                      int xcd_map, size_t ntiles, MsmGeom geom) {
  extern __shared__ uint8_t occupancy_fence[];
  for (size_t lb = (size_t)blockIdx.x * 4 + threadIdx.x / 256; lb < ntiles; lb += (size_t)gridDim.x * 4)
    msm_rows_tile<false>(lb, threadIdx.x % 256, Z, z_row_stride, rows, cols, strip, nstrips, table, g_off, nullptr, nullptr, 0, partial, xcd_map, geom);
}
// ---- balanced form of the row MSM (round 4) -----------------------------------------------------------------------------------
// The strip form above gives a thread a whole number of SCALARS, and the launch a number of workgroups that has nothing to do with the
// number the chip holds (3 per CU at 164 VGPRs = 768): a 256 x 1024 witness chunk is 1024 workgroups = 1.33 waves of workgroups (the
// second one a third full: 0.61 of the addition ceiling), the derefs column half 1248 active ones = 1.6 (0.59). Here the unit of work is
// one (column, window) pair = ONE table lookup + ONE mixed addition: a row's cols x nwin units are cut into nb equal runs, run k of
// row-block r is one workgroup, and nb is chosen so that the whole launch is (at most) as many workgroups as the chip holds at once —
// every CU works from the first to the last cycle of the launch and all finish together. A run starts and ends in the middle of a
// scalar: the signed recoding's carry into its first window is rebuilt from the lower windows (integer work, once per thread).
// The digit stream runs seamlessly from one scalar into the next, so the two table entries in flight stay in flight across scalars
// (the strip form drained and refilled its pipeline once per scalar). Lanes are still rows of one column: a wave's 64 gathers of a
// step fall into one (point, window) sub-table.
// Short scalars / zero rows (SNARK::encode's address and timestamp vectors, padding rows): when no lane of the wave has a non-zero
// digit left in the current scalar (ballot), the stream jumps to the next scalar without issuing the remaining gathers.
struct MsmFlatArgs {
  const Fq* Z; size_t z_row_stride, rows, cols;
  const Niels* table; size_t g_off; const uint32_t* idx; const Fq* blinds; size_t h_idx;
  Pt* partial;          // [rows][nb]
  unsigned nb, rb_count;  // runs per row; row-blocks of 256 rows
  MsmGeom geom;
};
struct MsmUnit { const MsmEntry* p; bool neg, nz, valid; };
template <bool AHEAD>  // AHEAD: the next scalar is requested one scalar ahead (8 registers; the 128-register background form does without)
struct MsmDigitStream {
  const MsmFlatArgs& A;
  const size_t row;
  size_t j, ncol;      // current column (wave-uniform); columns incl. the blind
  long left;           // units still to hand out (wave-uniform)
  int w;               // next window of the current scalar (wave-uniform)
  uint64_t s0, s1, s2, s3;  // the current scalar, canonical, shifted down by w windows
  int carry;
  Fq raw_next;         // Montgomery form of column j + 1, requested one scalar ahead
  const MsmEntry* base;
  __device__ __forceinline__ MsmDigitStream(const MsmFlatArgs& A_, size_t row_) : A(A_), row(row_) {}
  __device__ __forceinline__ const Fq* scalar_ptr(size_t jj) const { return jj < A.cols ? A.Z + row * A.z_row_stride + jj : A.blinds + row; }
  __device__ __forceinline__ void set_base(size_t jj) {
    size_t pt = jj < A.cols ? (A.idx ? (size_t)A.idx[jj] : A.g_off + jj) : A.h_idx;
    base = reinterpret_cast<const MsmEntry*>(A.table) + pt * A.geom.pt_entries;
  }
  __device__ __forceinline__ void take(const Fq& raw) {
    Fq s = fq_from_mont(raw);  // canonical integer < q < 2^253 (scalar/mod.rs:32-36 does the same for dalek)
    s0 = s.l[0]; s1 = s.l[1]; s2 = s.l[2]; s3 = s.l[3];
    carry = 0;
  }
  __device__ __forceinline__ void shift(int c) {
    s0 = (s0 >> c) | (s1 << (64 - c));
    s1 = (s1 >> c) | (s2 << (64 - c));
    s2 = (s2 >> c) | (s3 << (64 - c));
    s3 >>= c;
  }
  __device__ __forceinline__ void open(size_t u0, size_t u1) {
    const int nwin = A.geom.nwin;
    ncol = A.cols + (A.blinds ? 1 : 0);
    left = (long)(u1 - u0);
    j = u0 / (size_t)nwin;
    w = (int)(u0 % (size_t)nwin);
    if (left <= 0) { left = 0; base = reinterpret_cast<const MsmEntry*>(A.table); s0 = s1 = s2 = s3 = 0; carry = 0; raw_next = fq_zero(); return; }
    take(ld_fq(scalar_ptr(j)));
    set_base(j);
    if (AHEAD) raw_next = j + 1 < ncol ? ld_fq(scalar_ptr(j + 1)) : fq_zero();
    for (int k = 0; k < w; k++) {  // the carry into window w depends on all lower windows
      const int c = msm_wbits_of(A.geom, k);
      int d = (int)(s0 & ((1u << c) - 1)) + carry;
      carry = d >= (1 << (c - 1));
      shift(c);
    }
  }
  __device__ __forceinline__ void next(MsmUnit& u) {
    const int nwin = A.geom.nwin;
    for (;;) {
      if (left == 0) { u.p = base; u.neg = false; u.nz = false; u.valid = false; return; }
      if (w == nwin) {
        j++;
        if (AHEAD) {
          take(raw_next);
          raw_next = j + 1 < ncol ? ld_fq(scalar_ptr(j + 1)) : fq_zero();
        } else {
          take(ld_fq(scalar_ptr(j)));
        }
        set_base(j);
        w = 0;
      }
      if (__all((s0 | s1 | s2 | s3) == 0 && carry == 0)) {  // nothing left in this scalar on any lane of the wave: no gathers for its upper windows
        long k = nwin - w;
        if (k > left) k = left;
        left -= k;
        w = nwin;
        continue;
      }
      break;
    }
    const int c = msm_wbits_of(A.geom, w);
    int d = (int)(s0 & ((1u << c) - 1)) + carry;
    carry = d >= (1 << (c - 1));
    d -= carry << c;
    uint32_t m = (uint32_t)(d < 0 ? -d : d);
    u.p = base + msm_woff(A.geom, w) + (m ? m - 1 : 0);
    u.neg = d < 0; u.nz = m != 0; u.valid = true;
    shift(c);
    w++; left--;
  }
};
// Two entries in flight, each for the time of two additions: the loop is unrolled twice so that the registers of an entry in flight are never
// the source of a copy — the compiler's waits then allow the 12 most recent loads to stay outstanding. (The rolled form with the second
// entry copied each step, and the one-entry form of a 128-register background variant, were measured in round 4 and retired in round 6.)
__device__ __forceinline__ void msm_flat_tile(const MsmFlatArgs& A, unsigned lb, unsigned tid) {
  const unsigned rb = lb % A.rb_count, bk = lb / A.rb_count;
  if (bk >= A.nb) return;
  const size_t row = (size_t)rb * 256 + tid;
  const size_t U = (A.cols + (A.blinds ? 1 : 0)) * (size_t)A.geom.nwin;
  const size_t u0 = U * bk / A.nb, u1 = U * (bk + 1) / A.nb;
  Pt acc = pt_identity();
  MsmDigitStream<true> ds(A, row);
  ds.open(u0, u1);
  MsmUnit a;
  ds.next(a);
  MsmEntry X = msm_load(a.p);
  MsmUnit b;
  ds.next(b);
  MsmEntry Y = msm_load(b.p);
#pragma unroll 1
  while (a.valid) {
    MsmEntry cur = X;
    MsmUnit ca = a;
    ds.next(a);
    X = msm_load(a.p);
    if (ca.nz) acc = pt_madd(acc, msm_entry_niels(cur), ca.neg);
    if (!b.valid) break;
    cur = Y;
    ca = b;
    ds.next(b);
    Y = msm_load(b.p);
    if (ca.nz) acc = pt_madd(acc, msm_entry_niels(cur), ca.neg);
  }
  A.partial[row * A.nb + bk] = acc;
}
__global__ void __launch_bounds__(256) k_msm_flat(MsmFlatArgs A) { msm_flat_tile(A, blockIdx.x, threadIdx.x); }

// Latency-bound shapes (Sigma-protocol commits, IPA rounds, single-row commits): one thread per (row, column,
// window) performs a single table lookup, so the serial chain per thread is one mixed addition instead of 32.
// partial[row][w*cols + j].  The blind, if any, is column `cols` (generator h_idx).
__global__ void __launch_bounds__(256) k_msm_windows(const Fq* __restrict__ Z, size_t z_row_stride, size_t rows, size_t cols,
                                                     const Niels* __restrict__ table, size_t g_off, const uint32_t* __restrict__ idx,
                                                     const Fq* __restrict__ blinds, size_t h_idx, Pt* __restrict__ partial, MsmGeom geom) { SP_FG_PRIO();
  size_t ncol = cols + (blinds ? 1 : 0);
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= rows * ncol * geom.nwin) return;
  size_t row = t % rows, rest = t / rows;
  size_t j = rest % ncol;
  int w = (int)(rest / ncol);
  Fq sc = j < cols ? ld_fq(Z + row * z_row_stride + j) : ld_fq(blinds + row);
  size_t pt = j < cols ? (idx ? (size_t)idx[j] : g_off + j) : h_idx;
  Pt acc = pt_identity();
  if (!fq_is_zero(sc)) {
    Fq s = fq_from_mont(sc);
    int d = msm_digit(s, w, geom);
    if (d != 0) acc = pt_madd(acc, table[msm_tidx(geom, pt, w, d < 0 ? -d : d)], d < 0);
  }
  partial[row * (ncol * geom.nwin) + (size_t)w * ncol + j] = acc;
}
// In-place LDS tree sum of sm[0..n) into sm[0] for a 256-thread block, with every point addition spread over FOUR lanes.
// A full extended addition is 9 dependent-ish field multiplications (~3.5 us for a lone lane); here lane `role` of a quad
// computes one of A = (Y1-X1)(Y2-X2), B = (Y1+X1)(Y2+X2), C = T1*(2d*T2), D = (2*Z1)*Z2, the quad exchanges them through
// LDS, and then one of X3 = E*F, Y3 = G*H, Z3 = F*G, T3 = E*H: three multiplications deep, same field elements (so the
// same canonical bytes). All roles run ONE instruction stream — operands are chosen by LDS address and by selects, never
// by branches — and the code is three multiplication bodies instead of nine (it is fetched cold by these few waves).
__device__ __forceinline__ Fe10 fe10_pick(const Fe10& a, const Fe10& b, bool take_b) { return fe10_select(a, b, take_b); }
// Hand-over between two tree levels that live in ONE wavefront (no block barrier): the LDS unit executes a wave's accesses in order, but
// the compiler sees no dependence between a lane's store of its sum and ANOTHER lane's load of it in the next level and may hoist that
// load; a workgroup-scope release/acquire fence pair plus a wave barrier pins the order in the instruction stream.
__device__ __forceinline__ void tree_wave_handover() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
#ifdef SP_TREE_LDS
// Round 1-3 form (A/B variant: make variant NAME=treelds FLAGS=-DSP_TREE_LDS): the quad exchanges A, B, C, D through LDS, three block
// barriers per level.
__device__ __forceinline__ void pt10_tree_quad(Pt10* sm, Fe10* xch /*[256]*/, size_t n) {
  const int t = threadIdx.x, role = t & 3;
  const Fe10 zero = Fe10{{0, 0, 0, 0, 0, 0, 0, 0, 0, 0}};
  const Fe10 k2d = fe10_pick(fe10_one(), fe10_load(fp_D2()), role == 2);  // the factor 2d belongs to C only
  // field offsets inside Pt10, in units of Fe10: X 0, Y 1, Z 2, T 3
  const int fsel = role <= 1 ? 1 : (role == 2 ? 3 : 2);  // first field:  Y, Y, T, Z
  const int gsel = role <= 1 ? 0 : (role == 2 ? 3 : 2);  // second field: X, X, -, Z
  int top = 128;
  while (top > 1 && (size_t)top >= n) top >>= 1;  // no levels that would only add identities
  for (int s = top; s > 0; s >>= 1) {
    for (int base = 0; base < s; base += 64) {
      const int q = base + (t >> 2);
      const bool act = q < s && (size_t)(q + s) < n;
      Fe10 res = zero;
      if (act) {  // whole quads are active or not
        const Fe10* P = reinterpret_cast<const Fe10*>(sm + q);
        const Fe10* Q = reinterpret_cast<const Fe10*>(sm + q + s);
        Fe10 f1 = P[fsel], g1 = P[gsel], f2 = Q[fsel], g2 = Q[gsel];
        // role 0: f - g | role 1: f + g | role 2: f (T) | role 3: P: Z + Z, Q: Z
        g1 = fe10_pick(g1, fe10_neg(g1), role == 0);
        g1 = fe10_pick(g1, zero, role == 2);
        g2 = fe10_pick(g2, fe10_neg(g2), role == 0);
        g2 = fe10_pick(g2, zero, role >= 2);
        Fe10 u = fe10_add(f1, g1), v = fe10_add(f2, g2);
        res = fe10_mul(u, fe10_mul(v, k2d));
      }
      xch[t] = res;
      __syncthreads();
      Fe10 out = zero;
      if (act) {
        const int tb = t & ~3;
        Fe10 A = xch[tb], B = xch[tb + 1], C = xch[tb + 2], D = xch[tb + 3];
        Fe10 E = fe10_sub(B, A), F = fe10_sub(D, C), G = fe10_add(D, C), H = fe10_add(B, A);
        // X3 = E*F, Y3 = G*H, Z3 = F*G, T3 = E*H
        Fe10 m1 = fe10_pick(fe10_pick(E, G, role == 1), F, role == 2);
        Fe10 m2 = fe10_pick(fe10_pick(F, H, (role & 1) != 0), G, role == 2);
        out = fe10_mul(m1, m2);
      }
      __syncthreads();  // every read of sm and xch above is done
      if (act) reinterpret_cast<Fe10*>(sm + q)[role == 0 ? 0 : (role == 1 ? 1 : (role == 2 ? 2 : 3))] = out;
      __syncthreads();
    }
  }
}
#else
// Round 4 form: the four lanes of a quad are always lanes 4k..4k+3 of ONE wavefront, so A, B, C, D never need LDS or a barrier: they
// travel by DPP quad_perm moves (v_mov_b32 ... quad_perm:[..], 20 per addition).
//   stage 1   lane 0: A = (Y1-X1)(Y2-X2)   lane 1: B = (Y1+X1)(Y2+X2)   lane 2: C = T1 (2d T2)   lane 3: D = (2 Z1) Z2
//   swap with the neighbour (quad_perm [1,0,3,2]): lane 0: E = B - A   lane 1: H = B + A   lane 2: F = D - C   lane 3: G = D + C
//   fetch the second factor (quad_perm [1,3,0,2]): lane 0: T3 = E H    lane 1: Y3 = H G     lane 2: X3 = F E     lane 3: Z3 = G F
// The sum of a pair replaces its first point in LDS, written by the quad that read it (the LDS unit executes a wave's accesses in
// order, and no other quad touches those two points in this level), so a level needs ONE block barrier — before the next level reads
// points written by other wavefronts — and the levels of at most 16 additions, which live entirely in wavefront 0, need none.
// Same field elements as the LDS form, hence the same canonical bytes. `xch` is unused (kept for the callers' LDS layout).
__device__ __forceinline__ Fe10 fe10_quad_perm_1032(const Fe10& a) {
  Fe10 r;
#pragma unroll
  for (int i = 0; i < 10; i++) r.v[i] = __builtin_amdgcn_mov_dpp(a.v[i], 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true);
  return r;
}
__device__ __forceinline__ Fe10 fe10_quad_perm_1302(const Fe10& a) {
  Fe10 r;
#pragma unroll
  for (int i = 0; i < 10; i++) r.v[i] = __builtin_amdgcn_mov_dpp(a.v[i], 0x8D /* quad_perm [1,3,0,2] */, 0xF, 0xF, true);
  return r;
}
__device__ __forceinline__ void pt10_tree_quad(Pt10* sm, Fe10* /*xch*/, size_t n) {
  const int t = threadIdx.x, role = t & 3;
  const Fe10 zero = Fe10{{0, 0, 0, 0, 0, 0, 0, 0, 0, 0}};
  const Fe10 k2d = fe10_pick(fe10_one(), fe10_load(fp_D2()), role == 2);  // the factor 2d belongs to C only
  // field offsets inside Pt10, in units of Fe10: X 0, Y 1, Z 2, T 3
  const int fsel = role <= 1 ? 1 : (role == 2 ? 3 : 2);  // first field:  Y, Y, T, Z
  const int gsel = role <= 1 ? 0 : (role == 2 ? 3 : 2);  // second field: X, X, -, Z
  const int osel = role == 0 ? 3 : (role == 1 ? 1 : (role == 2 ? 0 : 2));  // coordinate this lane produces: T, Y, X, Z
  int top = 128;
  while (top > 1 && (size_t)top >= n) top >>= 1;  // no levels that would only add identities
  for (int s = top; s > 0; s >>= 1) {
    for (int base = 0; base < s; base += 64) {
      const int q = base + (t >> 2);
      const bool act = q < s && (size_t)(q + s) < n;
      if (act) {  // whole quads are active or not
        Fe10* P = reinterpret_cast<Fe10*>(sm + q);
        const Fe10* Q = reinterpret_cast<const Fe10*>(sm + q + s);
        Fe10 f1 = P[fsel], g1 = P[gsel], f2 = Q[fsel], g2 = Q[gsel];
        // role 0: f - g | role 1: f + g | role 2: f (T) | role 3: P: Z + Z, Q: Z
        g1 = fe10_pick(g1, fe10_neg(g1), role == 0);
        g1 = fe10_pick(g1, zero, role == 2);
        g2 = fe10_pick(g2, fe10_neg(g2), role == 0);
        g2 = fe10_pick(g2, zero, role >= 2);
        Fe10 u = fe10_add(f1, g1), v = fe10_add(f2, g2);
        Fe10 mine = fe10_mul(u, fe10_mul(v, k2d));               // A | B | C | D
        Fe10 other = fe10_quad_perm_1032(mine);                    // B | A | D | C
        Fe10 val = (role & 1) ? fe10_add(mine, other) : fe10_sub(other, mine);  // E | H | F | G
        Fe10 fac = fe10_quad_perm_1302(val);                       // H | G | E | F
        P[osel] = fe10_mul(val, fac);                              // T3 = E H | Y3 = H G | X3 = F E | Z3 = G F
      }
    }
    if (s > 16) __syncthreads();  // the next level pairs points written by other wavefronts (additions q and q + s/2 sit 4 s/2 >= 64 lanes apart)
    else tree_wave_handover();    // levels inside wavefront 0: the hardware orders a wave's LDS accesses, the COMPILER is told not to reorder them
  }
  __syncthreads();  // sm[0] is the sum for every thread of the block
}
#endif
// Round 4, second form: a tree level TWO multiplications deep instead of three. The unified addition above spends its third level of
// depth on the curve constant (C = T1 * (2d * T2)); the dedicated addition of Hisil-Wong-Carter-Dawson for a = -1 ("add-2008-hwcd-4")
// has no constant at all:
//   lane 0: A = (Y1-X1)(Y2+X2)   lane 1: B = (Y1+X1)(Y2-X2)   lane 2: C = (Z1+Z1) T2   lane 3: D = (T1+T1) Z2
//   swap with the neighbour:      lane 0: F = B - A   lane 1: G = B + A   lane 2: E = D + C   lane 3: H = D - C
//   fetch the second factor (quad_perm [1,3,0,2]):  lane 0: Z3 = F G   lane 1: Y3 = G H   lane 2: X3 = E F   lane 3: T3 = H E
// Same group element as the unified formula (a different projective representative), hence the same canonical bytes. The price is
// that the formula is not complete, and the tree pays it in two places:
//   * the neutral element (zero digits, lanes past the end: exactly (0, 1, 1, 0)) is never fed to it — `idf[i]` marks the points that
//     are the untouched neutral element; a sum with one is a copy, with two stays marked;
//   * P = Q (possible only when a caller-supplied generator list repeats a point under equal scalars; sums over disjoint sets of
//     independent generators cannot collide) gives F = H = 0 in identical limbs, i.e. the all-zero quadruple, which every later
//     addition — either formula — maps to the all-zero quadruple again: the ROOT then has Z = 0, which no valid point has, and the
//     host re-runs the launch with the unified tree (ipa.hip). P = -Q gives (0, Y, Z, 0) with Y = Z, a valid neutral element.
__device__ __forceinline__ void pt10_tree_quad_ded(Pt10* sm, unsigned char* idf, size_t n) {
  const int t = threadIdx.x, role = t & 3;
  const Fe10 zero = Fe10{{0, 0, 0, 0, 0, 0, 0, 0, 0, 0}};
  // field offsets inside Pt10, in units of Fe10: X 0, Y 1, Z 2, T 3
  const int fP = role <= 1 ? 1 : role;                    // P: Y, Y, Z, T
  const int gP = role <= 1 ? 0 : role;                    //    X (minus for lane 0), X, Z, T
  const int fQ = role <= 1 ? 1 : (role == 2 ? 3 : 2);     // Q: Y, Y, T, Z
  const int osel = role == 0 ? 2 : (role == 1 ? 1 : (role == 2 ? 0 : 3));  // coordinate this lane produces: Z, Y, X, T
  int top = 128;
  while (top > 1 && (size_t)top >= n) top >>= 1;
  for (int s = top; s > 0; s >>= 1) {
    for (int base = 0; base < s; base += 64) {
      const int q = base + (t >> 2);
      const bool act = q < s && (size_t)(q + s) < n;
      if (act) {  // whole quads are active or not, and take the same branch below
        Fe10* P = reinterpret_cast<Fe10*>(sm + q);
        const Fe10* Q = reinterpret_cast<const Fe10*>(sm + q + s);
        const bool idP = idf[q] != 0, idQ = idf[q + s] != 0;
        if (!idQ) {
          if (idP) {
            P[role] = Q[role];
            if (role == 0) idf[q] = 0;
          } else {
            Fe10 f1 = P[fP], g1 = P[gP], f2 = Q[fQ], g2 = Q[0];
            g1 = fe10_pick(g1, fe10_neg(g1), role == 0);
            g2 = fe10_pick(fe10_pick(g2, fe10_neg(g2), role == 1), zero, role >= 2);
            Fe10 mine = fe10_mul(fe10_add(f1, g1), fe10_add(f2, g2));      // A | B | C | D
            Fe10 other = fe10_quad_perm_1032(mine);                         // B | A | D | C
            Fe10 minuend = fe10_pick(mine, other, role == 0), subtrahend = fe10_pick(other, mine, role == 0);
            Fe10 val = (role == 1 || role == 2) ? fe10_add(mine, other) : fe10_sub(minuend, subtrahend);  // F | G | E | H
            Fe10 fac = fe10_quad_perm_1302(val);                            // G | H | F | E
            P[osel] = fe10_mul(val, fac);                                   // Z3 = F G | Y3 = G H | X3 = E F | T3 = H E
          }
        }
      }
    }
    if (s > 16) __syncthreads();
    else tree_wave_handover();
  }
  __syncthreads();
}
// the untouched neutral element as the kernels build it: (0, 1, 1, 0) limb for limb (a computed neutral element has other limbs and is an
// ordinary operand)
__device__ __forceinline__ bool pt10_is_blank(const Pt10& p) {
  int32_t acc = 0;
#pragma unroll
  for (int i = 0; i < 10; i++) acc |= p.X.v[i] | p.T.v[i];
#pragma unroll
  for (int i = 1; i < 10; i++) acc |= p.Y.v[i] | p.Z.v[i];
  return acc == 0 && p.Y.v[0] == 1 && p.Z.v[0] == 1;
}

// Latency path, fused form (rows <= SP_HOST_ENCODE_ROWS): grid (nblk, rows). Thread p of a row looks up the table entry
// of its (column, window) pair and the block sums its 256 entries in an LDS tree, so a Sigma-protocol commitment is ONE
// launch (nblk = 1, FINAL: the sum goes out as an extended point for the host to encode) and an inner-product round is two
// (Pt10 partials, then k_msm_reduce) instead of three. An affine table entry (y+x, y-x, 2dxy) becomes the extended point
// (2(yp-ym), 2(yp+ym), 4, (yp-ym)(yp+ym)) = 4*(x, y, 1, xy) with one multiplication.
template <bool FINAL>
__global__ void __launch_bounds__(256) k_msm_windows_tree(const Fq* __restrict__ Z, size_t z_row_stride, size_t cols,
                                                          const Niels* __restrict__ table, size_t g_off, const uint32_t* __restrict__ idx,
                                                          size_t idx_row_stride, const Fq* __restrict__ blinds, size_t h_idx,
                                                          void* __restrict__ out, MsmGeom geom) { SP_FG_PRIO();
  __shared__ Pt10 sm[256];
  __shared__ Fe10 xch[256];
  size_t ncol = cols + (blinds ? 1 : 0), P = ncol * geom.nwin, row = blockIdx.y;
  int t = threadIdx.x;
  size_t p = (size_t)blockIdx.x * 256 + t;
  Pt10 acc = pt10_identity();
  if (p < P) {
    size_t j = p % ncol;
    int w = (int)(p / ncol);
    Fq sc = j < cols ? ld_fq(Z + row * z_row_stride + j) : ld_fq(blinds + row);
    size_t pt = j < cols ? (idx ? (size_t)idx[row * idx_row_stride + j] : g_off + j) : h_idx;
    if (!fq_is_zero(sc)) {
      int d = msm_digit(fq_from_mont(sc), w, geom);
      if (d != 0) {
        Niels n = table[msm_tidx(geom, pt, w, d < 0 ? -d : d)];
        Fp dx = fp_sub(n.yp, n.ym), sy = fp_add(n.yp, n.ym);
        Fp X = fp_add(dx, dx), T = fp_mul(dx, sy);
        if (d < 0) { X = fp_neg(X); T = fp_neg(T); }
        acc = Pt10{fe10_load(X), fe10_load(fp_add(sy, sy)), Fe10{{4, 0, 0, 0, 0, 0, 0, 0, 0, 0}}, fe10_load(T)};
      }
    }
  }
  sm[t] = acc;
  __syncthreads();
  size_t live = P - (size_t)blockIdx.x * 256;  // partial indices of this block that exist
  pt10_tree_quad(sm, xch, live);
  if (t == 0) {
    Pt10 r = sm[0];
    if (FINAL) ((Pt*)out)[row] = Pt{fe10_to_fp(r.X), fe10_to_fp(r.Y), fe10_to_fp(r.Z), fe10_to_fp(r.T)};
    else ((Pt10*)out)[row * gridDim.x + blockIdx.x] = r;
  }
}
// The same lookups + tree for commitments of more than 256 (column, window) pairs per row (rows <= 8), in ONE launch: every
// workgroup leaves its partial sum in device memory and takes a ticket at its row's counter; the workgroup that draws the last
// ticket of a row adds the row's partial sums in a second tree and writes the row sum to the host-mapped page; the last
// workgroup of the launch raises the completion flag. (Before: lookups + tree, reduction and flag as three launches.)
__global__ void __launch_bounds__(256) k_msm_windows_tree_fused(const Fq* __restrict__ Z, size_t z_row_stride, size_t cols, const Niels* __restrict__ table,
                                                                size_t g_off, const uint32_t* __restrict__ idx, size_t idx_row_stride,
                                                                const Fq* __restrict__ blinds, size_t h_idx, Pt10* __restrict__ part, uint32_t* __restrict__ tickets,
                                                                Pt* __restrict__ sums_out, MsmGeom geom, DoneSig sig) { SP_FG_PRIO();
  __shared__ Pt10 sm[256];
  __shared__ Fe10 xch[256];
  __shared__ unsigned ticket;
  size_t ncol = cols + (blinds ? 1 : 0), P = ncol * geom.nwin, row = blockIdx.y;
  int t = threadIdx.x;
  size_t p = (size_t)blockIdx.x * 256 + t;
  Pt10 acc = pt10_identity();
  if (p < P) {
    size_t j = p % ncol;
    int w = (int)(p / ncol);
    Fq sc = j < cols ? ld_fq(Z + row * z_row_stride + j) : ld_fq(blinds + row);
    size_t pt = j < cols ? (idx ? (size_t)idx[row * idx_row_stride + j] : g_off + j) : h_idx;
    if (!fq_is_zero(sc)) {
      int d = msm_digit(fq_from_mont(sc), w, geom);
      if (d != 0) {
        Niels n = table[msm_tidx(geom, pt, w, d < 0 ? -d : d)];
        Fp dx = fp_sub(n.yp, n.ym), sy = fp_add(n.yp, n.ym);
        Fp X = fp_add(dx, dx), T = fp_mul(dx, sy);
        if (d < 0) { X = fp_neg(X); T = fp_neg(T); }
        acc = Pt10{fe10_load(X), fe10_load(fp_add(sy, sy)), Fe10{{4, 0, 0, 0, 0, 0, 0, 0, 0, 0}}, fe10_load(T)};
      }
    }
  }
  sm[t] = acc;
  __syncthreads();
  pt10_tree_quad(sm, xch, P - (size_t)blockIdx.x * 256);
  const unsigned nblk = gridDim.x;
  if (t == 0) {
    part[row * nblk + blockIdx.x] = sm[0];
    __threadfence();
    ticket = atomicAdd(tickets + row, 1u);
  }
  __syncthreads();
  if (ticket == nblk - 1) {
    __threadfence();
    Pt10 r = pt10_identity();
    bool any = false;
    for (size_t k2 = t; k2 < nblk; k2 += 256) {
      Pt10 q2 = part[row * nblk + k2];
      r = any ? pt10_add(r, q2) : q2;
      any = true;
    }
    __syncthreads();
    sm[t] = r;
    __syncthreads();
    pt10_tree_quad(sm, xch, nblk < 256 ? nblk : 256);
    if (t == 0) {
      Pt10 z = sm[0];
      sums_out[row] = Pt{fe10_to_fp(z.X), fe10_to_fp(z.Y), fe10_to_fp(z.Z), fe10_to_fp(z.T)};
      tickets[row] = 0;
    }
  }
  signal_done(sig);
}
// ---- one inner-product round in ONE launch (BulletReductionProof::prove, src/nizk/bullet.rs:72-100) ---------------------
// L = <a_L, G_R>, R = <a_R, G_L> over the ORIGINAL generators (spartan_hip.h: the folded generators are never built): row 0
// pairs scalar a'[i] s'[p] with generator p n_cur + h + i, row 1 scalar a'[h + i] s'[p] with generator p n_cur + i, for
// p < n0 / n_cur, i < h = n_cur / 2. The c_L Q + blind_L H part of each row (two terms) is added by the calling thread from
// its host-side window tables while this kernel runs (ipa.hip), so every column here is a plain generator column.
// Blocks [0, 2 nblk): 256 (column, window) lookups of one row each, summed in the LDS tree; the block that arrives last at
// its row's counter adds the row's nblk partial sums and writes the row sum to the host-mapped page — what used to be three
// launches (scalar rows, lookups + tree, reduction) and a flag kernel. With `fold` set the vectors are those of the previous
// round and the fold of bullet.rs:105-109 by (u, u^-1) is applied on the way (a' = a_L u + u^-1 a_R, s'[2p] = s[p] u^-1,
// s'[2p+1] = s[p] u: 3 multiplications per lookup, redundantly per window, instead of a launch of their own).
// Blocks [2 nblk, 2 nblk + nd) work off the critical path: they materialise a', b', s' for the next round and form the eight
// quarter dot products from which the NEXT round's c_L, c_R follow once its challenge is known:
//   c_L'' = <a'_LL,b'_LR> + u^2 <a'_LL,b'_RR> + u^-2 <a'_RL,b'_LR> + <a'_RL,b'_RR>
//   c_R'' = <a'_LR,b'_LL> + u^2 <a'_LR,b'_RL> + u^-2 <a'_RR,b'_LL> + <a'_RR,b'_RL>        (quarters of a', b' in index order LL, LR, RL, RR)
// so no round waits for a dot product over the whole vectors (bullet.rs:80-81).
struct IpaRoundArgs {
  const Fq *a, *b, *s;         // vectors before the pending fold (a, b: 2 n_cur entries when fold; s: n0 / (2 n_cur) entries), else current
  Fq *a_new, *b_new, *s_new;   // the vectors of this round, written when fold != 0
  size_t n_cur, n0, g_off;
  int fold;
  Fq u, u_inv;
  Fq u2, u2_inv;               // u^2, u^-2: the lookups' scalar a'[i] s'[p] in two multiplications (k_ipa_round)
  Pt10* part;                  // [2][nblk]
  uint32_t* counters;          // [2], zero between launches
  Pt* sums_out;                // host page: the two row sums
  Fq* dots_out;                // host page: [nd][8] partial quarter dot products
  unsigned nblk, nd;
};
__device__ __forceinline__ Fq ipa_fold_a(const Fq* __restrict__ a, size_t x, size_t n_cur, int fold, const Fq& u, const Fq& u_inv) {
  Fq v = ld_fq(a + x);
  return fold ? fq_add(fq_mul(v, u), fq_mul(u_inv, ld_fq(a + n_cur + x))) : v;  // a' = a_L u + u^-1 a_R
}
__device__ __forceinline__ Fq ipa_fold_b(const Fq* __restrict__ b, size_t x, size_t n_cur, int fold, const Fq& u, const Fq& u_inv) {
  Fq v = ld_fq(b + x);
  return fold ? fq_add(fq_mul(v, u_inv), fq_mul(u, ld_fq(b + n_cur + x))) : v;  // b' = b_L u^-1 + u b_R
}
template <bool DED>  // DED: the two-multiplication tree level (pt10_tree_quad_ded); false: the unified formula (the fallback of an exceptional sum)
__global__ void __launch_bounds__(256) k_ipa_round(IpaRoundArgs A, const Niels* __restrict__ table, MsmGeom geom, DoneSig sig) { SP_FG_PRIO();
  __shared__ Pt10 sm[256];
  __shared__ Fe10 xch[256];
  __shared__ unsigned ticket;
  unsigned char* const idf = reinterpret_cast<unsigned char*>(xch);  // DED: one mark per point (the unified tree's exchange buffer is unused then)
  const int t = threadIdx.x;
  const size_t h = A.n_cur / 2;
  if (blockIdx.x >= 2 * A.nblk) {
    // ---- the next round's vectors and quarter dot products
    Fq* const red = reinterpret_cast<Fq*>(sm);  // [8][64]
    const size_t qlen = A.n_cur >= 4 ? A.n_cur / 4 : 1;
    const unsigned db = blockIdx.x - 2 * A.nblk;
    const size_t z = (size_t)db * 64 + (t >> 2);
    const int k = t & 3;
    const size_t x = z + (size_t)k * qlen;
    const bool live = z < qlen && x < A.n_cur;
    Fq av = fq_zero(), bv = fq_zero();
    if (live) {
      av = ipa_fold_a(A.a, x, A.n_cur, A.fold, A.u, A.u_inv);
      bv = ipa_fold_b(A.b, x, A.n_cur, A.fold, A.u, A.u_inv);
      if (A.fold) { st_fq(A.a_new + x, av); st_fq(A.b_new + x, bv); }
      // the last round (two entries left): a', b' also go to the host page, behind the dot-product slots — with them, the round's two raw row
      // sums and the last challenge the calling thread finishes the argument itself (ipa.hip, sp_ipa_finish_commit)
      if (A.n_cur == 2) { st_fq(A.dots_out + 8 + x, av); st_fq(A.dots_out + 10 + x, bv); }
    }
    if (A.fold)
      for (size_t p = (size_t)db * 256 + t; p < A.n0 / A.n_cur; p += (size_t)A.nd * 256)
        st_fq(A.s_new + p, fq_mul(ld_fq(A.s + p / 2), (p & 1) ? A.u : A.u_inv));  // s'[2p] = s[p] u^-1, s'[2p+1] = s[p] u
    // lane k of an index multiplies its a' with two of the b' of the index: exchange the b' through LDS
    Fq* const bx = reinterpret_cast<Fq*>(xch);  // [64][4]
    bx[(t >> 2) * 4 + k] = bv;
    __syncthreads();
    Fq e0 = fq_zero(), e1 = fq_zero();
    if (live && A.n_cur >= 4) {
      e0 = fq_mul(av, bx[(t >> 2) * 4 + ((k & 1) ? 0 : 1)]);
      e1 = fq_mul(av, bx[(t >> 2) * 4 + ((k & 1) ? 2 : 3)]);
    }
    // dot d = 2 * slot(k) + {0, 1} with slot: k = 0 (a_LL) -> d0, d1 | k = 2 (a_RL) -> d2, d3 | k = 1 (a_LR) -> d4, d5 | k = 3 (a_RR) -> d6, d7
    const int slot = k == 0 ? 0 : (k == 2 ? 1 : (k == 1 ? 2 : 3));
    __syncthreads();
    red[(2 * slot) * 64 + (t >> 2)] = e0;
    red[(2 * slot + 1) * 64 + (t >> 2)] = e1;
    __syncthreads();
    for (int st = 32; st > 0; st >>= 1) {
      for (int v = t; v < 8 * st; v += 256) {
        const int d = v / st, i = v % st;
        red[d * 64 + i] = fq_add(red[d * 64 + i], red[d * 64 + i + st]);
      }
      __syncthreads();
    }
    if (t < 8) st_fq(A.dots_out + (size_t)db * 8 + t, red[t * 64]);
    signal_done(sig);
    return;
  }
  // ---- lookups of one row + tree
  const unsigned row = blockIdx.x / A.nblk, blk = blockIdx.x % A.nblk;
  DoneSig kt0 = sig; if (blockIdx.x != 0) kt0.kt = nullptr;   // stamps 0..5: the first workgroup
  SP_KT(kt0, 0);
  const size_t cols = A.n0 / 2, P = cols * (size_t)geom.nwin;
  const size_t p = (size_t)blk * 256 + t;
  Pt10 acc = pt10_identity();
  if (p < P) {
    const size_t q = p % cols;
    const int w = (int)(p / cols);
    const size_t pb = q / h, i = q % h;
    const size_t gen = A.g_off + pb * A.n_cur + (row == 0 ? h + i : i);
    // a'[x] s'[p] with the pending fold applied: (a_L u + a_R u^-1) * s[p/2] u^(+-1) = s[p/2] (a_L u^2 + a_R) for odd p, s[p/2] (a_L + a_R u^-2) for
    // even p — two multiplications per lookup instead of four (the same field element)
    const size_t x = row == 0 ? i : h + i;
    Fq sc;
    if (A.fold) {
      const Fq aL = ld_fq(A.a + x), aR = ld_fq(A.a + A.n_cur + x);
      const Fq t = (pb & 1) ? fq_add(fq_mul(aL, A.u2), aR) : fq_add(aL, fq_mul(aR, A.u2_inv));
      sc = fq_mul(ld_fq(A.s + pb / 2), t);
    } else {
      sc = fq_mul(ld_fq(A.a + x), ld_fq(A.s + pb));
    }
    SP_KT(kt0, 1);
    if (!fq_is_zero(sc)) {
      int d = msm_digit(fq_from_mont(sc), w, geom);
      if (d != 0) {
        Niels n = table[msm_tidx(geom, gen, w, d < 0 ? -d : d)];
        SP_KT(kt0, 2);
        Fp dx = fp_sub(n.yp, n.ym), sy = fp_add(n.yp, n.ym);
        Fp X = fp_add(dx, dx), T = fp_mul(dx, sy);
        if (d < 0) { X = fp_neg(X); T = fp_neg(T); }
        acc = Pt10{fe10_load(X), fe10_load(fp_add(sy, sy)), Fe10{{4, 0, 0, 0, 0, 0, 0, 0, 0, 0}}, fe10_load(T)};
      }
    }
  }
  sm[t] = acc;
  if (DED) idf[t] = pt10_is_blank(acc) ? 1 : 0;
  SP_KT(kt0, 3);
  __syncthreads();
  if (DED) pt10_tree_quad_ded(sm, idf, P - (size_t)blk * 256);
  else pt10_tree_quad(sm, xch, P - (size_t)blk * 256);
  SP_KT(kt0, 4);
  if (t == 0) {
    A.part[(size_t)row * A.nblk + blk] = sm[0];
    __threadfence();  // the partial sum is visible device-wide before this block counts itself in
    ticket = atomicAdd(A.counters + row, 1u);
  }
  __syncthreads();
  SP_KT(kt0, 5);
  if (ticket == A.nblk - 1) {  // last block of the row: add the row's partial sums
    DoneSig kt1 = sig; if (row != 0) kt1.kt = nullptr;   // stamps 8..11: the reducing workgroup of row 0
    SP_KT(kt1, 8);
    __threadfence();
    Pt10 r = pt10_identity();
    bool any = false;
    for (size_t k2 = t; k2 < A.nblk; k2 += 256) {
      Pt10 q2 = A.part[(size_t)row * A.nblk + k2];
      r = any ? pt10_add(r, q2) : q2;
      any = true;
    }
    __syncthreads();
    sm[t] = r;
    if (DED) idf[t] = pt10_is_blank(r) ? 1 : 0;
    SP_KT(kt1, 9);
    __syncthreads();
    if (DED) pt10_tree_quad_ded(sm, idf, A.nblk < 256 ? A.nblk : 256);
    else pt10_tree_quad(sm, xch, A.nblk < 256 ? A.nblk : 256);
    SP_KT(kt1, 10);
    if (t == 0) {
      Pt10 z = sm[0];
      A.sums_out[row] = Pt{fe10_to_fp(z.X), fe10_to_fp(z.Y), fe10_to_fp(z.Z), fe10_to_fp(z.T)};
      A.counters[row] = 0;
    }
    SP_KT(kt1, 11);
  }
  signal_done(sig);
}
// unified: the tree with the complete addition formula (the re-run of a round whose row sum came back with Z = 0; SPARTAN_IPA_UNIFIED_TREE=1 selects it
// for every round: the A/B switch)
extern "C" int32_t ipa_round_launch(sp_ctx* c, const sp_gens* g, IpaRoundArgs* A, DoneSig* sig_out, int unified) {
  const size_t P = (A->n0 / 2) * (size_t)g->geom.nwin;
  A->nblk = (unsigned)((P + 255) / 256);
  const size_t qlen = A->n_cur >= 4 ? A->n_cur / 4 : 1;
  A->nd = (unsigned)((qlen + 63) / 64);
  if (1024 + ((size_t)A->nd * 8 + 4) * 32 > HOST_SUM_BYTES) return SP_EINVAL;  // the dot-product partials share the host page with the two row sums (n_cur <= 16384)
  SPCHK(ensure(&c->scratch, &c->scratch_cap, sizeof(Pt10) * 2 * (size_t)A->nblk + 256));
  A->part = (Pt10*)c->scratch;
  A->sums_out = (Pt*)hres(c);
  A->dots_out = (Fq*)(hres(c) + 1024);
  DoneSig sig = sig_make(c, 2 * (size_t)A->nblk + A->nd);
  {
    ProfScope ps(c, PF_IPA, 32.0 * 3 * (double)A->n0 + 160.0 * 2 * (double)A->nblk, nullptr, (double)(2 * P));
    // The dedicated (two-multiplication) tree is incomplete, and its exceptional pairs are detected at the root only for P = Q and
    // P = Q + (0, -1) (an all-zero quadruple propagates to Z = 0); for Q = P +- (i, 0) the formula gives Y3 = Z3 = 0 with X3 != 0, which a
    // later addition can turn back into Z != 0 — undetected. Such pairs need a known relation between partial
    // sums over disjoint generators: impossible for points the library derived itself by hash-to-curve (MultiCommitGens::new), possible in
    // principle for a caller-supplied list (sp_gens_upload). Caller-supplied sets therefore always get the complete (unified) tree.
    const bool always_unified = c->opt.v[OPT_IPA_UNIFIED_TREE] != 0 || (!g->derived && !c->opt.v[OPT_IPA_DEDICATED_UPLOADED]);
    if (unified || always_unified) hipLaunchKernelGGL(k_ipa_round<false>, dim3(2 * A->nblk + A->nd), dim3(256), 0, c->stream, *A, (const Niels*)g->table, g->geom, sig);
    else hipLaunchKernelGGL(k_ipa_round<true>, dim3(2 * A->nblk + A->nd), dim3(256), 0, c->stream, *A, (const Niels*)g->table, g->geom, sig);
  }
  *sig_out = sig;
  return hipGetLastError() == hipSuccess ? SP_OK : SP_EHIP;
}

// reduction pass: grid (rows, nchunks); block sums `chunk` consecutive partials of its row into one point.
// Reductions run on the radix-2^25.5 serial-chain arithmetic (fe10.hpp): few waves, latency-bound.
__global__ void __launch_bounds__(256) k_pt_reduce_pass(const Pt* __restrict__ in, size_t P, size_t chunk, Pt10* __restrict__ out,
                                                        const unsigned* __restrict__ counts /* queue form: slots in use per 64-row group, or null */) { SP_FG_PRIO();
  __shared__ Pt10 sm[256];
  __shared__ Fe10 xch[256];
  size_t row = blockIdx.x, ck = blockIdx.y, nchunks = gridDim.y;
  int t = threadIdx.x;
  size_t lo = ck * chunk, hi = lo + chunk;
  if (hi > P) hi = P;
  if (counts && hi > counts[row / 64]) hi = counts[row / 64];
  Pt10 acc = pt10_identity();
  bool any = false;
  for (size_t s = lo + t; s < hi; s += 256) {
    Pt10 p = pt10_load(in[row * P + s]);
    acc = any ? pt10_add(acc, p) : p;
    any = true;
  }
  sm[t] = acc;
  __syncthreads();
  pt10_tree_quad(sm, xch, 256);
  if (t == 0) out[row * nchunks + ck] = sm[0];
}
// one block per row: sum the row's partials, RFC 9496 encode. IN10: partials already in Pt10 form (second pass).
// ENCODE = false: the sum is written as an extended point (4 x 32 B, canonical coordinates) and the caller encodes it on
// the host. The encode is ONE serial chain of ~265 field multiplications (an inverse square root): ~100 us for a lone
// wavefront at one instruction per ~5 cycles, ~3 us for a CPU core with a 64-bit multiplier. Commitments of a few rows
// (every Sigma-protocol and inner-product round: 151 sequential ones per 2^20 proof) take the host route, batched row
// commitments (hundreds of rows in parallel) the device route.
template <bool IN10, bool ENCODE>
__global__ void __launch_bounds__(256) k_msm_reduce(const void* __restrict__ partial_, size_t nstrips, uint8_t* __restrict__ out,
                                                    const unsigned* __restrict__ counts /* queue form: slots in use per 64-row group, or null */) { SP_FG_PRIO();
  __shared__ Pt10 sm[256];
  __shared__ Fe10 xch[256];
  size_t row = blockIdx.x;
  int t = threadIdx.x;
  Pt10 acc = pt10_identity();
  bool any = false;
  size_t nlive = nstrips;  // the row's partial sums lie nstrips apart; the first nlive of them hold a sum
  if (counts && !IN10 && counts[row / 64] < nstrips) nlive = counts[row / 64];
  for (size_t s = t; s < nlive; s += 256) {
    Pt10 p = IN10 ? ((const Pt10*)partial_)[row * nstrips + s] : pt10_load(((const Pt*)partial_)[row * nstrips + s]);
    acc = any ? pt10_add(acc, p) : p;
    any = true;
  }
  sm[t] = acc;
  __syncthreads();
  pt10_tree_quad(sm, xch, nlive < 256 ? nlive : 256);
  if (t == 0) {
    Pt10 r = sm[0];
    if (ENCODE) {
      uint8_t c[32];
      fe10_pin(r.X); fe10_pin(r.Y); fe10_pin(r.Z); fe10_pin(r.T);
      pt10_compress(r, c);
      for (int k = 0; k < 32; k++) out[32 * row + k] = c[k];
    } else {
      ((Pt*)out)[row] = Pt{fe10_to_fp(r.X), fe10_to_fp(r.Y), fe10_to_fp(r.Z), fe10_to_fp(r.T)};
    }
  }
}


// RFC 9496 encode of many row sums at once: one LANE per row. (With the encode inside k_msm_reduce one lane per BLOCK runs
// the ~100 us inverse-square-root chain while 255 wait: a 1024-row commit spent 0.6 ms there; this way the 1024 chains
// run side by side in 16 wavefronts.)
__device__ __forceinline__ void pt_encode_rows(const Pt* __restrict__ in, size_t rows, uint8_t* __restrict__ out) {
  size_t row = (size_t)blockIdx.x * 64 + threadIdx.x;
  if (row >= rows) return;
  uint8_t c[32];
  pt10_compress(pt10_load(in[row]), c);
  for (int k = 0; k < 32; k++) out[32 * row + k] = c[k];
}
__global__ void __launch_bounds__(64) k_pt_encode(const Pt* __restrict__ in, size_t rows, uint8_t* __restrict__ out) { SP_FG_PRIO();
  pt_encode_rows(in, rows, out);
}
// The same in at most 168 VGPRs (the compiler takes all 256 for its schedule of the chain above; here it spills ~100 dwords): the encode of
// the ROW half of `derefs` runs while the column half's queue-form MSM holds two wavefronts per SIMD on every CU — 272 of a lane's 512
// registers — and a 256-register wavefront could not be placed until that launch had ended (first round-6 traces: k_pt_encode 0.13 -> 3 ms
// on average at 2^22, the row half's shares absorbed after the MSM instead of under it).
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 3))) k_pt_encode_lean(const Pt* __restrict__ in, size_t rows, uint8_t* __restrict__ out) { SP_FG_PRIO();
  pt_encode_rows(in, rows, out);
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

static int32_t ctx_init(sp_ctx* c, int device_id);
int32_t sp_ctx_create(int device_id, sp_ctx** out) {
  if (!out) return SP_EINVAL;
  *out = nullptr;
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
  int32_t rc = ctx_init(c, device_id);
  if (rc != SP_OK) {  // release whatever was created before the failing step
    sp_ctx_destroy(c);
    return rc;
  }
  *out = c;
  return SP_OK;
}
// The proving thread and the device exchange ~330 small messages per proof over PCIe (launches, the completion flag, challenges): with the thread on
// the other socket every one of them crosses the inter-socket link as well — 22.2-22.8 ms per 2^20 proof from the GPU's own node against 22.7-23.4 from
// the other one or unpinned on one box of the pool, no difference on another (profiles/r6_ab_numa.txt): part of the "box-to-box" spread of the earlier rounds. Narrow the CALLING thread's affinity (threads it
// creates later inherit it; the pinned host pages allocated below are then first touched on that node) to the CPUs sysfs lists as local to the
// device's PCI function. Never widens a mask, does nothing when the lists do not intersect or sysfs has no answer.
static void pin_thread_to_device_node(int device_id) {
  char bus[32] = {0};
  if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, device_id) != hipSuccess) { (void)hipGetLastError(); return; }
  for (char* p = bus; *p; p++) *p = (char)tolower((unsigned char)*p);
  char path[128];
  snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/local_cpulist", bus);
  FILE* f = fopen(path, "r");
  if (!f) return;
  char list[1024] = {0};
  const bool got = fgets(list, sizeof list, f) != nullptr;
  fclose(f);
  if (!got) return;
  cpu_set_t have, want;
  if (sched_getaffinity(0, sizeof have, &have) != 0) return;
  CPU_ZERO(&want);
  int n = 0;
  for (char* p = list; *p && *p != '\n';) {  // "64-127,192-255"
    char* end = nullptr;
    long lo = strtol(p, &end, 10), hi = lo;
    if (end == p) break;
    if (*end == '-') { p = end + 1; hi = strtol(p, &end, 10); if (end == p) break; }
    for (long k = lo; k <= hi && k < CPU_SETSIZE; k++)
      if (k >= 0 && CPU_ISSET((int)k, &have)) { CPU_SET((int)k, &want); n++; }
    p = *end == ',' ? end + 1 : end;
  }
  if (n >= 8 && n < CPU_COUNT(&have)) (void)sched_setaffinity(0, sizeof want, &want);  // a handful of CPUs would starve the threads this one creates (uploader, small-commitment worker)
}
static int32_t ctx_init(sp_ctx* c, int device_id) {
  c->dev = device_id;
  c->stream = c->stream_bg = c->stream_side = nullptr;
  c->sync_ev = c->side_ev = nullptr;
  c->scratch = c->scratch2 = c->dstage = nullptr;
  c->scratch_cap = c->scratch2_cap = c->dstage_cap = 0;
  c->pinned = nullptr;
  c->pinned_cap = 0;
  c->hmap = nullptr;
  c->done_flag = nullptr;
  c->sync_epoch = 0;
  c->eq_next = 0;
  for (int k = 0; k < 8; k++) c->eq_slot_epoch[k] = 0;
  c->opt = sp_default_options();
  if (c->opt.v[OPT_HOST_PIN_THREAD]) pin_thread_to_device_node(device_id);  // before the pinned host pages below are allocated
  c->device_encode = c->opt.v[OPT_ENCODE_DEVICE] != 0;  // diagnostic: keep every RFC 9496 encode on the GPU
  c->prof_on = 0;
  c->prof_mask = ~0ULL;
  c->pool_bytes = 0;
  memset(c->prof_ms, 0, sizeof c->prof_ms);
  memset(c->prof_n, 0, sizeof c->prof_n);
  memset(c->prof_bytes, 0, sizeof c->prof_bytes);
  memset(c->prof_ops, 0, sizeof c->prof_ops);
  {
    // main stream: the Fiat-Shamir critical path, highest priority; background stream: throughput MSMs queued under it,
    // lowest priority. (CU masks would be the cleaner partition, but hipExtStreamCreateWithCUMask is not honoured on this
    // platform: it succeeds, and a 1/8 mask runs an MSM exactly as fast as 8/8 — bench/bg_probe.py.)
    int lo = 0, hi = 0;
    HIPCHK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    HIPCHK(hipStreamCreateWithPriority(&c->stream.s, hipStreamNonBlocking, hi));
    c->stream.owner = c;
    HIPCHK(hipStreamCreateWithPriority(&c->stream_side, hipStreamNonBlocking, hi));
    HIPCHK(hipEventCreateWithFlags(&c->side_ev, hipEventDisableTiming));
    HIPCHK(hipStreamCreateWithPriority(&c->stream_bg, hipStreamNonBlocking, lo));
    HIPCHK(hipStreamCreateWithPriority(&c->stream_low, hipStreamNonBlocking, lo));
    // background MSMs: one 1024-thread workgroup per CU on half of the CUs (k_msm_rows_bg). Measured at 2^20 with the
    // derefs row half in the background, share in eighths 2 / 3 / 4 / 5 / 6 / 8 -> 63.1 / 59.2 / 58.0 / 59.0 / 61.5 / 62.5 ms
    // per proof (63.3 without the overlap): less and the MSM is not done when it is needed, more and the second
    // sum-check's kernels queue behind MSM workgroups. Round 3 (the foreground under the MSM got shorter: look-ahead, fused inner-product
    // rounds), share 3 / 4 / 5 / 6 -> 29.2 / 28.3 / 27.5 / 27.8 ms per proof (29.98 without the overlap): 5.
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device_id));
    c->n_cus = prop.multiProcessorCount;
    c->bg_lds = 0;
    c->bg_blocks = c->n_cus * (int)c->opt.v[OPT_BG_EIGHTHS] / 8;
  }
  HIPCHK(hipHostMalloc((void**)&c->hmap, HMAP_SIZE, hipHostMallocDefault));
  HIPCHK(hipHostMalloc((void**)&c->done_flag, 128, hipHostMallocDefault));
  memset((void*)c->done_flag, 0, 128);
  HIPCHK(hipHostMalloc((void**)&c->bell, sizeof(AheadBell), hipHostMallocDefault));
  memset(c->bell, 0, sizeof(AheadBell));
  c->done_seq = 0;
  HIPCHK(hipMalloc((void**)&c->done_counter, 512));  // word 0: DoneSig::counter | words 4..: row tickets of the fused small commitment | word 64: AheadArgs::decision | words 80..95: AheadArgs::chal
  HIPCHK(hipMemset(c->done_counter, 0, 512));
  HIPCHK(hipMalloc((void**)&c->q_heads, 4 * (size_t)MSMQ_BLOCK_WORDS * MSMQ_BLOCKS));
#ifdef SP_KTIME
  if (c->opt.v[OPT_DEBUG_KTIME]) { HIPCHK(hipMalloc((void**)&c->ktime, 64 * 8)); HIPCHK(hipMemset(c->ktime, 0, 64 * 8)); }
#endif
  HIPCHK(hipEventCreateWithFlags(&c->sync_ev, hipEventDisableTiming));
  return SP_OK;
}
}  // extern "C"
// state derived from options (sp_ctx_set_option / sp_ctx_copy_options, options.hip); which < 0: all of it
void ctx_options_changed(sp_ctx* c, int which) {
  if (which < 0 || which == OPT_ENCODE_DEVICE) c->device_encode = c->opt.v[OPT_ENCODE_DEVICE] != 0;
  if (which < 0 || which == OPT_BG_EIGHTHS) c->bg_blocks = c->n_cus * (int)c->opt.v[OPT_BG_EIGHTHS] / 8;
#ifdef SP_KTIME
  if ((which < 0 || which == OPT_DEBUG_KTIME) && c->opt.v[OPT_DEBUG_KTIME] && !c->ktime && hipSetDevice(c->dev) == hipSuccess &&
      hipMalloc((void**)&c->ktime, 64 * 8) == hipSuccess)
    (void)hipMemset(c->ktime, 0, 64 * 8);
#endif
}
extern "C" {
void sp_ctx_destroy(sp_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->dev);
  ahead_cancel(c);
  (void)hipStreamSynchronize(c->stream);
  if (c->ahead.n_armed && sp_default_options().v[OPT_HOST_CALLSTATS])
    fprintf(stderr, "[callstats] kernels enqueued ahead of their challenges: %llu, rung %llu, cancelled %llu, gave up %llu\n", (unsigned long long)c->ahead.n_armed,
            (unsigned long long)c->ahead.n_rung, (unsigned long long)c->ahead.n_cancelled, (unsigned long long)c->ahead.n_gave_up);
  prof_drain(c);
  for (auto e : c->free_events) (void)hipEventDestroy(e);
  for (auto& kv : c->pool)
    for (void* p : kv.second) (void)hipFree(p);
  if (c->scratch) (void)hipFree(c->scratch);
  if (c->scratch2) (void)hipFree(c->scratch2);
  if (c->dstage) (void)hipFree(c->dstage);
  if (c->pinned) (void)hipHostFree(c->pinned);
  if (c->hmap) (void)hipHostFree(c->hmap);
  if (c->done_flag) (void)hipHostFree((void*)c->done_flag);
  if (c->bell) (void)hipHostFree((void*)c->bell);
  if (c->done_counter) (void)hipFree(c->done_counter);
  if (c->prof_epoch) (void)hipEventDestroy(c->prof_epoch);
  if (c->q_heads) (void)hipFree(c->q_heads);
  if (c->ktime) (void)hipFree(c->ktime);
  if (c->vm_pinned) (void)hipHostFree(c->vm_pinned);
  if (c->vm_dstage) (void)hipFree(c->vm_dstage);
  if (c->vm_ev) (void)hipEventDestroy(c->vm_ev);
  if (c->sync_ev) (void)hipEventDestroy(c->sync_ev);
  if (c->side_ev) (void)hipEventDestroy(c->side_ev);
  if (c->stream_side) (void)hipStreamDestroy(c->stream_side);
  if (c->stream_bg) (void)hipStreamDestroy(c->stream_bg);
  if (c->stream_low) (void)hipStreamDestroy(c->stream_low);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}
#ifdef SP_KTIME
// diagnostic build only (not in the header): the in-kernel time stamps of the last instrumented launch, in 100 MHz ticks
int32_t sp_debug_ktime(sp_ctx* c, long long* out, int n) {
  if (!c || !c->ktime || n > 64) return SP_EINVAL;
  HIPCHK(hipStreamSynchronize(c->stream));
  HIPCHK(hipMemcpy(out, c->ktime, 8 * (size_t)n, hipMemcpyDeviceToHost));
  HIPCHK(hipMemset(c->ktime, 0, 64 * 8));
  return SP_OK;
}
#endif
int sp_ctx_device(const sp_ctx* c) { return c ? c->dev : -1; }
uint64_t sp_ctx_trips(const sp_ctx* c) { return c ? c->sync_epoch : 0; }
int32_t sp_ctx_sync(sp_ctx* c) {
  if (!c) return SP_EINVAL;
  HIPCHK(hipSetDevice(c->dev));
  SPCHK(sync_spin(c));
  return hipGetLastError() == hipSuccess ? SP_OK : SP_EHIP;
}
int32_t sp_prof_enable(sp_ctx* c, int on) {
  if (!c) return SP_EINVAL;
  prof_drain(c);
  if (on && !c->prof_epoch) {  // the zero of the spans' clock: recorded once, on the main stream
    HIPCHK(hipSetDevice(c->dev));
    HIPCHK(hipEventCreate(&c->prof_epoch));
    HIPCHK(hipEventRecord(c->prof_epoch, c->stream));
    HIPCHK(hipEventSynchronize(c->prof_epoch));
  }
  c->prof_on = on;
  return SP_OK;
}
int32_t sp_prof_select(sp_ctx* c, const char* family) {
  if (!c) return SP_EINVAL;
  prof_drain(c);
  if (!family) { c->prof_mask = ~0ULL; return SP_OK; }
  for (int i = 0; i < PF_COUNT; i++)
    if (strcmp(kProfNames[i], family) == 0) { c->prof_mask = 1ULL << i; return SP_OK; }
  return SP_EINVAL;
}
int32_t sp_prof_reset(sp_ctx* c) {
  if (!c) return SP_EINVAL;
  prof_drain(c);
  memset(c->prof_ms, 0, sizeof c->prof_ms);
  memset(c->prof_n, 0, sizeof c->prof_n);
  memset(c->prof_bytes, 0, sizeof c->prof_bytes);
  memset(c->prof_ops, 0, sizeof c->prof_ops);
  c->prof_shapes.clear();
  c->prof_spans.clear();
  return SP_OK;
}
int32_t sp_prof_read_spans(sp_ctx* c, const char* family, uint64_t* shape, double* t0_ms, double* t1_ms, double* issued_adds, int cap) {
  if (!c || !family) return SP_EINVAL;
  prof_drain(c);
  int fam = -1;
  for (int i = 0; i < PF_COUNT; i++)
    if (strcmp(kProfNames[i], family) == 0) fam = i;
  if (fam < 0) return SP_EINVAL;
  int k = 0;
  for (auto& sp : c->prof_spans) {
    if (sp.fam != fam) continue;
    if (k < cap) {
      if (shape) shape[k] = sp.shape;
      if (t0_ms) t0_ms[k] = sp.t0;
      if (t1_ms) t1_ms[k] = sp.t1;
      if (issued_adds) issued_adds[k] = sp.issued;
    }
    k++;
  }
  return k;
}
int32_t sp_prof_read_ops(sp_ctx* c, double* alg_ops, int cap) {
  if (!c || !alg_ops) return SP_EINVAL;
  prof_drain(c);
  for (int i = 0; i < PF_COUNT && i < cap; i++) alg_ops[i] = c->prof_ops[i];
  return PF_COUNT;
}
int32_t sp_prof_read_shapes(sp_ctx* c, const char* family, uint64_t* shape, double* total_ms, uint64_t* launches, double* alg_bytes, double* alg_ops, int cap) {
  if (!c || !family) return SP_EINVAL;
  prof_drain(c);
  int fam = -1;
  for (int i = 0; i < PF_COUNT; i++)
    if (strcmp(kProfNames[i], family) == 0) fam = i;
  if (fam < 0) return SP_EINVAL;
  int k = 0;
  for (auto& kv : c->prof_shapes) {
    if (kv.first.first != fam) continue;
    if (k < cap) {
      if (shape) shape[k] = kv.first.second;
      if (total_ms) total_ms[k] = kv.second.ms;
      if (launches) launches[k] = kv.second.n;
      if (alg_bytes) alg_bytes[k] = kv.second.bytes;
      if (alg_ops) alg_ops[k] = kv.second.ops;
    }
    k++;
  }
  return k;
}
int sp_msm_window_bits(void) { return MSM_WBITS; }
int sp_gens_window_bits(const sp_gens* g) { return g ? g->geom.wbits : 0; }
int sp_gens_windows(const sp_gens* g) { return g ? g->geom.nwin : 0; }
// Window counts for TWO generator streams that one prover will hold side by side (SNARKGens: the gens_r1cs_sat stream, n_a points, commits
// w_a scalars per proof; the gens_r1cs_eval stream, n_b points, w_b scalars — 2^s and 6 * 2^s for SNARK::prove). choose_geom sees one set at
// a time: whichever is built first takes what it likes and the other what is left. Here the pair is chosen together: the (windows_a,
// windows_b) that minimises the mixed additions of a proof, w_a * windows_a + w_b * windows_b, among the pairs whose tables fit the free
// device memory less the proof's reserve (and msm.table_gb each); ties go to the smaller tables. 2^20: 17 / 17 (35.5 + 141.8 GB; one set
// at a time: 17 / 18); 2^22: 17 / 18 (the only pair that fits); 2^24: 22 / 20 (18.3 + 146 GB with 299 GB free; one at a time: 18 / 22, as much memory
// and 6 % more additions; 20 / 20 — 183 GB, fewer additions still — would leave the 2^24 proof too little). The caller sets option msm.windows to each result around the creation of its stream (0 = leave it to the per-set policy:
// returned when a geometry is forced by msm.wbits / msm.windows, or for streams too small to matter).
int32_t sp_gens_plan_pair(sp_ctx* c, size_t n_a, size_t n_b, double w_a, double w_b, int* windows_a, int* windows_b) {
  if (!c || !windows_a || !windows_b || n_a == 0 || n_b == 0) return SP_EINVAL;
  *windows_a = *windows_b = 0;
  if (c->opt.v[OPT_MSM_WBITS] >= 4 || c->opt.v[OPT_MSM_WINDOWS] >= 17 || c->opt.v[OPT_MSM_PLAN_PAIR] == 0) return SP_OK;
  HIPCHK(hipSetDevice(c->dev));
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); return SP_OK; }
  const double budget = (double)c->opt.v[OPT_MSM_TABLE_GB];
  const size_t n_max = n_a > n_b ? n_a : n_b;
  auto gb = [](size_t n, int nw) { return (double)n * (double)msm_geom_windows(nw).pt_entries * sizeof(Niels) / 1e9; };
  if (gb(n_max, 17) < 1.0) return SP_OK;  // small sets: everything fits, the per-set policy gives 17 windows anyway
  const double avail = (double)free_b / 1e9 - std::max(24.0, 5.0e-7 * (double)n_max * (double)n_max);
  static const int kCand[] = {17, 18, 19, 20, 21, 22, 24, 26, 32};
  double best_cost = 1e300, best_gb = 1e300;
  for (int wa : kCand)
    for (int wb : kCand) {
      const double ga = gb(n_a, wa), gbb = gb(n_b, wb);
      if (ga > budget || gbb > budget || ga + gbb > avail) continue;
      const double cost = w_a * wa + w_b * wb;
      if (cost < best_cost - 1e-9 || (cost < best_cost + 1e-9 && ga + gbb < best_gb)) { best_cost = cost; best_gb = ga + gbb; *windows_a = wa; *windows_b = wb; }
    }
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


// Window tables are public parameters (MultiCommitGens is immutable, commitments.rs:8-13) and large (7.5 MiB per point):
// they are built once per (device, generator bytes) and shared by every context of the process — concurrent proving
// contexts on one GPU hold one copy, and re-creating a SNARKGens is free. Reference-counted; freed with the last handle.
struct GensCacheEntry {
  int dev, mode;
  MsmGeom geom;
  size_t n, refs;
  std::vector<uint8_t> in, comp;
  Niels* table;
  NielsP* table_lds = nullptr;  // LDS-form tables (msm_lds.hip), built with the set when option msm.lds_bits asks for them — or when the wide tables came out narrow
  int wbits_lds = 0;
  bool prefer_lds = false;
};
static std::mutex g_gens_mu;
static std::list<GensCacheEntry> g_gens_cache;

// Window geometry of a generator set. A committed scalar costs one mixed addition (and one 128-byte gather) per window, so the policy
// minimises the NUMBER of windows under a memory budget; for each number of windows the table is as small as 254 bits allow (mixed
// widths, msm.hpp): per generator 17 windows = 34.6 MB, 18 = 21.0, 19 = 13.6, 20 = 8.9, 21 = 6.0, 22 = 4.5, 24 = 2.5, 26 = 1.5, 32 = 0.5.
//   * 17 windows only while the set's tables stay under msm.wide_gb (default 80: the 1025- and 2049-point streams of a 2^20 / 2^22
//     instance keep them; the proof time of 17 against 18 windows for the larger stream was a tie in rounds 2-4 and costs 56 GB);
//   * otherwise the fewest windows that fit msm.table_gb (default 180 per set: the 8194-point stream of a 2^22 instance at 18 windows is
//     172 GB) and the free device memory less a reserve for the proof's own tables: 24 GB, or 5.0e-7 GB x n^2 when that is more — the
//     working set of a proof grows with the square of its larger generator stream: 134 GB for the 16386 points of a 2^24 instance (instance,
//     encode, proof tables and what the buffer pool keeps: measured on the 309 GB (= 288 GiB) the device reports — with 124 GB left the bench's
//     2^24 run ran out of memory, with 132 it did not);
//     applied only to tables that are themselves large (a 1.5 MB table set must not be refused because another process holds the HBM).
// 2^20: 17 / 18 windows (35.5 + 86.1 GB; rounds 2-5: uniform 15 / 14 bits = 17 / 19 windows, 36.6 + 81.6 GB); 2^22: 17 / 18 (70.9 + 172 GB;
// before 17 / 19, 73 + 163); 2^24: 18 / 22 (86 + 73 GB; before 14 / 12 bits = 19 / 22 windows, 82 + 95). Option msm.windows forces a
// number of windows, msm.wbits a uniform width (the tests use both). Fewer windows than the first choice are reported on stderr (once per
// set): a silent narrowing would be a performance cliff nobody sees. Returns false when not even 32 windows fit in free memory.
static bool choose_geom(const sp_ctx* c, size_t n, MsmGeom* out) {
  if (c->opt.v[OPT_MSM_WBITS] >= 4) { *out = msm_geom((int)c->opt.v[OPT_MSM_WBITS]); return true; }
  if (c->opt.v[OPT_MSM_WINDOWS] >= 17) { *out = msm_geom_windows((int)c->opt.v[OPT_MSM_WINDOWS]); return true; }
  const double budget = (double)c->opt.v[OPT_MSM_TABLE_GB], wide = (double)c->opt.v[OPT_MSM_WIDE_GB];
  size_t free_b = 0, total_b = 0;
  double free_gb = 1e9;
  if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) free_gb = (double)free_b / 1e9;
  int first_choice = 0;  // what the policy picks when memory is no object
  for (int nw : {17, 18, 19, 20, 21, 22, 24, 26, 32}) {
    MsmGeom g = msm_geom_windows(nw);
    double gb = (double)n * (double)g.pt_entries * sizeof(Niels) / 1e9;
    if (nw == 17 && gb > wide) continue;
    if (gb > budget) continue;
    if (!first_choice) first_choice = nw;
    double reserve = gb >= 1.0 ? std::max(24.0, 5.0e-7 * (double)n * (double)n) : 0.25;  // room for the proof's working set next to a large table; a small table only has to fit
    if (gb + reserve <= free_gb || (nw == 32 && gb * 1.05 <= free_gb)) {
      if (nw > first_choice)
        fprintf(stderr, "spartan_hip: window tables of %zu generators cut from %d to %d windows (%.1f GB of device memory free): %d instead of %d additions per scalar\n",
                n, first_choice, nw, free_gb, nw, first_choice);
      *out = g;
      return true;
    }
  }
  return false;
}
const uint8_t* gens_compressed_bytes(const sp_gens* g, size_t* n) {
  const GensCacheEntry* e = (const GensCacheEntry*)g->cache_entry;
  *n = e->n;
  return e->mode == 0 ? e->in.data() : e->comp.data();  // the list is never modified while a handle on it exists
}
static int32_t gens_build(sp_ctx* c, const uint8_t* in, int mode, size_t n, uint8_t* comp_out, sp_gens** out) {
  if (!c || !in || !out || n == 0) return SP_EINVAL;
  HIPCHK(hipSetDevice(c->dev));
  size_t in_bytes = (mode == 0 ? 32 : 64) * n;
  std::lock_guard<std::mutex> lk(g_gens_mu);  // also serialises concurrent builds of the same table
  GensCacheEntry* hit = nullptr;
  for (auto& e : g_gens_cache)  // a resident table set serves every later handle on the same points, whatever width it was built with
    if (e.dev == c->dev && e.mode == mode && e.n == n && memcmp(e.in.data(), in, in_bytes) == 0) { hit = &e; break; }
  if (!hit) {
    MsmGeom geom;
    if (!choose_geom(c, n, &geom)) {
      fprintf(stderr, "spartan_hip: no window tables fit: %zu generators need at least %.2f GB of free device memory (32 windows)\n", n,
              (double)n * (double)msm_geom_windows(32).pt_entries * sizeof(Niels) / 1e9);
      return SP_ENOMEM;
    }
    const int wbits = geom.wbits;
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
    HIPCHK(hipMalloc((void**)&table, n * geom.pt_entries * sizeof(Niels)));
    NielsP* table_lds = nullptr;
    int lds_bits = (int)c->opt.v[OPT_MSM_LDS_BITS];   // 0 or 6..10 (options.hip refuses the rest)  // 10 bits: 512 x 96 B = 48 KB per sub-table, double-buffered in 96 of a CU's 160 KB of LDS
    // THE DEFAULT PER GENERATOR SET (measured: profiles/r5_ab_msm_forms.txt). While the wide tables keep >= 12 bits (17-22 additions per scalar)
    // the gathered forms win at every size measured (2^20, 2^22, 2^24). When HBM is short and the policy above lands on 10 or 8 bits (26-32
    // additions, at 0.68 of the addition ceiling), the LDS-staged form does the same 26 additions at 0.83 from tables that are SMALLER (96-byte
    // packed entries instead of 128-byte lines): the set then gets the packed tables too and its row commits take that form unless msm.form says otherwise.
    bool prefer_lds = false;
    // (not when msm.wbits forces the width: a caller that asks for 8- or 10-bit gathered tables gets them, and nothing outside its budget: ADVICE r5)
    if (!lds_bits && geom.nwin >= 26 && n >= 512 && c->opt.v[OPT_MSM_WBITS] < 4 && c->opt.v[OPT_MSM_WINDOWS] < 17) { lds_bits = 10; prefer_lds = true; }
    if (lds_bits && hipMalloc((void**)&table_lds, n * msm_geom(lds_bits).pt_entries * sizeof(NielsP)) != hipSuccess) {
      (void)hipGetLastError();
      table_lds = nullptr;
      if (!prefer_lds) { (void)hipFree(table); return SP_ENOMEM; }   // asked for by option: an error; chosen by the policy: do without
      lds_bits = 0; prefer_lds = false;
    }
    if (prefer_lds) fprintf(stderr, "spartan_hip: %zu generators with %d-window gathered tables: row commitments of this set take the LDS-staged form (10-bit windows streamed through LDS: the same 26 additions at a higher rate)\n", n, geom.nwin);
    {
      ProfScope ps(c, PF_GENS_TABLE, (double)n * geom.pt_entries * sizeof(Niels));
      hipLaunchKernelGGL(k_points_load, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, c->stream, base, mode, n, (Pt*)(base + off_pts),
                         mode == 1 ? base + off_comp : (uint8_t*)nullptr, (int*)(base + off_bad));
      size_t nt = n * (size_t)(geom.nwin + geom.nwide) * (size_t)(geom.tent < 128 ? 1 : geom.tent / 128);
      hipLaunchKernelGGL(k_table_build<Niels>, dim3((unsigned)((nt + 63) / 64)), dim3(64), 0, c->stream, (const Pt*)(base + off_pts), n, table, geom);
      if (lds_bits) {
        MsmGeom gl = msm_geom(lds_bits);
        size_t ntl = n * gl.nwin * (size_t)(gl.tent < 128 ? 1 : gl.tent / 128);
        hipLaunchKernelGGL(k_table_build<NielsP>, dim3((unsigned)((ntl + 63) / 64)), dim3(64), 0, c->stream, (const Pt*)(base + off_pts), n, table_lds, gl);
      }
    }
    int bad = 0;
    std::vector<uint8_t> comp(mode == 1 ? 32 * n : 0);
    int32_t rc = fetch_out(c, base + off_bad, &bad, 4);
    if (rc == SP_OK && mode == 1) rc = fetch_out(c, base + off_comp, comp.data(), 32 * n);
    if (rc == SP_OK && hipGetLastError() != hipSuccess) rc = SP_EHIP;
    if (rc != SP_OK || bad) {
      (void)hipFree(table);
      if (table_lds) (void)hipFree(table_lds);
      return rc != SP_OK ? rc : SP_EPOINT;
    }
    (void)wbits;
    g_gens_cache.push_back(GensCacheEntry{c->dev, mode, geom, n, 0, std::vector<uint8_t>(in, in + in_bytes), std::move(comp), table, table_lds, lds_bits, prefer_lds});
    hit = &g_gens_cache.back();
  }
  // a resident entry serves every later handle on the same points — but a context that asks for the LDS-staged form's tables must get them:
  // built for the cached entry now if it has none (msm.form = 1 silently took the gathered form otherwise); another width than
  // the one already there is reported and the resident tables kept
  if (const int want_lds = (int)c->opt.v[OPT_MSM_LDS_BITS]) {
    if (hit->table_lds && hit->wbits_lds != want_lds)
      fprintf(stderr, "spartan_hip: generator set of %zu points is resident with %d-bit LDS-form tables; msm.lds_bits = %d ignored for it\n", n, hit->wbits_lds, want_lds);
    if (!hit->table_lds) {
      MsmGeom gl = msm_geom(want_lds);
      NielsP* tl = nullptr;
      HIPCHK(hipStreamSynchronize(c->stream));
      size_t off_pts = (in_bytes + 255) & ~(size_t)255, off_bad = off_pts + n * sizeof(Pt) + ((32 * n + 255) & ~(size_t)255);
      SPCHK(ensure(&c->scratch, &c->scratch_cap, off_bad + 256));
      uint8_t* base = (uint8_t*)c->scratch;
      if (hipMalloc((void**)&tl, n * gl.pt_entries * sizeof(NielsP)) != hipSuccess) { (void)hipGetLastError(); return SP_ENOMEM; }
      HIPCHK(hipMemcpyAsync(base, hit->in.data(), in_bytes, hipMemcpyHostToDevice, c->stream));
      HIPCHK(hipMemsetAsync(base + off_bad, 0, 4, c->stream));
      hipLaunchKernelGGL(k_points_load, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, c->stream, base, mode, n, (Pt*)(base + off_pts), (uint8_t*)nullptr, (int*)(base + off_bad));
      size_t ntl = n * gl.nwin * (size_t)(gl.tent < 128 ? 1 : gl.tent / 128);
      hipLaunchKernelGGL(k_table_build<NielsP>, dim3((unsigned)((ntl + 63) / 64)), dim3(64), 0, c->stream, (const Pt*)(base + off_pts), n, tl, gl);
      if (hipStreamSynchronize(c->stream) != hipSuccess || hipGetLastError() != hipSuccess) { (void)hipFree(tl); return SP_EHIP; }
      hit->table_lds = tl; hit->wbits_lds = want_lds;
    }
  }
  sp_gens* g = new (std::nothrow) sp_gens();
  if (!g) {
    if (hit->refs == 0) {  // just built for this call
      (void)hipFree(hit->table);
      if (hit->table_lds) (void)hipFree(hit->table_lds);
      g_gens_cache.pop_back();
    }
    return SP_ENOMEM;
  }
  hit->refs++;
  if (mode == 1 && comp_out) memcpy(comp_out, hit->comp.data(), 32 * n);
  g->ctx = c;
  g->n = n;
  g->table = hit->table;
  g->geom = hit->geom;
  g->derived = hit->mode == 1;
  g->prefer_lds = hit->prefer_lds;
  g->table_lds = hit->table_lds;
  g->geom_lds = msm_geom(hit->wbits_lds ? hit->wbits_lds : 10);
  g->cache_entry = hit;
  *out = g;
  return SP_OK;
}
int32_t sp_gens_upload(sp_ctx* c, const uint8_t* compressed, size_t n, sp_gens** out) { return gens_build(c, compressed, 0, n, nullptr, out); }
int32_t sp_gens_from_uniform(sp_ctx* c, const uint8_t* uniform, size_t n, uint8_t* compressed_out, sp_gens** out) {
  return gens_build(c, uniform, 1, n, compressed_out, out);
}
size_t sp_gens_len(const sp_gens* g) { return g ? g->n : 0; }
size_t sp_gens_table_bytes(const sp_gens* g) {  // everything the set holds in HBM: the gathered tables and, when built, the packed LDS-form tables
  if (!g) return 0;
  return g->n * g->geom.pt_entries * sizeof(Niels) + (g->table_lds ? g->n * g->geom_lds.pt_entries * sizeof(NielsP) : 0);
}
void sp_gens_free(sp_gens* g) {
  if (!g) return;
  (void)hipSetDevice(g->ctx->dev);
  (void)hipStreamSynchronize(g->ctx->stream);
  (void)hipStreamSynchronize(g->ctx->stream_bg);  // a background commit may still be reading the tables
  {
    std::lock_guard<std::mutex> lk(g_gens_mu);
    GensCacheEntry* e = (GensCacheEntry*)g->cache_entry;
    if (--e->refs == 0) {
      host_commit_forget(e);
      (void)hipFree(e->table);
      if (e->table_lds) (void)hipFree(e->table_lds);
      for (auto it = g_gens_cache.begin(); it != g_gens_cache.end(); ++it)
        if (&*it == e) { g_gens_cache.erase(it); break; }
    }
  }
  delete g;
}

constexpr size_t SP_HOST_ENCODE_ROWS = 8;  // commitments of up to this many rows are encoded by the host core (see k_msm_reduce)
// MSM launch plan: kernel shapes and scratch sizes for a (rows x cols) fixed-base commit
struct MsmPlan {
  bool windowed, two_pass;
  bool lds = false;  // the LDS-staged small-window form (msm_lds.hip), P = runs per row-block
  bool queue = false; // the queue form (msm_queue.hip, k_msm_q): P = runs per row
  MsmQRuns qruns{0, 0, 0};
  int qrole = MSMQ_ALONE;
  int flat;  // 0: strip form; 2: balanced form (k_msm_flat), P = runs per row
  size_t strip, nstrips, P, chunk, nchunks, part_bytes, part2_bytes;
};
// workgroups of 256 threads the chip holds at once for the balanced row MSM (occupancy of the kernel x CUs), per device
static size_t msm_flat_slots() {
  const int pipe = 1;
  static std::mutex mu;
  static std::map<std::pair<int, int>, size_t> slots;  // (device, pipe) -> resident workgroups
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 768;
  std::lock_guard<std::mutex> lk(mu);
  auto it = slots.find({dev, pipe});
  if (it != slots.end()) return it->second;
  int per_cu = 0;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 768;
  hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_msm_flat, 256, 0);
  if (e != hipSuccess || per_cu < 1) per_cu = 3;
  return slots[{dev, pipe}] = (size_t)per_cu * (size_t)prop.multiProcessorCount;
}
// (c: the LAUNCHING context — options, CU count and background share are its own; a virtual shard launches its parent's generator set)
static MsmPlan msm_plan(const sp_ctx* c, const sp_gens* g, size_t rows, size_t cols, bool has_blinds, size_t bg_subblocks = 0 /* background launch: 256-thread tiles it runs at once */,
                        size_t launch_rows = 0 /* rows per launch when the commit is issued in row chunks (sp_commit_rows_upload_start) */,
                        bool shares_chip = false /* a background commit of this context is in flight */) {
  MsmPlan m;
  m.flat = 0;
  const size_t NWIN = (size_t)g->geom.nwin;
  size_t total = rows * cols, ncol = cols + (has_blinds ? 1 : 0);
  m.windowed = rows * ncol * NWIN <= ((size_t)1 << 19);  // latency-bound shapes: one addition per thread
  m.strip = 1; m.nstrips = 0;
  if (m.windowed) {
    m.P = ncol * NWIN;
  } else {
    const SpOptions& opt = c->opt;
    const size_t target_threads = 524288;  // (256 k and 1 M threads measured in round 3: both slower)
    m.strip = total / target_threads;  // enough threads for >= 4 waves per SIMD on 256 CUs
    if (m.strip < 1) m.strip = 1;
    if (m.strip > cols) m.strip = cols;
    m.nstrips = (cols + m.strip - 1) / m.strip;
    m.P = m.nstrips;
    if (launch_rows == 0) launch_rows = rows;
    // balanced form: a launch of at most four row-blocks that has the chip to itself (the witness commitment and its upload chunks,
    // stand-alone commits of few rows). Everything else that is not queue-sized keeps the strip form: the background launch (its
    // balanced form finished a 768 x 4096 commit in 4.7 instead of 5.8 ms on 5/8 of the chip and the latency-bound kernels next to it then
    // ran 2x slower instead of 1.3x: profiles/r4_ab_msm_forms.txt), a foreground commit that shares the chip with a background one (the
    // balanced form is exactly as many workgroups as an EMPTY chip holds, each as long as the launch), and a commit of many row-blocks
    // whose rows carry unlike scalars (equal additions per workgroup are equal time only when the scalars are alike)
    if (rows % 256 == 0 && launch_rows % 256 == 0 && launch_rows / 256 <= 4 && !bg_subblocks && !shares_chip) {
      const size_t rb = launch_rows / 256, units = ncol * NWIN;
      size_t nb = msm_flat_slots() / rb;        // runs per row: the launch is one full set of resident workgroups
      if (nb > units / 4) nb = units / 4;       // at least four additions per thread
      if (nb >= 1) { m.flat = 2; m.P = nb; }
    }
    if (opt.v[OPT_MSM_FORM] == 1 && !g->table_lds) {  // asked for, not possible: say so once instead of silently measuring another form
      static std::atomic<bool> warned{false};
      if (!warned.exchange(true)) fprintf(stderr, "spartan_hip: msm.form = 1 but the generator set was built without LDS-form tables (set msm.lds_bits before creating it): its row commitments take the default forms\n");
    }
    // LDS-staged small-window form: a workgroup is up to 1024 rows, so it needs rows to fill a CU with (>= 768 for 3 waves per SIMD)
    if (g->table_lds && (opt.v[OPT_MSM_FORM] == 1 || (opt.v[OPT_MSM_FORM] == 0 && g->prefer_lds)) && launch_rows >= 512) {
      const size_t cus = (size_t)c->n_cus;
      // one workgroup per CU (96 KB of LDS each); next to a background commit the launch is cut three times finer, so that the CUs
      // the background job leaves free are handed runs as they come
      size_t slots = bg_subblocks ? bg_subblocks / 4 : (shares_chip ? 3 * cus : cus);
      m.lds = true; m.flat = 0;
      m.P = msm_lds_runs(g, launch_rows, cols, has_blinds, slots);
    } else if ((opt.v[OPT_MSM_FORM] == 0 || opt.v[OPT_MSM_FORM] == 2) && launch_rows >= 256 && launch_rows == rows && (rows + 63) / 64 <= MSMQ_MAX_GROUPS) {
      // THE DEFAULT for commits of >= 256 rows since round 6: the queue form (msm_queue.hip). Measured against the strip / balanced forms it
      // replaces there (profiles/r6_ab_queue_form.txt): 8-18 % faster per launch, SNARK::prove 2^20 -0.6 ms; msm.form = 3 keeps the old forms.
      // (a commit issued in row chunks behind its upload has the chip to itself and rows that are alike: the balanced form's case)
      m.queue = true; m.flat = 0;
      // the background launch and a foreground launch that meets one in flight share the chip with each other and with the latency kernels
      m.qrole = bg_subblocks || shares_chip ? MSMQ_CORESIDENT : MSMQ_ALONE;
      m.qruns = msm_q_cut(c, g, launch_rows, cols, has_blinds, m.qrole);
      m.P = m.qruns.S;
    }
  }
  m.chunk = 1024; m.nchunks = (m.P + m.chunk - 1) / m.chunk;
  m.two_pass = m.P > 2048;
  m.part_bytes = (rows * m.P * sizeof(Pt) + 255) & ~(size_t)255;
  m.part2_bytes = m.two_pass ? ((rows * m.nchunks * sizeof(Pt10) + 255) & ~(size_t)255) : 0;
  return m;
}
// enqueue the kernels of one commit on `st`; scratch holds part_bytes + part2_bytes; dout receives 32*rows bytes
// (encode) or, with encode = false, 128*rows bytes of extended points
static void msm_enqueue(sp_ctx* c, hipStream_t st, bool prof, const MsmPlan& m, const sp_gens* g, const Fq* dZ, size_t z_stride, size_t rows,
                        size_t cols, size_t g_off, const uint32_t* didx, const Fq* dblinds, size_t h_idx, uint8_t* scratch, uint8_t* dout,
                        bool encode = true, uint8_t* sums_extra = nullptr /* 128*rows bytes for the row sums, or null */, bool do_msm = true,
                        bool do_reduce = true) {
  size_t total = rows * cols;
  Pt* partial = (Pt*)scratch;
  Pt10* partial2 = (Pt10*)(scratch + m.part_bytes);
  // many rows: sums first, then one lane per row encodes (k_pt_encode). The sums reuse the head of the partial buffer of
  // the pass that has already been consumed: two-pass -> `partial`, one-pass -> needs rows*128 B <= part2... so one-pass
  // shapes keep them behind the partials (the caller's scratch has 32*rows spare there only when sums_extra is set)
  bool batch_encode = encode && rows >= 64 && sums_extra != nullptr;
  Pt* sums = (Pt*)sums_extra;
  (void)prof;
  auto scope = [&](int fam, double bytes) { return ProfScope(c, fam, bytes, st); };  // HIP events on the stream the kernels run on
  const unsigned* qcounts = nullptr;  // queue form: partial-sum slots in use per 64-row group (device), for the reduction that follows
  if (!do_msm) {
  } else if (m.windowed) {
    ProfScope ps = scope(PF_MSM_WINDOWS, 32.0 * (double)total + 128.0 * (double)(rows * m.P));
    size_t nthreads = rows * m.P;
    hipLaunchKernelGGL(k_msm_windows, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, st, dZ, z_stride, rows, cols, (const Niels*)g->table,
                       g_off, didx, dblinds, h_idx, partial, g->geom);
  } else {
    // K1: 32 B read per committed scalar + 32 B written per row
    // ops: mixed additions if no scalar is zero; shape key: rows | cols | background launch
    uint64_t shape = ((uint64_t)rows << 32) | (uint64_t)cols | (st != c->stream ? (1ULL << 63) : 0);
    ProfScope ps(c, PF_MSM_ROWS, 32.0 * (double)total + 32.0 * (double)rows, st, (double)total * g->geom.nwin, shape);
    int xcd_map = rows % 256 == 0;
    size_t nblocks = xcd_map ? ((m.nstrips + 7) / 8) * 8 * (rows / 256) : (rows * m.nstrips + 255) / 256;
    if (m.queue) {
      msm_q_enqueue(c, st, g, dZ, z_stride, rows, cols, g_off, didx, dblinds, h_idx, partial, m.qruns, m.qrole, &qcounts, &ps.issued);
    } else if (m.lds) {
      msm_lds_enqueue(c, st, g, dZ, z_stride, rows, cols, g_off, didx, dblinds, h_idx, partial, m.P, st != c->stream && c->bg_blocks > 0 ? (unsigned)c->bg_blocks : 0u);
    } else if (m.flat) {
      MsmFlatArgs A{dZ, z_stride, rows, cols, (const Niels*)g->table, g_off, didx, dblinds, h_idx, partial, (unsigned)m.P, (unsigned)(rows / 256), g->geom};
      const unsigned ntiles = A.nb * A.rb_count;
      hipLaunchKernelGGL(k_msm_flat, dim3(ntiles), dim3(256), 0, st, A);
    } else if (st != c->stream && !didx && !dblinds && c->bg_blocks > 0) {
      hipLaunchKernelGGL(k_msm_rows_bg, dim3((unsigned)c->bg_blocks), dim3(1024), (unsigned)c->bg_lds, st, dZ, z_stride, rows, cols, m.strip, m.nstrips,
                         (const Niels*)g->table, g_off, partial, xcd_map, nblocks, g->geom);
    } else {
      hipLaunchKernelGGL(k_msm_rows<true>, dim3((unsigned)nblocks), dim3(256), 0, st, dZ, z_stride, rows, cols, m.strip, m.nstrips,
                         (const Niels*)g->table, g_off, didx, dblinds, h_idx, partial, xcd_map, g->geom);
    }
  }
  if (!do_reduce) return;
  if (m.two_pass) {
    {
      ProfScope ps = scope(PF_MSM_REDUCE_PASS, (double)(rows * m.P * sizeof(Pt)) + (double)(rows * m.nchunks * sizeof(Pt10)));
      hipLaunchKernelGGL(k_pt_reduce_pass, dim3((unsigned)rows, (unsigned)m.nchunks), dim3(256), 0, st, (const Pt*)partial, m.P, m.chunk, partial2, qcounts);
    }
    ProfScope ps = scope(PF_MSM_REDUCE, (double)(rows * m.nchunks * sizeof(Pt10)) + 32.0 * (double)rows);
    if (encode && batch_encode) hipLaunchKernelGGL((k_msm_reduce<true, false>), dim3((unsigned)rows), dim3(256), 0, st, (const void*)partial2, m.nchunks, (uint8_t*)sums, (const unsigned*)nullptr);
    else if (encode) hipLaunchKernelGGL((k_msm_reduce<true, true>), dim3((unsigned)rows), dim3(256), 0, st, (const void*)partial2, m.nchunks, dout, (const unsigned*)nullptr);
    else hipLaunchKernelGGL((k_msm_reduce<true, false>), dim3((unsigned)rows), dim3(256), 0, st, (const void*)partial2, m.nchunks, dout, (const unsigned*)nullptr);
  } else {
    ProfScope ps = scope(PF_MSM_REDUCE, (double)(rows * m.P * sizeof(Pt)) + 32.0 * (double)rows);
    if (encode && batch_encode) hipLaunchKernelGGL((k_msm_reduce<false, false>), dim3((unsigned)rows), dim3(256), 0, st, (const void*)partial, m.P, (uint8_t*)sums, qcounts);
    else if (encode) hipLaunchKernelGGL((k_msm_reduce<false, true>), dim3((unsigned)rows), dim3(256), 0, st, (const void*)partial, m.P, dout, qcounts);
    else hipLaunchKernelGGL((k_msm_reduce<false, false>), dim3((unsigned)rows), dim3(256), 0, st, (const void*)partial, m.P, dout, qcounts);
  }
  if (encode && batch_encode) {
    ProfScope ps = scope(PF_MSM_REDUCE, 160.0 * (double)rows);
    // (the background launch of a co-resident pair encodes under the other launch: the variant that fits next to it)
    if (m.queue && m.qrole == MSMQ_CORESIDENT && st != c->stream)
      hipLaunchKernelGGL(k_pt_encode_lean, dim3((unsigned)((rows + 63) / 64)), dim3(64), 0, st, (const Pt*)sums, rows, dout);
    else
      hipLaunchKernelGGL(k_pt_encode, dim3((unsigned)((rows + 63) / 64)), dim3(64), 0, st, (const Pt*)sums, rows, dout);
  }
}
// core: Z on device (row stride in elements), optional idx (device), optional blinds (device); synchronous
int32_t msm_launch(sp_ctx* c, const sp_gens* g, const Fq* dZ, size_t z_stride, size_t rows, size_t cols, size_t g_off,
                   const uint32_t* didx, const Fq* dblinds, size_t h_idx, uint8_t* out_host, size_t idx_row_stride, Pt* points_out) {
  if (idx_row_stride && (!didx || rows > SP_HOST_ENCODE_ROWS)) return SP_EINVAL;
  if (points_out && (rows > SP_HOST_ENCODE_ROWS || c->device_encode)) return SP_EINVAL;  // row sums as points: the few-row path only
  MsmPlan m = msm_plan(c, g, rows, cols, dblinds != nullptr, 0, 0, c->bg_inflight > 0);
  size_t out_al = (32 * rows + 255) & ~(size_t)255;
  SPCHK(ensure(&c->scratch, &c->scratch_cap, m.part_bytes + m.part2_bytes + out_al + sizeof(Pt) * rows));
  if (rows <= SP_HOST_ENCODE_ROWS) {  // latency path: the device sums, the host core runs the encode chain
    // (SPARTAN_DEVICE_ENCODE: the sums stay in device memory and k_pt_encode runs the chain there — same bytes, ~100 us later)
    uint8_t* sums_dst = c->device_encode ? (uint8_t*)c->scratch + m.part_bytes + m.part2_bytes + out_al : hres(c);
    if (m.windowed) {
      size_t nblk = (m.P + 255) / 256;
      Pt10* part = (Pt10*)c->scratch;  // nblk * rows * 160 B <= part_bytes
      if (nblk > 1 && !c->device_encode && c->done_counter) {
        DoneSig sig = sig_make(c, nblk * rows);
        {
          ProfScope ps(c, PF_MSM_WINDOWS, 32.0 * (double)(rows * cols) + 160.0 * (double)(rows * nblk));
          hipLaunchKernelGGL(k_msm_windows_tree_fused, dim3((unsigned)nblk, (unsigned)rows), dim3(256), 0, c->stream, dZ, z_stride, cols, (const Niels*)g->table, g_off,
                             didx, idx_row_stride, dblinds, h_idx, part, c->done_counter + 4, (Pt*)hres(c), g->geom, sig);
        }
        SPCHK(sig_wait(c, sig));
        if (hipGetLastError() != hipSuccess) return SP_EHIP;
        Pt sums[SP_HOST_ENCODE_ROWS];  // out of the host-mapped page first: the encodes read each coordinate several times
        memcpy(sums, hres(c), sizeof(Pt) * rows);
        if (points_out) memcpy(points_out, sums, sizeof(Pt) * rows);
        else pt_compress_many(sums, rows, out_host);
        return SP_OK;
      }
      {
        ProfScope ps(c, PF_MSM_WINDOWS, 32.0 * (double)(rows * cols) + 160.0 * (double)(rows * nblk));
        if (nblk == 1)
          hipLaunchKernelGGL((k_msm_windows_tree<true>), dim3(1, (unsigned)rows), dim3(256), 0, c->stream, dZ, z_stride, cols, (const Niels*)g->table,
                             g_off, didx, idx_row_stride, dblinds, h_idx, (void*)sums_dst, g->geom);
        else
          hipLaunchKernelGGL((k_msm_windows_tree<false>), dim3((unsigned)nblk, (unsigned)rows), dim3(256), 0, c->stream, dZ, z_stride, cols,
                             (const Niels*)g->table, g_off, didx, idx_row_stride, dblinds, h_idx, (void*)part, g->geom);
      }
      if (nblk > 1) {
        ProfScope ps(c, PF_MSM_REDUCE, 160.0 * (double)(rows * nblk) + 128.0 * (double)rows);
        hipLaunchKernelGGL((k_msm_reduce<true, false>), dim3((unsigned)rows), dim3(256), 0, c->stream, (const void*)part, nblk, sums_dst, (const unsigned*)nullptr);
      }
    } else {
      if (idx_row_stride) return SP_EINVAL;  // per-row index lists exist on the lookup+tree path only
      msm_enqueue(c, c->stream, true, m, g, dZ, z_stride, rows, cols, g_off, didx, dblinds, h_idx, (uint8_t*)c->scratch, sums_dst, false);
    }
    if (c->device_encode) {
      {
        ProfScope ps(c, PF_MSM_REDUCE, 160.0 * (double)rows);
        hipLaunchKernelGGL(k_pt_encode, dim3(1), dim3(64), 0, c->stream, (const Pt*)sums_dst, rows, hres(c));
      }
      SPCHK(fetch_small(c, out_host, 32 * rows));
      return hipGetLastError() == hipSuccess ? SP_OK : SP_EHIP;
    }
    Pt sums[SP_HOST_ENCODE_ROWS];
    SPCHK(fetch_small(c, sums, sizeof(Pt) * rows));
    if (hipGetLastError() != hipSuccess) return SP_EHIP;
    if (points_out) memcpy(points_out, sums, sizeof(Pt) * rows);
    else pt_compress_many(sums, rows, out_host);
    return SP_OK;
  }
  bool small_out = 32 * rows <= HMAP_SIZE - HMAP_IN;
  uint8_t* dout = small_out ? hres(c) : (uint8_t*)c->scratch + m.part_bytes + m.part2_bytes;
  msm_enqueue(c, c->stream, true, m, g, dZ, z_stride, rows, cols, g_off, didx, dblinds, h_idx, (uint8_t*)c->scratch, dout, true,
              (uint8_t*)c->scratch + m.part_bytes + m.part2_bytes + out_al);
  if (small_out) SPCHK(fetch_small(c, out_host, 32 * rows));
  else SPCHK(fetch_out(c, dout, out_host, 32 * rows));
  if (hipGetLastError() != hipSuccess) return SP_EHIP;
  return SP_OK;
}

// ---- background commit (overlaps a throughput-bound MSM with the latency-bound rounds that follow on the main stream)
int32_t sp_commit_rows_dev_begin(sp_ctx* c, const sp_gens* g, size_t g_off, const sp_table* Z, size_t z_off, size_t rows, size_t cols,
                                 sp_job** out) {
  if (!c || !g || !Z || !out || rows == 0 || cols == 0 || g_off + cols > g->n || z_off + rows * cols > Z->cap) return SP_EINVAL;
  HIPCHK(hipSetDevice(c->dev));
  MsmPlan m = msm_plan(c, g, rows, cols, false, c->bg_blocks > 0 ? (size_t)c->bg_blocks * 4 : 0);
  sp_job* j = new (std::nothrow) sp_job();
  if (!j) return SP_ENOMEM;
  j->ctx = c; j->rows = rows; j->scratch = nullptr; j->stream = c->stream_bg;
  size_t out_al = (32 * rows + 255) & ~(size_t)255;
  j->out_off = m.part_bytes + m.part2_bytes;
  j->scratch_bytes = j->out_off + out_al + sizeof(Pt) * rows;
  int32_t rc = pool_alloc(c, j->scratch_bytes, (void**)&j->scratch);
  if (rc != SP_OK) { delete j; return rc; }
  hipEvent_t ready;
  if (hipEventCreateWithFlags(&ready, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&j->done, hipEventDisableTiming) != hipSuccess) {
    pool_release(c, j->scratch, j->scratch_bytes); delete j; return SP_EHIP;
  }
  // the job reads Z as produced by everything queued so far on the main stream
  (void)hipEventRecord(ready, c->stream);
  (void)hipStreamWaitEvent(c->stream_bg, ready, 0);
  msm_enqueue(c, c->stream_bg, false, m, g, Z->d + z_off, cols, rows, cols, g_off, nullptr, nullptr, 0, j->scratch, j->scratch + j->out_off, true,
              j->scratch + j->out_off + out_al);
  (void)hipEventRecord(j->done, c->stream_bg);
  (void)hipEventDestroy(ready);
  if (hipGetLastError() != hipSuccess) { (void)hipEventDestroy(j->done); pool_release(c, j->scratch, j->scratch_bytes); delete j; return SP_EHIP; }
  c->bg_inflight++;
  *out = j;
  return SP_OK;
}
// Foreground variant: the commit (with blinds) is queued on the MAIN stream at full width and the call returns; the caller
// may do host work (it must not make another call on this context) until sp_job_wait. Used to hash the computation
// commitment into the transcript (0.65 ms of Keccak at 2^20) while the witness commitment is being computed.
int32_t sp_commit_rows_dev_start(sp_ctx* c, const sp_gens* g, size_t g_off, size_t h_idx, const sp_table* Z, size_t z_off, size_t rows, size_t cols,
                                 const uint64_t* blinds, sp_job** out) {
  if (!c || !g || !Z || !out || rows <= SP_HOST_ENCODE_ROWS || cols == 0 || g_off + cols > g->n || (blinds && h_idx >= g->n) ||
      z_off + rows * cols > Z->cap)
    return SP_EINVAL;
  HIPCHK(hipSetDevice(c->dev));
  const Fq* dbl = nullptr;
  if (blinds) {
    SPCHK(ensure_dstage(c, 32 * rows));
    SPCHK(stage_in(c, 0, blinds, 32 * rows));
    dbl = (const Fq*)c->dstage;
  }
  MsmPlan m = msm_plan(c, g, rows, cols, blinds != nullptr, 0, 0, c->bg_inflight > 0);
  sp_job* j = new (std::nothrow) sp_job();
  if (!j) return SP_ENOMEM;
  j->ctx = c; j->rows = rows; j->scratch = nullptr; j->stream = c->stream;
  size_t out_al = (32 * rows + 255) & ~(size_t)255;
  j->out_off = m.part_bytes + m.part2_bytes;
  j->scratch_bytes = j->out_off + out_al + sizeof(Pt) * rows;
  int32_t rc = pool_alloc(c, j->scratch_bytes, (void**)&j->scratch);
  if (rc != SP_OK) { delete j; return rc; }
  if (hipEventCreateWithFlags(&j->done, hipEventDisableTiming) != hipSuccess) { pool_release(c, j->scratch, j->scratch_bytes); delete j; return SP_EHIP; }
  msm_enqueue(c, c->stream, true, m, g, Z->d + z_off, cols, rows, cols, g_off, nullptr, dbl, h_idx, j->scratch, j->scratch + j->out_off, true,
              j->scratch + j->out_off + out_al);
  (void)hipEventRecord(j->done, c->stream);
  if (hipGetLastError() != hipSuccess) { (void)hipEventDestroy(j->done); pool_release(c, j->scratch, j->scratch_bytes); delete j; return SP_EHIP; }
  *out = j;
  return SP_OK;
}
// The same commit with the rows still in HOST memory (the satisfying assignment SNARK::prove is handed: src/lib.rs:339-344): the rows
// are copied into Z[z_off..] in four chunks and the MSM of a chunk is launched behind its copy, so the additions of chunk k run while
// chunk k + 1 crosses PCIe (32 MB at 2^20: 0.65 ms of copy, 0.87 ms of MSM); one reduction + encode over all rows at the end.
// Collected with sp_job_wait like the other forms. Shapes the chunking does not fit (few rows, rows not a multiple of 1024, the
// lookup form) take copy-then-commit.
int32_t sp_commit_rows_upload_start(sp_ctx* c, const sp_gens* g, size_t g_off, size_t h_idx, sp_table* Z, size_t z_off, const uint64_t* src, size_t rows,
                                    size_t cols, const uint64_t* blinds, sp_job** out) {
  if (!c || !g || !Z || !src || !out || rows <= SP_HOST_ENCODE_ROWS || cols == 0 || g_off + cols > g->n || (blinds && h_idx >= g->n) ||
      z_off + rows * cols > Z->cap)
    return SP_EINVAL;
  HIPCHK(hipSetDevice(c->dev));
  const size_t nch = (size_t)c->opt.v[OPT_UPLOAD_CHUNKS];
  const bool chunked_on = c->opt.v[OPT_UPLOAD_OVERLAP] != 0;  // A/B switch
  MsmPlan m = msm_plan(c, g, rows, cols, blinds != nullptr, 0, chunked_on && rows % (256 * nch) == 0 ? rows / nch : 0, c->bg_inflight > 0);
  if (!chunked_on || m.windowed || rows % (256 * nch) != 0) {
    HIPCHK(hipMemcpyAsync(Z->d + z_off, src, 32 * rows * cols, hipMemcpyHostToDevice, c->stream));
    return sp_commit_rows_dev_start(c, g, g_off, h_idx, Z, z_off, rows, cols, blinds, out);
  }
  const Fq* dbl = nullptr;
  if (blinds) {
    SPCHK(ensure_dstage(c, 32 * rows));
    SPCHK(stage_in(c, 0, blinds, 32 * rows));
    dbl = (const Fq*)c->dstage;
  }
  sp_job* j = new (std::nothrow) sp_job();
  if (!j) return SP_ENOMEM;
  j->ctx = c; j->rows = rows; j->scratch = nullptr; j->stream = c->stream;
  size_t out_al = (32 * rows + 255) & ~(size_t)255;
  j->out_off = m.part_bytes + m.part2_bytes;
  j->scratch_bytes = j->out_off + out_al + sizeof(Pt) * rows;
  int32_t rc = pool_alloc(c, j->scratch_bytes, (void**)&j->scratch);
  if (rc != SP_OK) { delete j; return rc; }
  if (hipEventCreateWithFlags(&j->done, hipEventDisableTiming) != hipSuccess) { pool_release(c, j->scratch, j->scratch_bytes); delete j; return SP_EHIP; }
  const size_t rch = rows / nch;
  // copies on the side stream, back to back; the additions of a chunk on the main stream, behind that chunk's copy
  hipEvent_t ready = nullptr;
  hipError_t e = hipEventCreateWithFlags(&ready, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventRecord(ready, c->stream);  // Z as left by what the main stream has queued (its zero fill)
  if (e == hipSuccess) e = hipStreamWaitEvent(c->stream_side, ready, 0);
  if (ready) (void)hipEventDestroy(ready);
  for (size_t k = 0; k < nch && e == hipSuccess; k++) {
    const size_t r0 = k * rch;
    // (a copy from pageable memory returns when the data has left the host buffer)
    e = hipMemcpyAsync(Z->d + z_off + r0 * cols, src + 4 * r0 * cols, 32 * rch * cols, hipMemcpyHostToDevice, c->stream_side);
    hipEvent_t copied = nullptr;  // one event per chunk (an event re-recorded while a wait on it is pending is not something to rely on)
    if (e == hipSuccess) e = hipEventCreateWithFlags(&copied, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventRecord(copied, c->stream_side);
    if (e == hipSuccess) e = hipStreamWaitEvent(c->stream, copied, 0);
    if (copied) (void)hipEventDestroy(copied);
    if (e == hipSuccess)
      msm_enqueue(c, c->stream, true, m, g, Z->d + z_off + r0 * cols, cols, rch, cols, g_off, nullptr, dbl ? dbl + r0 : nullptr, h_idx,
                  j->scratch + r0 * m.P * sizeof(Pt), nullptr, true, nullptr, true, false);
  }
  if (e == hipSuccess)
    msm_enqueue(c, c->stream, true, m, g, Z->d + z_off, cols, rows, cols, g_off, nullptr, dbl, h_idx, j->scratch, j->scratch + j->out_off, true,
                j->scratch + j->out_off + out_al, false, true);
  (void)hipEventRecord(j->done, c->stream);
  if (e != hipSuccess || hipGetLastError() != hipSuccess) {
    (void)hipStreamSynchronize(c->stream);
    (void)hipEventDestroy(j->done); pool_release(c, j->scratch, j->scratch_bytes); delete j;
    return SP_EHIP;
  }
  *out = j;
  return SP_OK;
}
int32_t sp_job_wait(sp_job* j, uint8_t* out) {
  if (!j || !out) return SP_EINVAL;
  sp_ctx* c = j->ctx;
  HIPCHK(hipSetDevice(c->dev));
  int32_t rc = SP_OK;
  for (;;) {
    hipError_t e = hipEventQuery(j->done);
    if (e == hipSuccess) break;
    if (e != hipErrorNotReady) { rc = SP_EHIP; break; }
  }
  if (rc == SP_OK) {
    if (hipMemcpyAsync(out, j->scratch + j->out_off, 32 * j->rows, hipMemcpyDeviceToHost, j->stream) != hipSuccess ||
        hipStreamSynchronize(j->stream) != hipSuccess)
      rc = SP_EHIP;
  }
  if (j->stream == c->stream_bg && c->bg_inflight > 0) c->bg_inflight--;
  // the scratch goes back to the pool, which hands it to main-stream work: make that work wait for the job
  (void)hipStreamWaitEvent(c->stream, j->done, 0);
  (void)hipEventDestroy(j->done);
  pool_release(c, j->scratch, j->scratch_bytes);
  delete j;
  return rc;
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
// A commitment restricted to a run of generators, left as points: sum_j Z[r * z_stride + j] * G[g_off + j], j < cols, for rows <= 8, no
// blind. The building block of a COLUMN-sharded commitment (the north-star's partial sums): every shard sums its slice of
// the generators, the W partial points of a row are gathered and added (sp_host_points_sum_encode) — the only way to shard a commitment
// with fewer rows than shards (Cx of DotProductProofLog, a small instance's witness).
int32_t sp_commit_rows_partial(sp_ctx* c, const sp_gens* g, size_t g_off, const sp_table* Z, size_t z_off, size_t z_stride, size_t rows, size_t cols,
                               sp_host_point* out) {
  if (!c || !g || !Z || !out || rows == 0 || rows > SP_HOST_ENCODE_ROWS || cols == 0 || z_stride < cols || g_off + cols > g->n ||
      z_off + (rows - 1) * z_stride + cols > Z->cap)
    return SP_EINVAL;
  static_assert(sizeof(sp_host_point) == sizeof(Pt), "sp_host_point carries an extended point");
  HIPCHK(hipSetDevice(c->dev));
  return msm_launch(c, g, Z->d + z_off, z_stride, rows, cols, g_off, nullptr, nullptr, 0, nullptr, 0, (Pt*)out);
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
int32_t msm_small_enqueue(sp_ctx* c, hipStream_t st, const sp_gens* g, const uint32_t* idx, size_t cols, const uint64_t* S, size_t rows,
                          uint8_t* sums_out) {
  if (!c || !g || !idx || !S || !sums_out || rows == 0 || rows > SP_HOST_ENCODE_ROWS || cols == 0 || cols * (size_t)g->geom.nwin > 256) return SP_EINVAL;
  for (size_t j = 0; j < cols; j++)
    if (idx[j] >= g->n) return SP_EINVAL;
  size_t sb = 32 * rows * cols, ib = (4 * cols + 31) & ~(size_t)31;
  if (sb + ib > HMAP_GEN) return SP_EINVAL;
  const Fq* ds = (const Fq*)stage_small(c, 0, S, sb);
  const uint32_t* di = (const uint32_t*)stage_small(c, sb, idx, 4 * cols);
  ProfScope ps(c, PF_MSM_WINDOWS, 32.0 * (double)(rows * cols) + 128.0 * (double)rows, st);
  hipLaunchKernelGGL((k_msm_windows_tree<true>), dim3(1, (unsigned)rows), dim3(256), 0, st, ds, cols, cols, (const Niels*)g->table, (size_t)0, di, (size_t)0,
                     (const Fq*)nullptr, (size_t)0, (void*)sums_out, g->geom);
  return SP_OK;
}
int32_t sp_msm_indexed(sp_ctx* c, const sp_gens* g, const uint32_t* idx, size_t cols, const uint64_t* S, size_t rows, uint8_t* out) {
  if (!c || !g || !idx || !S || !out || rows == 0 || cols == 0) return SP_EINVAL;
  for (size_t j = 0; j < cols; j++)
    if (idx[j] >= g->n) return SP_EINVAL;
  HIPCHK(hipSetDevice(c->dev));
  size_t sb = 32 * rows * cols, ib = (4 * cols + 31) & ~(size_t)31;
  if (sb + ib <= HMAP_GEN) {  // Sigma-protocol sized: the kernel reads scalars and indices straight from the host-mapped page
    const Fq* ds = (const Fq*)stage_small(c, 0, S, sb);
    const uint32_t* di = (const uint32_t*)stage_small(c, sb, idx, 4 * cols);
    return msm_launch(c, g, ds, cols, rows, cols, 0, di, nullptr, 0, out);
  }
  SPCHK(ensure_dstage(c, sb + ib));
  SPCHK(stage_in(c, 0, S, sb));
  SPCHK(stage_in(c, sb, idx, 4 * cols));
  return msm_launch(c, g, (const Fq*)c->dstage, cols, rows, cols, 0, (const uint32_t*)((uint8_t*)c->dstage + sb), nullptr, 0, out);
}

// ---- tables
int32_t table_new(sp_ctx* c, size_t len, bool zero, sp_table** out) {
  if (!c || !out || len == 0) return SP_EINVAL;
  HIPCHK(hipSetDevice(c->dev));
  sp_table* t = new (std::nothrow) sp_table();
  if (!t) return SP_ENOMEM;
  t->ctx = c;
  t->cap = t->len = len;
  t->owner = 1;
  t->d = nullptr;
  t->d_bytes = 32 * len;
  t->alt = nullptr;
  t->alt_bytes = 0;
  int32_t rc = pool_alloc(c, 32 * len, (void**)&t->d);
  if (rc != SP_OK) { delete t; return rc; }
  if (zero && hipMemsetAsync(t->d, 0, 32 * len, c->stream) != hipSuccess) { pool_release(c, t->d, 32 * len); delete t; return SP_EHIP; }
  *out = t;
  return SP_OK;
}
int32_t sp_table_alloc(sp_ctx* c, size_t len, sp_table** out) { return table_new(c, len, true, out); }
int32_t sp_table_alloc_uninit(sp_ctx* c, size_t len, sp_table** out) { return table_new(c, len, false, out); }
int32_t sp_table_write(sp_ctx* c, sp_table* t, size_t off, const uint64_t* Z, size_t len) {
  if (!c || !t || !Z || off + len > t->cap) return SP_EINVAL;
  HIPCHK(hipSetDevice(c->dev));
  HIPCHK(hipMemcpyAsync(t->d + off, Z, 32 * len, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));  // caller may reuse Z immediately
  return SP_OK;
}
int32_t sp_table_upload(sp_ctx* c, const uint64_t* Z, size_t len, sp_table** out) {
  if (!Z) return SP_EINVAL;
  SPCHK(table_new(c, len, false, out));
  int32_t rc = sp_table_write(c, *out, 0, Z, len);
  if (rc != SP_OK) { sp_table_free(*out); *out = nullptr; }
  return rc;
}
__global__ void k_copy_small(const Fq* __restrict__ src, size_t n, Fq* __restrict__ dst) { SP_FG_PRIO();
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) st_fq(dst + i, ld_fq(src + i));
}
int32_t sp_table_download(sp_ctx* c, const sp_table* t, size_t off, size_t len, uint64_t* out) {
  if (!c || !t || !out || off + len > t->cap) return SP_EINVAL;
  HIPCHK(hipSetDevice(c->dev));
  if (32 * len <= 4096) {  // a few elements (product-circuit roots, final claims): kernel write into the mapped page + poll
    hipLaunchKernelGGL(k_copy_small, dim3((unsigned)((len + 63) / 64)), dim3(64), 0, c->stream, (const Fq*)(t->d + off), len, (Fq*)hres(c));
    return fetch_small(c, out, 32 * len);
  }
  HIPCHK(hipMemcpyAsync(out, t->d + off, 32 * len, hipMemcpyDeviceToHost, c->stream));
  SPCHK(sync_spin(c));
  return SP_OK;
}
int32_t sp_table_clone(sp_ctx* c, const sp_table* t, sp_table** out) {
  if (!t) return SP_EINVAL;
  SPCHK(table_new(c, t->cap, false, out));
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
}  // extern "C"
int32_t table_ensure_alt(sp_table* t, size_t elems) {
  if (t->alt && t->alt_bytes >= 32 * elems) return SP_OK;
  if (t->alt) pool_release(t->ctx, t->alt, t->alt_bytes);
  t->alt = nullptr;
  t->alt_bytes = 0;
  SPCHK(pool_alloc(t->ctx, 32 * elems, (void**)&t->alt));
  t->alt_bytes = 32 * elems;
  return SP_OK;
}
void table_swap_to_alt(sp_table* t, size_t new_len) {
  Fq* old = t->d;
  int old_owned = t->owner;
  size_t old_bytes = t->d_bytes;
  t->d = t->alt; t->owner = 1; t->d_bytes = t->alt_bytes;
  t->cap = t->alt_bytes / 32; t->len = new_len;
  if (old_owned) { t->alt = old; t->alt_bytes = old_bytes; }
  else { t->alt = nullptr; t->alt_bytes = 0; }  // a view's original storage belongs to its parent
}
extern "C" {
void sp_table_free(sp_table* t) {
  if (!t) return;
  (void)hipSetDevice(t->ctx->dev);
  ahead_cancel(t->ctx);  // a kernel enqueued ahead of its challenges may hold this table's buffers
  if (t->owner) pool_release(t->ctx, t->d, t->d_bytes);
  if (t->alt) pool_release(t->ctx, t->alt, t->alt_bytes);
  delete t;
}


}  // extern "C"
