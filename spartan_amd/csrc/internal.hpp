// spartan_amd: internal declarations shared by the HIP translation units (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <map>
#include <vector>

#include "../../include/spartan_hip.h"
#include "curve.hpp"
#include "field.hpp"
#include "msm.hpp"
#include "options.hpp"

using namespace sp;

// ------------------------------------------------------------------------------------------------ host structs
enum ProfFamily {
  PF_GENS_TABLE = 0,
  PF_MSM_ROWS,
  PF_MSM_WINDOWS,
  PF_MSM_REDUCE_PASS,
  PF_MSM_REDUCE,
  PF_EQ_EXPAND,
  PF_SC_EVAL,
  PF_SC_BIND,
  PF_SC_BIND_EVAL,
  PF_VECMAT,
  PF_DOT,
  PF_REDUCE,
  PF_SPARSE,
  PF_IPA,
  PF_SPARK,
  PF_MISC,
  PF_COUNT
};
extern const char* kProfNames[PF_COUNT];

struct ProfRec {
  hipEvent_t e0, e1;
  int fam;
  const unsigned long long* issued;  // queue-form row MSM: device counter of the tiles (64 mixed additions each) the launch issued, or null
  uint64_t shape;  // launch shape inside the family (0 = not tracked): the MSM families key (rows, cols, background)
  double bytes, ops;
};
constexpr uint64_t PROF_SHAPE_BIG = 0x4000000000000001ULL;  // pseudo-shape: launches of an untracked family with >= 64 MB of algorithmic bytes
struct ProfSpan { int fam; uint64_t shape; double t0, t1, issued; };  // one tracked launch on the context's clock (ms since prof_epoch)
struct ProfShape {
  double ms = 0, bytes = 0, ops = 0;
  uint64_t n = 0;
};

// a queued job collected later with sp_job_wait: 32 * rows result bytes at scratch + out_off once `done` has fired
struct sp_job {
  sp_ctx* ctx;
  uint8_t* scratch;
  size_t scratch_bytes, rows, out_off;
  hipEvent_t done;
  hipStream_t stream;  // the stream the job runs on (a background stream, or the main stream for sp_commit_rows_dev_start)
};

// ---- a trip's kernel launched AHEAD of its challenges (round 6; bench/trip_probe.hip: launch + dispatch are 3-6 us of a ~20 us trip) -------------
// While the kernel of trip j runs, the kernel of trip j + 1 is already enqueued behind it on the main stream (in-order: it starts when j has
// finished, with j's tables visible) and waits for the proving thread to write the challenges into `AheadBell` and ring it. The first workgroup
// polls the bell in host memory and publishes the decision — go, or give up (cancelled by the host, or nothing heard for AHEAD_TIMEOUT) — in a device
// word the other workgroups poll; a kernel that gives up touches nothing, says so in host memory (`gave_up`), and the host launches the trip the
// ordinary way. Nothing can hang: every wait on either side is bounded.
struct AheadBell {        // host memory, 128-byte aligned (its own pinned allocation): ONE snapshot of it carries the bell and the challenges
  uint32_t bell;          // written last by the host: (sequence << 1) | cancel
  uint32_t check;         // sequence ^ the xor of r[0..16): a snapshot that shows the bell but not yet all of r is recognised and read again
  uint32_t r[16];         // the trip's challenges r0 | r1 as 32-bit words
  uint32_t pad[14];
};
struct AheadArgs {        // kernel argument; bell == nullptr: an ordinary launch, challenges in the arguments
  const AheadBell* bell;
  uint32_t* decision;          // device word: (sequence << 2) | {1 go, 2 give up}: the FIRST proposal for a sequence stands, every workgroup follows it
  uint32_t* chal;              // device: r0 | r1 (16 words) as a polling workgroup read them from the bell
  volatile uint32_t* gave_up;  // host word: the sequence of the last launch that gave up
  uint32_t seq;
};
static inline AheadArgs ahead_none() { return AheadArgs{nullptr, nullptr, nullptr, nullptr, 0}; }
constexpr long long AHEAD_TIMEOUT = 2000000;  // 20 ms of the 100 MHz wall clock
constexpr unsigned AHEAD_POLLERS = 32;        // workgroups (the first of the grid) that watch the bell in host memory themselves; the rest follow the decision word
#if defined(__HIPCC__)
// Threads 0..31 of every workgroup (half a wavefront, converged). Returns (to all of them) 1 = go: the challenges are in ah.chal; 2 = give up.
// A polling workgroup loads the bell's 128 bytes with one instruction (32 lanes x 4 bytes) — bell, check and challenges in one PCIe round trip —
// and proposes what it sees: go (a ring with a consistent snapshot), give up (a cancel, a bell that has moved on to a later launch, or nothing for
// AHEAD_TIMEOUT). The first proposal for this sequence number wins the compare-and-swap on the decision word and EVERY workgroup follows it, so a
// ring that arrives while some workgroup is timing out cannot split the grid.
__device__ __forceinline__ uint32_t ahead_wait(const AheadArgs& ah) {
  const long long t0 = wall_clock64();
  const unsigned lane = threadIdx.x;
  const unsigned wg = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
  const bool poller = wg < AHEAD_POLLERS;
  const uint32_t* line = reinterpret_cast<const uint32_t*>(ah.bell);
  uint32_t d = 0;
  uint32_t w = poller ? __hip_atomic_load(line + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0u;
  for (;;) {
    // the next look at the bell is under way while this one is examined: a PCIe read takes ~2 us, and the ring should be seen one read after it lands
    const uint32_t wn = poller ? __hip_atomic_load(line + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0u;
    const uint32_t v = __hip_atomic_load(ah.decision, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((v >> 2) == ah.seq) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); d = v & 3; break; }
    uint32_t mine = 0;
    if (poller) {
      const uint32_t b = __shfl(w, 0, 32), chk = __shfl(w, 1, 32);
      if ((b >> 1) == ah.seq) {
        if (b & 1) mine = 2;
        else {
          uint32_t x = (lane >= 2 && lane < 18) ? w : 0u;
#pragma unroll
          for (int m = 16; m > 0; m >>= 1) x ^= __shfl_xor(x, m, 32);
          if ((x ^ ah.seq) == chk) {
            mine = 1;
            if (lane >= 2 && lane < 18) __hip_atomic_store(ah.chal + (lane - 2), w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
      } else if ((b >> 1) > ah.seq) mine = 2;  // the bell has moved on to a later launch: this one was abandoned
    }
    if (!mine && wall_clock64() - t0 > (poller ? AHEAD_TIMEOUT : 4 * AHEAD_TIMEOUT)) mine = 2;
    if (mine) {
      uint32_t got = 0;
      if (lane == 0) {
        uint32_t old = v;
        const uint32_t want = (ah.seq << 2) | mine;
        for (;;) {  // the first proposal for this sequence stands
          if ((old >> 2) == ah.seq) { got = old & 3; break; }
          if (__hip_atomic_compare_exchange_strong(ah.decision, &old, want, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) {
            got = mine;
            if (mine == 2) { *ah.gave_up = ah.seq; __threadfence_system(); }
            break;
          }
        }
      }
      d = __shfl(got, 0, 32);
      break;
    }
    if (!poller) __builtin_amdgcn_s_sleep(2);
    w = wn;
  }
  return d;
}
__device__ __forceinline__ Fq ahead_challenge(const uint32_t* p) {  // from ah.chal, after the decision was seen (acquire)
  Fq x;
#pragma unroll
  for (int w = 0; w < 4; w++) {
    const uint64_t lo = __hip_atomic_load(p + 2 * w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), hi = __hip_atomic_load(p + 2 * w + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    x.l[w] = lo | (hi << 32);
  }
  return x;
}
#endif
struct sp_table;
struct sp_ctx;
// The context's main stream. A kernel that was enqueued ahead of its challenges (AheadArm) is only valid while NOTHING else is queued on, or waits
// for, that stream: every use of the handle — a launch, a copy, an event, a synchronisation, a comparison — first tells such a kernel to give up
// (`.s` is the raw handle, for the two places that poll the stream while they wait for the very trip that was enqueued ahead).
struct MainStream {
  hipStream_t s = nullptr;
  sp_ctx* owner = nullptr;
  inline operator hipStream_t() const;
  MainStream& operator=(hipStream_t v) { s = v; return *this; }
};
struct AheadArm {  // what the enqueued kernel was armed for: the next call has to be exactly this to ring it, anything else cancels it
  bool on = false;
  uint32_t seq = 0;
  size_t ninst = 0, len = 0, nblk = 0;
  bool host = false, weighted = false, tail = false;
  sp_table *A[24], *B[24], *C[24];
  const void* buf[24][4];  // the device buffers the enqueued kernel was given (a, b, c, c_out per instance): handles can be freed and their addresses reused
  Fq w[24];
  uint32_t sig_seq = 0, sig_total = 0;
  uint64_t n_armed = 0, n_rung = 0, n_cancelled = 0, n_gave_up = 0;  // printed with the call statistics (option host.callstats)
};

struct sp_ctx {
  int dev;
  SpOptions opt;  // options.hpp: a copy of the process-wide defaults at creation, changed by sp_ctx_set_option
  int n_cus = 256;
  MainStream stream;
  hipStream_t stream_side;  // same priority as `stream`: small commitments that run NEXT TO a sum-check kernel of the same round
  hipEvent_t side_ev;
  hipStream_t stream_bg;  // lower-priority background stream: throughput MSMs overlapped with latency-bound rounds
  hipStream_t stream_low = nullptr;  // a second low-priority stream: short streaming jobs (sp_sparse_evaluate_begin) that must not queue behind a background MSM
  bool device_encode;     // SPARTAN_DEVICE_ENCODE: small commitments are encoded by the device too (100 us instead of 3 us each)
  int bg_blocks;          // workgroups of a background MSM (one per CU, fewer than CUs); 0 = plain launches
  int bg_inflight = 0;    // background commits queued and not yet collected: while one runs, foreground commits keep the strip form (core.hip, msm_plan)
  size_t bg_lds;          // dynamic LDS each of them claims (a whole CU's)
  unsigned* q_heads = nullptr;  // queue form of the row MSM (msm_queue.hip): a ring of MSMQ_BLOCKS counter blocks, one per launch in flight
  unsigned q_next = 0;
  // scratch
  void* scratch;
  size_t scratch_cap;
  void* scratch2;
  size_t scratch2_cap;
  uint8_t* pinned;  // host pinned staging
  size_t pinned_cap;
  void* dstage;  // device staging for host inputs copied by DMA
  size_t dstage_cap;
  hipEvent_t sync_ev;
  // completion of the main stream is signalled by a one-thread kernel that stores a sequence number into host memory,
  // which the host polls: 6-9 us per round trip against 11-14 us for hipEventRecord + hipEventQuery (bench/flag_probe.hip)
  uint64_t sync_epoch;              // completed waits on the main stream
  uint64_t eq_slot_epoch[8];        // sync_epoch + 1 when the slot was last filled (0 = never)
  unsigned eq_next;
  volatile uint32_t* done_flag;
  uint32_t done_seq;
  uint32_t* done_counter = nullptr;  // DoneSig::counter
  long long* ktime = nullptr;        // DoneSig::kt (SP_KTIME builds with SPARTAN_KTIME set)
  uint8_t *vm_pinned = nullptr, *vm_dstage = nullptr;  // sp_vecmat_dev's own staging pair for L: the call does not wait (core.hip: vm_stage)
  size_t vm_cap = 0;
  hipEvent_t vm_ev = nullptr;
  struct { bool active; int kind; size_t nblk; bool on_host; uint32_t seq; } pend_eval = {false, 0, 0, false, 0};  // sp_sumcheck_bind_eval_start .. _collect
  uint8_t* hmap;  // host-mapped (fine-grained) page: small kernel inputs are read, small results written, without a DMA hop
  AheadBell* bell = nullptr;  // launches ahead of their challenges (AheadArm): the bell page; the decision word is done_counter[64], the relayed challenges done_counter[80..96), gave_up is done_flag[16]
  AheadArm ahead;
  uint32_t ahead_seq = 0;


  // size-class pool of device buffers: per-proof tables are recycled instead of hipMalloc/hipFree'd
  std::map<size_t, std::vector<void*>> pool;
  size_t pool_bytes;
  // profiling
  int prof_on;
  uint64_t prof_mask;  // bit i set = record HIP events for kernel family i
  std::vector<ProfRec> pending;
  std::vector<hipEvent_t> free_events;
  double prof_ms[PF_COUNT];
  uint64_t prof_n[PF_COUNT];
  double prof_bytes[PF_COUNT];
  double prof_ops[PF_COUNT];  // algorithmic field multiplications (F_q kernels) / mixed point additions (MSM kernels)
  std::map<std::pair<int, uint64_t>, ProfShape> prof_shapes;
  std::vector<ProfSpan> prof_spans;  // launches of the shape-tracked families (the row MSMs) as intervals: sp_prof_read_spans
  hipEvent_t prof_epoch = nullptr;
};
// an enqueued launch that waits for its bell is told to give up: called by whatever is about to queue other work on the main stream, or wait for it
static inline void ahead_cancel(sp_ctx* c) {
  if (!c->ahead.on) return;
  c->ahead.on = false;
  c->ahead.n_cancelled++;
  __atomic_store_n(&c->bell->bell, (c->ahead.seq << 1) | 1u, __ATOMIC_RELEASE);
}
inline MainStream::operator hipStream_t() const {
  if (owner) ahead_cancel(owner);
  return s;
}
struct sp_gens {
  sp_ctx* ctx;
  size_t n;
  Niels* table;  // [n][nwin][tent]; owned by the process-wide table cache (core.hip), shared between contexts
  MsmGeom geom;  // window geometry these tables were built with
  bool prefer_lds = false;  // the set's wide tables came out narrow (<= 10 bits: HBM was short): its row commits take the LDS-staged form by default
  bool derived = false;  // the points came out of the library's own hash-to-curve (sp_gens_from_uniform: MultiCommitGens::new), not from a caller's list
  NielsP* table_lds = nullptr;  // [n][nwin][tent] packed entries of the LDS-staged form (msm_lds.hip), or null when the set was built without it
  MsmGeom geom_lds;
  void* cache_entry;
};
struct sp_index {  // a usize vector kept as u32 on the device (addresses of the SPARK memory checks)
  sp_ctx* ctx;
  uint32_t* d;
  size_t n;
};
struct sp_table {
  sp_ctx* ctx;
  Fq* d;           // current contents
  size_t cap, len; // elements addressable through d / current (bound) length
  int owner;       // d came from the pool (0 for views into another table)
  size_t d_bytes;  // pool size of d when owned
  Fq* alt;         // second pool buffer for out-of-place binds of tables shared between kernel instances
  size_t alt_bytes;
};
// the 32-byte encodings of a generator set's points (kept by the process-wide table cache, core.hip); *n = number of points
extern "C" const uint8_t* gens_compressed_bytes(const sp_gens* g, size_t* n);
void host_commit_forget(const void* cache_entry);  // host_commit.hip: drop the host-side window tables of a freed set
int32_t table_ensure_alt(sp_table* t, size_t elems);
void table_swap_to_alt(sp_table* t, size_t new_len);

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


struct ProfScope {
  sp_ctx* c;
  int fam;
  hipEvent_t e0, e1;
  bool on;
  hipStream_t st;  // the stream the timed kernels are launched on
  uint64_t shape;
  double bytes, ops;
  const unsigned long long* issued = nullptr;
  ProfScope(sp_ctx* c_, int fam_, double bytes_, hipStream_t st_ = nullptr, double ops_ = 0.0, uint64_t shape_ = 0)
      : c(c_), fam(fam_), on(fam_ >= 0 && c_->prof_on != 0 && ((c_->prof_mask >> fam_) & 1)), st(st_ ? st_ : c_->stream), shape(shape_), bytes(bytes_), ops(ops_) {
    if (st == c->stream) ahead_cancel(c);  // a launch on the main stream that is not the one an enqueued kernel waits for (AheadArm)
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
    c->prof_ops[fam] += ops;
    (void)hipEventRecord(e0, st);
  }
  ~ProfScope() {
    if (!on) return;
    (void)hipEventRecord(e1, st);
    c->pending.push_back(ProfRec{e0, e1, fam, issued, shape, bytes, ops});
  }
};


int32_t ensure(void** p, size_t* cap, size_t need);
// stream-ordered pool: a buffer released here may be handed out again to a later call on the same stream
int32_t pool_alloc(sp_ctx* c, size_t bytes, void** out);
void pool_release(sp_ctx* c, void* p, size_t bytes);
extern "C" int32_t table_new(sp_ctx* c, size_t len, bool zero, sp_table** out);
int32_t ensure_pinned(sp_ctx* c, size_t need);
void prof_drain(sp_ctx* c);
int32_t stage_in(sp_ctx* c, size_t off, const void* src, size_t bytes);   // host -> c->dstage (+off), async
int32_t ensure_dstage(sp_ctx* c, size_t need);
int32_t fetch_out(sp_ctx* c, const void* dsrc, void* hdst, size_t bytes);  // device -> host, synchronous
int32_t vm_stage(sp_ctx* c, const void* src, size_t bytes, const void** dev);  // host -> device copy queued on the main stream, no wait; own buffers
// Completion signalled by the LAST KERNEL of a round trip itself instead of by a flag kernel queued behind it (~4 us of
// dispatch per trip, ~500 trips per proof): every workgroup fences its results to system scope and counts itself in; the
// last one stores the sequence number the host is spinning on. flag == nullptr: no signal (the kernel is not the last).
struct DoneSig {
  volatile uint32_t* flag;
  uint32_t* counter;  // device word, zero between kernels (the signalling workgroup resets it)
  uint32_t seq, total;  // total: workgroups of the launch
  long long* kt;        // diagnostic builds (-DSP_KTIME, bench/ktime_probe.py): device buffer for in-kernel time stamps; else null
};
static inline DoneSig sig_none() { return DoneSig{nullptr, nullptr, 0, 0, nullptr}; }
// in-kernel phase stamps of the first workgroup (100 MHz wall clock): compiled in only with -DSP_KTIME
#if defined(SP_KTIME) && defined(__HIPCC__)
#define SP_KT(sig, i) do { if ((sig).kt && threadIdx.x == 0) (sig).kt[i] = wall_clock64(); } while (0)
#else
#define SP_KT(sig, i) do { } while (0)
#endif
DoneSig sig_make(sp_ctx* c, size_t total_workgroups);
int32_t sig_wait(sp_ctx* c, const DoneSig& sig);  // reserves the next sequence number; wait for it with sync_wait(c, sig.seq)
#if defined(__HIPCC__)
__device__ __forceinline__ void signal_done(const DoneSig& d) {  // call once per workgroup, by all its threads, after its last result store
  if (!d.flag) return;
  __syncthreads();
  if (threadIdx.x == 0 && threadIdx.y == 0 && threadIdx.z == 0) {
    __threadfence_system();
    if (d.total > 1) {
      if (atomicAdd(d.counter, 1u) != d.total - 1) return;
      __threadfence_system();
      *d.counter = 0;
    }
    *d.flag = d.seq;
  }
}
#endif
uint32_t sync_post(sp_ctx* c);               // queue the completion flag behind everything on the main stream
int32_t sync_wait(sp_ctx* c, uint32_t seq);  // spin until that flag has arrived
// Enqueue (no wait) the lookups + tree of a commitment with <= 256 (column, window) pairs per row and rows <= 8 on `st`:
// scalars S[rows][cols] and generator indices are staged in the host-mapped input page, the row sums (extended points)
// land at sums_out (device-visible). core.hip.
extern "C" int32_t msm_small_enqueue(sp_ctx* c, hipStream_t st, const sp_gens* g, const uint32_t* idx, size_t cols, const uint64_t* S, size_t rows,
                                     uint8_t* sums_out);
// host-mapped page layout: [0, HMAP_GEN) inputs of calls that wait for their kernels, [HMAP_GEN, HMAP_IN) a ring of
// EQ_SLOTS slots for the challenge vectors of sp_eq_expand (which does not wait: the table is consumed by later launches
// on the same stream), [HMAP_IN, HMAP_SIZE) kernel results
constexpr size_t HMAP_IN = 32768, HMAP_SIZE = 65536, EQ_SLOTS = 8, EQ_SLOT_BYTES = 1280, HMAP_GEN = HMAP_IN - EQ_SLOTS * EQ_SLOT_BYTES;
void* stage_small(sp_ctx* c, size_t off, const void* src, size_t bytes);  // returns the device-visible address; bytes+off <= HMAP_IN
static inline uint8_t* hres(sp_ctx* c) { return c->hmap + HMAP_IN; }
// where a kernel should put its nblk x K partial sums: the host page when the calling thread can add them (reduce_and_fetch)
constexpr size_t HOST_SUM_BYTES = 30720;  // the last KiB of the 32 KiB result area is kept for row sums (sp_sumcheck_bind_eval_commit)
static inline Fq* partials_dst(sp_ctx* c, size_t nblk, int K) { return 32 * nblk * (size_t)K <= HOST_SUM_BYTES ? (Fq*)hres(c) : (Fq*)c->scratch; }
int32_t fetch_small(sp_ctx* c, void* hdst, size_t bytes);
int32_t sync_spin(sp_ctx* c);  // wait for everything queued on the context stream                 // stream sync + copy out of the result area
int32_t reduce_and_fetch(sp_ctx* c, Fq* partials, size_t nblk, int K, uint64_t* out);
// fixed-base MSM core: Z on device (row stride in elements), optional idx / blinds (device); out on host, synchronous
extern "C" int32_t msm_launch(sp_ctx* c, const sp_gens* g, const Fq* dZ, size_t z_stride, size_t rows, size_t cols, size_t g_off,
                              const uint32_t* didx, const Fq* dblinds, size_t h_idx, uint8_t* out_host, size_t idx_row_stride = 0,
                              Pt* points_out = nullptr /* rows <= 8: the row sums as extended points instead of encodings */);
// idx_row_stride: 0 = every row uses idx[0..cols); otherwise row r uses idx[r*idx_row_stride ..] (latency path only)

// LDS-staged small-window row MSM (msm_lds.hip): runs per row-block for `wg_slots` resident workgroups; enqueue of the lookups
// (partial[row][nb] extended points; the reduction is the caller's, as for the other forms). grid_limit != 0: persistent form on that many workgroups
size_t msm_lds_runs(const sp_gens* g, size_t rows, size_t cols, bool has_blinds, size_t wg_slots);
void msm_lds_enqueue(sp_ctx* c, hipStream_t st, const sp_gens* g, const Fq* dZ, size_t z_stride, size_t rows, size_t cols, size_t g_off,
                     const uint32_t* didx, const Fq* dblinds, size_t h_idx, Pt* partial, size_t nb, unsigned grid_limit);

// queue form (msm_queue.hip, k_msm_q): self-contained wavefronts with private LDS rings, items pulled from a device-side queue; the options and
// the queue heads are those of the LAUNCHING context c (a virtual shard launches its parent's generator set on its own streams)
struct MsmQRuns { unsigned nb, len, S; };  // queue form (msm_queue.hip): runs per row, units per run, partial-sum slots per row
constexpr unsigned MSMQ_MAX_GROUPS = 1024, MSMQ_BLOCK_WORDS = 2 * MSMQ_MAX_GROUPS + 2 /* + a 64-bit tile counter */, MSMQ_BLOCKS = 256;  // counter blocks of sp_ctx::q_heads
enum MsmQRole { MSMQ_ALONE = 0, MSMQ_CORESIDENT = 2 };  // what else runs on the chip next to a launch (msm_queue.hip, msm_q_shape)
MsmQRuns msm_q_cut(const sp_ctx* c, const sp_gens* g, size_t rows, size_t cols, bool has_blinds, int role);
void msm_q_enqueue(sp_ctx* c, hipStream_t st, const sp_gens* g, const Fq* dZ, size_t z_stride, size_t rows, size_t cols, size_t g_off,
                   const uint32_t* didx, const Fq* dblinds, size_t h_idx, Pt* partial, const MsmQRuns& r, int role,
                   const unsigned** counts_out /* per 64-row group: partial sums written (take min with r.S) */,
                   const unsigned long long** issued_out /* profiling runs: device counter of the tiles issued, else null */);

static inline size_t grid_for(size_t work, size_t maxblocks = 2048) {
  size_t b = (work + 255) / 256;
  if (b < 1) b = 1;
  return b > maxblocks ? maxblocks : b;
}
static inline bool is_pow2(size_t x) { return x && !(x & (x - 1)); }
static inline size_t ilog2(size_t x) {
  size_t l = 0;
  while (((size_t)1 << l) < x) l++;
  return l;
}

// Every kernel of the Fiat-Shamir chain (everything but the throughput-sized row MSMs and the table builds) raises the issue priority of its
// wavefronts at entry: VALU issue on a SIMD is arbitrated by priority, then age (MI355X_MICROARCH "two waves per SIMD"), and a latency
// kernel that shares a SIMD with the persistent wavefronts of a background MSM — older by construction, and ALU-saturating — otherwise
// gets the leftover issue slots only. No effect where nothing else shares the SIMD.
#ifndef SP_FG_PRIO_OFF   // (-DSP_FG_PRIO_OFF: the A/B variant without it)
#define SP_FG_PRIO() __builtin_amdgcn_s_setprio(3)
#else
#define SP_FG_PRIO() do { } while (0)
#endif

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


// partials[nblk][K] -> out[K] ; single block (defined in fq_ops.hip)
__global__ void k_reduce_partials(const Fq* __restrict__ partials, size_t nblk, int K, Fq* __restrict__ out);
