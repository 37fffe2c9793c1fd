// spartan_amd: resident sum-check sessions — the latency-bound tail of SumcheckInstanceProof::prove_cubic_batched
// (src/sumcheck.rs:254-424) without a kernel launch, and without a trip to HBM, per round.
//
// A 2^20 SNARK proof runs ~400 batched cubic rounds (product_tree.rs:259-383 drives one sum-check per circuit layer);
// ~270 of them work on tables of at most 512 entries. Such a round is three dependent F_q multiplications deep and
// nothing else, yet as a launch of its own it costs ~22 us: launch ramp, table reads from HBM, completion-flag kernel,
// host wake-up. A session keeps ONE kernel resident for the rest of a layer's sum-check: one 512-thread workgroup per
// batched instance copies that instance's three tables into LDS once (3 x 512 x 32 B = 48 KB) and keeps them there; each
// round the workgroup waits for the challenge in a host-memory mailbox, binds its tables at it in LDS
// (bound_poly_var_top, dense_mlpoly.rs:215-223), evaluates the next round's cubic at t = 0, 2, 3 and posts the three sums
// into its slot of host memory. The Fiat-Shamir transcript stays on the host (in the drop-in: in Rust, merlin untouched),
// which derives the next challenge and drops it into the mailbox: a round trip is two PCIe hops and ~4 multiplications.
//
// Ownership instead of synchronisation: a workgroup reads and writes only its own instance's tables. A C table shared by
// several instances (poly_C_par, sumcheck.rs:287-357) is copied into every sharer's LDS and bound there redundantly — one
// multiplication per entry, against a device-wide barrier per round. There is no device-side communication at all; every
// workgroup talks to the host only. (Measured first with a multi-workgroup variant that kept the tables in HBM and
// exchanged through device memory, profiles/r2_session_trace.txt: every system-scope release/acquire pair, every HBM
// re-read after it and every extra PCIe write cost more than the arithmetic of a round; it was no faster than a launch
// per round.)
// Every wait carries a wall-clock timeout: a session whose host went away writes its tables back and exits by itself.
#include "internal.hpp"
#include <time.h>

namespace {

enum : uint32_t { SC_EVAL = 1, SC_ROUND = 2, SC_FINISH = 3, SC_ABORT = 4 };
constexpr size_t SESS_MAX_LEN = 512;   // table entries a workgroup keeps in LDS (3 tables x 512 x 32 B = 48 KB)
constexpr size_t SESS_MAX_INST = 64;
constexpr int SESS_THREADS = 512;
constexpr uint64_t SESS_TIMEOUT_TICKS = 200000000ULL;  // 2 s of the 100 MHz wall clock

struct SessCmd {  // host memory (coherent), written by the host: r and type first, then seq
  uint64_t seq;
  uint32_t type, pad;
  Fq r;
};
struct SessSlot {  // host memory, one per workgroup, written by the device: v first (one store instruction), then seq
  Fq v[3];
  uint64_t seq;
  uint64_t pad[3];
};
static_assert(sizeof(SessSlot) == 128, "slot layout");
struct SessInst {  // the tables of one batched instance (current contents, current length)
  Fq *a, *b, *c;
  uint32_t c_owner, pad;  // this instance writes C back (one writer per distinct C table)
};

__device__ __forceinline__ uint64_t ld_sys(const uint64_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

// grid (ninst), 512 threads. LDS: the instance's A, B, C (3 x 512 entries) + the reduction array (3 x 128).
__global__ void __launch_bounds__(SESS_THREADS) k_cubic_session(const SessInst* __restrict__ insts, size_t len, uint64_t seq, const SessCmd* cmd,
                                                                SessSlot* slots, uint64_t* trace) {
  // limb-major LDS arrays: entry i of a table is the four words W[limb][i] — consecutive lanes touch consecutive 8-byte
  // words (no bank conflicts), where an array of 32-byte Fq structs puts eight lanes on the same banks
  __shared__ uint64_t TW[3][4][SESS_MAX_LEN];
  __shared__ uint64_t RW[3][4][128];
  __shared__ Fq sh_r;
  __shared__ uint32_t sh_type;
  const int tid = threadIdx.x;
  const SessInst I = insts[blockIdx.x];
  Fq* const gp[3] = {I.a, I.b, I.c};
  SessSlot* slot = slots + blockIdx.x;
  auto tget = [&](int k, size_t i) { return Fq{{TW[k][0][i], TW[k][1][i], TW[k][2][i], TW[k][3][i]}}; };
  auto tset = [&](int k, size_t i, const Fq& v) { TW[k][0][i] = v.l[0]; TW[k][1][i] = v.l[1]; TW[k][2][i] = v.l[2]; TW[k][3][i] = v.l[3]; };
  auto rget = [&](int p, int j) { return Fq{{RW[p][0][j], RW[p][1][j], RW[p][2][j], RW[p][3][j]}}; };
  auto rset = [&](int p, int j, const Fq& v) { RW[p][0][j] = v.l[0]; RW[p][1][j] = v.l[1]; RW[p][2][j] = v.l[2]; RW[p][3][j] = v.l[3]; };
  for (size_t x = tid; x < 3 * len; x += SESS_THREADS) tset((int)(x / len), x % len, ld_fq(gp[x / len] + x % len));
  const bool tr = trace && blockIdx.x == 0 && tid == 0;
  const uint64_t t0 = wall_clock64();
  for (;;) {
    // ---- wait for command `seq` (every workgroup polls the host mailbox itself: a handful of PCIe reads in flight)
    if (tid == 0) {
      uint32_t type = SC_ABORT;
      bool ok = false;
      while (wall_clock64() - t0 < SESS_TIMEOUT_TICKS) {
        if (ld_sys(&cmd->seq) == seq) { ok = true; break; }
        __builtin_amdgcn_s_sleep(2);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");  // system scope: the command body is read after its sequence number
      if (ok) {
        type = (uint32_t)ld_sys((const uint64_t*)&cmd->type);
        sh_r = Fq{{ld_sys(&cmd->r.l[0]), ld_sys(&cmd->r.l[1]), ld_sys(&cmd->r.l[2]), ld_sys(&cmd->r.l[3])}};
      }
      sh_type = type;
      if (tr) trace[0] = wall_clock64();
    }
    __syncthreads();
    const uint32_t type = sh_type;
    const Fq r = sh_r;
    const bool known = type == SC_EVAL || type == SC_ROUND || type == SC_FINISH;
    if (known && type != SC_EVAL) {
      // bind every table at r: T[k][x] += r (T[k][x + len/2] - T[k][x]), in place (entry x is read by its own task only)
      const size_t half = len / 2;
      for (size_t task = tid; task < 3 * half; task += SESS_THREADS) {
        const size_t k = task / half, x = task % half;
        Fq lo = tget((int)k, x), hi = tget((int)k, x + half);
        tset((int)k, x, fq_add(lo, fq_mul(r, fq_sub(hi, lo))));
      }
      len = half;
      __syncthreads();
    }
    if (tr) trace[1] = wall_clock64();
    if (!known || type == SC_FINISH) {
      // the end (the last round, length 2 -> 1: sumcheck.rs:379-393; or an abort / a time-out): the table objects describe
      // the bound tables again; at the end of a sum-check the remaining entries are the final claims (:395-419)
      for (size_t x = tid; x < 3 * len; x += SESS_THREADS) {
        const size_t k = x / len;
        if (k < 2 || I.c_owner) st_fq(gp[k] + x % len, tget((int)k, x % len));
      }
      if (!known) return;
      if (tid < 12) __hip_atomic_store(&((uint64_t*)slot->v)[tid], TW[tid >> 2][tid & 3][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    } else {
      // evaluations of sum_i A(t) B(t) C(t) at t = 0, 2, 3 over the pairs (i, i + len/2): thread = (point p, lane j)
      const size_t span = len / 2;
      const int p = tid >> 7, j = tid & 127;
      Fq e = fq_zero();
      if (p < 3)
        for (size_t i = j; i < span; i += 128) {
          Fq v[3];
#pragma unroll
          for (int k = 0; k < 3; k++) {
            Fq x0 = tget(k, i), x1 = tget(k, i + span);
            Fq x2 = fq_sub(fq_dbl(x1), x0), x3 = fq_sub(fq_add(x2, x1), x0);  // the line through (x0, x1) at t = 2, 3
#pragma unroll
            for (int w = 0; w < 4; w++) v[k].l[w] = p == 0 ? x0.l[w] : (p == 1 ? x2.l[w] : x3.l[w]);
          }
          e = fq_add(e, fq_mul(fq_mul(v[0], v[1]), v[2]));
        }
      if (p < 3) rset(p, j, e);
      __syncthreads();
      for (int s = 64; s > 0; s >>= 1) {
        if (p < 3 && j < s) rset(p, j, fq_add(rget(p, j), rget(p, j + s)));  // lanes beyond span hold zero
        __syncthreads();
      }
      if (tr) trace[2] = wall_clock64();
      // the 96 bytes of the slot in ONE store instruction (one limb per lane): one PCIe write, not twelve
      if (tid < 12) __hip_atomic_store(&((uint64_t*)slot->v)[tid], RW[tid >> 2][tid & 3][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // ---- post: the slot payload (and, at the end, the written-back tables) first, then the slot's sequence number
    __syncthreads();
    if (tid == 0) {
      __hip_atomic_store(&slot->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      if (tr) { trace[3] = wall_clock64(); trace += 4; }
    }
    if (type == SC_FINISH) return;
    seq++;
  }
}

bool sess_trace_on() {
  static const bool v = getenv("SPARTAN_SESSION_TRACE") != nullptr;
  return v;
}
double sess_now() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
}
struct SessTrace { double post_us = 0, done_us = 0; };

}  // namespace

struct sp_session {
  sp_ctx* ctx;
  size_t ninst, len, ncs;
  uint64_t seq;  // sequence number of the last command issued
  bool dead;
  std::vector<sp_table*> A, B, C;    // per instance
  std::vector<sp_table*> distinctC;  // in order of first appearance
  std::vector<size_t> owner_inst;    // distinct C -> the instance whose workgroup writes it back
  std::vector<SessTrace> tr;
  double t_begin = 0;
};

static int32_t sess_wait(sp_session* s) {
  sp_ctx* c = s->ctx;
  const SessSlot* slots = (const SessSlot*)c->sess_slots;
  for (size_t i = 0; i < s->ninst; i++) {
    const uint64_t* p = &slots[i].seq;
    for (uint64_t spins = 1; __atomic_load_n(p, __ATOMIC_ACQUIRE) != s->seq; spins++) {
      if ((spins & 0xFFFFF) == 0) {  // every ~ms: did the kernel die or time out?
        hipError_t e = hipStreamQuery(c->stream);
        if (e != hipErrorNotReady && __atomic_load_n(p, __ATOMIC_ACQUIRE) != s->seq) {
          fprintf(stderr, "spartan_hip: sum-check session ended without posting (%s)\n", hipGetErrorString(e));
          s->dead = true;
          return SP_EHIP;
        }
      }
    }
  }
  if (sess_trace_on() && !s->tr.empty()) s->tr.back().done_us = sess_now();
  return SP_OK;
}
static void sess_post(sp_session* s, uint32_t type, const uint64_t* r) {
  SessCmd* cmd = (SessCmd*)s->ctx->sess_cmd;
  if (sess_trace_on()) { s->tr.emplace_back(); s->tr.back().post_us = sess_now(); }
  cmd->type = type;
  if (r) memcpy(cmd->r.l, r, 32);
  s->seq = ++s->ctx->sess_seq;
  __atomic_store_n(&cmd->seq, s->seq, __ATOMIC_RELEASE);
}
static void sess_collect(sp_session* s, uint64_t* out) {  // out[4 * (3*i + k)]
  const SessSlot* slots = (const SessSlot*)s->ctx->sess_slots;
  for (size_t i = 0; i < s->ninst; i++) memcpy(out + 12 * i, slots[i].v, 96);
}
static void sess_set_len(sp_session* s) {
  for (size_t i = 0; i < s->ninst; i++) { s->A[i]->len = s->len; s->B[i]->len = s->len; }
  for (size_t k = 0; k < s->ncs; k++) s->distinctC[k]->len = s->len;
}
static void sess_release(sp_session* s) {
  if (sess_trace_on() && !s->tr.empty()) {  // diagnostic: where a round's time goes (host clock in us, device wall clock at 100 MHz)
    sp_ctx* c = s->ctx;
    (void)hipStreamSynchronize(c->stream);
    size_t n = s->tr.size(), nr = 0;
    std::vector<uint64_t> dt(4 * n);
    (void)hipMemcpy(dt.data(), (uint8_t*)c->sess_dev, 8 * dt.size(), hipMemcpyDeviceToHost);
    double host_rt = 0, host_gap = 0, d[4] = {0, 0, 0, 0};
    for (size_t k = 0; k < n; k++) {
      host_rt += s->tr[k].done_us - s->tr[k].post_us;
      if (k) host_gap += s->tr[k].post_us - s->tr[k - 1].done_us;
      if (k + 1 == n && s->len == 1) continue;  // the FINISH command takes a different path through the kernel
      nr++;
      for (int j = 0; j < 3; j++) d[j] += (double)(dt[4 * k + j + 1] - dt[4 * k + j]) * 0.01;
      if (k) d[3] += (double)(dt[4 * k] - dt[4 * k - 1]) * 0.01;
    }
    if (!nr) nr = 1;
    fprintf(stderr, "[session] ninst %zu cmds %zu | host us/cmd: post->all slots %.1f, think %.1f (begin->first post %.1f) | wg 0 us/cmd: cmd seen->bound %.1f, ->evaluated+summed %.1f, ->posted %.1f, ->next cmd seen %.1f\n",
            s->ninst, n, host_rt / n, n > 1 ? host_gap / (n - 1) : 0.0, s->tr[0].post_us - s->t_begin, d[0] / nr, d[1] / nr, d[2] / nr, d[3] / (nr > 1 ? nr - 1 : 1));
  }
  sess_set_len(s);
  delete s;
}

extern "C" {

size_t sp_sumcheck_session_max_len(void) { return SESS_MAX_LEN; }

int32_t sp_sumcheck_session_begin(sp_ctx* c, sp_table* const* A, sp_table* const* B, sp_table* const* C, size_t ninst, int first_eval, uint64_t* out_evals,
                                  sp_session** out) {
  if (!c || !A || !B || !C || !out || ninst == 0 || ninst > SESS_MAX_INST || (first_eval && !out_evals)) return SP_EINVAL;
  size_t len = A[0] ? A[0]->len : 0;
  if (len < 2 || len > SESS_MAX_LEN || !is_pow2(len)) return SP_EINVAL;
  for (size_t k = 0; k < ninst; k++)
    if (!A[k] || !B[k] || !C[k] || A[k]->len != len || B[k]->len != len || C[k]->len != len) return SP_EINVAL;
  HIPCHK(hipSetDevice(c->dev));
  if (!c->sess_cmd) {  // mailboxes of this context: allocated on first use, reused by every session
    HIPCHK(hipHostMalloc((void**)&c->sess_cmd, 4096, hipHostMallocCoherent | hipHostMallocMapped));
    HIPCHK(hipHostMalloc((void**)&c->sess_slots, sizeof(SessSlot) * SESS_MAX_INST, hipHostMallocCoherent | hipHostMallocMapped));
    HIPCHK(hipMalloc((void**)&c->sess_dev, 4096));  // trace area (SPARTAN_SESSION_TRACE)
    memset(c->sess_cmd, 0, 4096);
    memset(c->sess_slots, 0, sizeof(SessSlot) * SESS_MAX_INST);
    c->sess_seq = 0;
  }
  sp_session* s = new (std::nothrow) sp_session();
  if (!s) return SP_ENOMEM;
  s->ctx = c; s->ninst = ninst; s->len = len; s->dead = false;
  s->t_begin = sess_trace_on() ? sess_now() : 0;
  s->A.assign(A, A + ninst); s->B.assign(B, B + ninst); s->C.assign(C, C + ninst);
  std::vector<SessInst> insts(ninst);
  for (size_t k = 0; k < ninst; k++) {
    size_t j = 0;
    while (j < s->distinctC.size() && s->distinctC[j] != C[k]) j++;
    bool owner = j == s->distinctC.size();
    if (owner) { s->distinctC.push_back(C[k]); s->owner_inst.push_back(k); }
    insts[k] = SessInst{A[k]->d, B[k]->d, C[k]->d, owner ? 1u : 0u, 0u};
  }
  s->ncs = s->distinctC.size();
  // the instance list is read by the kernel when it starts: it travels in the session's own corner of the command page
  SessInst* dinst = (SessInst*)((uint8_t*)c->sess_cmd + 256);
  static_assert(256 + sizeof(SessInst) * SESS_MAX_INST <= 4096, "command page layout");
  memcpy(dinst, insts.data(), sizeof(SessInst) * ninst);
  uint64_t seq0 = c->sess_seq + 1;
  {
    ProfScope ps(c, PF_SESSION, 96.0 * (double)len * (double)ninst);
    hipLaunchKernelGGL(k_cubic_session, dim3((unsigned)ninst), dim3(SESS_THREADS), 0, c->stream, (const SessInst*)dinst, len, seq0,
                       (const SessCmd*)c->sess_cmd, (SessSlot*)c->sess_slots, sess_trace_on() ? (uint64_t*)c->sess_dev : (uint64_t*)nullptr);
  }
  if (hipGetLastError() != hipSuccess) { delete s; return SP_EHIP; }
  if (first_eval) {
    sess_post(s, SC_EVAL, nullptr);
    int32_t rc = sess_wait(s);
    if (rc != SP_OK) { delete s; return rc; }
    sess_collect(s, out_evals);
  }
  *out = s;
  return SP_OK;
}

int32_t sp_sumcheck_session_round(sp_session* s, const uint64_t r[4], uint64_t* out_evals) {
  if (!s || !r || !out_evals || s->dead || s->len < 4) return SP_EINVAL;
  sess_post(s, SC_ROUND, r);
  s->len /= 2;
  sess_set_len(s);  // sp_table_len() stays truthful inside a session
  SPCHK(sess_wait(s));
  sess_collect(s, out_evals);
  return SP_OK;
}

int32_t sp_sumcheck_session_finish(sp_session* s, const uint64_t r[4], uint64_t* out_heads) {
  if (!s || !r || !out_heads || s->dead || s->len != 2) return SP_EINVAL;
  sess_post(s, SC_FINISH, r);
  int32_t rc = sess_wait(s);
  if (rc == SP_OK) {
    const SessSlot* slots = (const SessSlot*)s->ctx->sess_slots;
    Fq* o = (Fq*)out_heads;
    for (size_t i = 0; i < s->ninst; i++) { o[2 * i] = slots[i].v[0]; o[2 * i + 1] = slots[i].v[1]; }
    for (size_t k = 0; k < s->ncs; k++) o[2 * s->ninst + k] = slots[s->owner_inst[k]].v[2];
    s->len = 1;
  }
  sess_release(s);
  return rc;
}

void sp_sumcheck_session_abort(sp_session* s) {
  if (!s) return;
  if (!s->dead) {
    sess_post(s, SC_ABORT, nullptr);
    (void)hipStreamSynchronize(s->ctx->stream);  // the resident kernel writes its tables back before it leaves
  }
  sess_release(s);
}

}  // extern "C"
