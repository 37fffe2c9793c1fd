// spartan_amd: resident sum-check sessions — the latency-bound tail of SumcheckInstanceProof::prove_cubic_batched
// (src/sumcheck.rs:254-424) without a kernel launch per round.
//
// A 2^20 SNARK proof runs ~400 batched cubic rounds (product_tree.rs:259-383 drives one sum-check per circuit layer), and
// ~330 of them work on tables of at most a few thousand entries: a few microseconds of arithmetic each, against ~25 us of
// launch + completion-flag kernel + host wake-up when every round is its own launch. A session keeps ONE kernel resident
// for all remaining rounds of a layer: its workgroups wait for the round challenge in a mailbox, bind every table at it
// (bound_poly_var_top, dense_mlpoly.rs:215-223), evaluate the next round's cubic at t = 0, 2, 3 and post their partial
// sums into per-workgroup slots of host memory; the Fiat-Shamir transcript stays on the host (in the drop-in: in Rust,
// merlin untouched), which adds the slots, derives the next challenge and drops it into the mailbox. A round trip is then
// two PCIe hops instead of a launch.
//
// Synchronisation. There is no device-side barrier: every workgroup synchronises with the HOST only. A workgroup posts
// its slot after a system-scope release fence (its writes to the tables are then visible device-wide), the host issues the
// next command only after ALL active slots of the round have arrived, and a workgroup starts a round with an agent-scope
// acquire — so round j+1 never reads an entry round j has not finished writing, whichever workgroup (or XCD) wrote it.
// Only workgroup (0,0) polls host memory (one PCIe read in flight); it republishes the command in a device-memory
// mailbox that the other workgroups poll. Every wait carries a wall-clock timeout: a session whose host went away exits
// by itself and flags the error; it cannot hang the GPU.
#include "internal.hpp"
#include <time.h>

namespace {

enum : uint32_t { SC_EVAL = 1, SC_ROUND = 2, SC_FINISH = 3, SC_ABORT = 4 };

struct SessCmd {  // host memory (coherent), written by the host: r and type first, then seq
  uint64_t seq;
  uint32_t type, pad;
  Fq r;
};
struct SessSlot {  // host memory, one per workgroup, written by the device: v first, then seq
  Fq v[3];
  uint64_t seq;
  uint64_t pad[3];
};
static_assert(sizeof(SessSlot) == 128, "slot layout");
struct SessDev {  // device memory: the command as republished by workgroup (0,0), and the error flag
  uint64_t seq;
  uint32_t type, err;
  Fq r;
};
struct SessInst {  // one batched instance: A and B bound in place; C ping-pongs between two buffers, written by its owner only
  Fq *a, *b, *c0, *c1;
  uint32_t c_owner, pad;
};
constexpr size_t SESS_GX_MAX = 16;    // workgroups per instance
constexpr size_t SESS_MAX_INST = 64;
constexpr uint64_t SESS_TIMEOUT_TICKS = 200000000ULL;  // 2 s of the 100 MHz wall clock

__device__ __forceinline__ uint64_t ld_sys(const uint64_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ Fq ld_fq_sys(const Fq* p) { return Fq{{ld_sys(&p->l[0]), ld_sys(&p->l[1]), ld_sys(&p->l[2]), ld_sys(&p->l[3])}}; }
__device__ __forceinline__ void st_fq_sys(Fq* p, const Fq& v) {
#pragma unroll
  for (int k = 0; k < 4; k++) __hip_atomic_store(&p->l[k], v.l[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

}  // namespace

// number of workgroups (per instance) that take part in a command on tables of current length `len`: 32 indices per
// workgroup pass; shared by the kernel (who stays) and the host (whose slots to wait for)
SP_HD size_t sess_active(uint32_t type, size_t len, size_t gx) {
  size_t work = type == SC_EVAL ? len / 2 : (type == SC_ROUND ? len / 4 : 1);
  size_t n = (work + 31) / 32;
  return n < 1 ? 1 : (n > gx ? gx : n);
}

// grid (gx, ninst), 256 threads: 32 indices x 8 lanes per pass, as k_cubic_bind_eval_tiny (spark.hip): lane 2k+h of an
// index handles half h of table k (load, bind), then lanes 0..2 evaluate t = 0, 2, 3 with one instruction stream.
__global__ void __launch_bounds__(256) k_cubic_session(const SessInst* __restrict__ insts, size_t len, uint64_t seq, const SessCmd* cmd, SessSlot* slots,
                                                       SessDev* dev, uint32_t flags, uint64_t* trace) {
  __shared__ Fq bound[32][6];
  __shared__ Fq red[3][32];
  __shared__ Fq sh_r;
  __shared__ uint32_t sh_type;
  const size_t gx = gridDim.x, bx = blockIdx.x, inst = blockIdx.y;
  const SessInst I = insts[inst];
  Fq* cc = I.c0;  // current C
  Fq* cn = I.c1;  // where the owner writes the bound C
  Fq* const ptr_ab[2] = {I.a, I.b};
  const int li = threadIdx.x >> 3, role = threadIdx.x & 7;
  SessSlot* slot = slots + inst * gx + bx;
  const uint64_t t0 = wall_clock64();
  for (;;) {
    // ---- wait for command `seq`
    if (threadIdx.x == 0) {
      uint32_t type = SC_ABORT;
      Fq r = fq_zero();
      bool ok = false;
      if (bx == 0 && inst == 0) {
        while (wall_clock64() - t0 < SESS_TIMEOUT_TICKS) {
          if (ld_sys(&cmd->seq) == seq) { ok = true; break; }
          __builtin_amdgcn_s_sleep(2);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");  // system scope: the command body is read after its sequence number
        if (ok) {
          type = (uint32_t)ld_sys((const uint64_t*)&cmd->type);
          r = ld_fq_sys(&cmd->r);
        } else {
          dev->err = 1;
        }
        dev->r = r;
        dev->type = type;
        __hip_atomic_store(&dev->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        // one copy per instance (256 B apart), so that a copy is polled by that instance's workgroups only
        for (size_t k = 1; k < gridDim.y; k++) {
          SessDev* d = (SessDev*)((uint8_t*)dev + 256 * k);
          d->r = r;
          d->type = type;
          __hip_atomic_store(&d->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
      } else {
        const SessDev* d = (const SessDev*)((const uint8_t*)dev + 256 * inst);
        while (wall_clock64() - t0 < SESS_TIMEOUT_TICKS) {
          if (__hip_atomic_load(&d->seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == seq) { ok = true; break; }  // no cache invalidate per poll
          __builtin_amdgcn_s_sleep(4);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (ok) {
          type = __hip_atomic_load(&d->type, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const uint64_t* rp = d->r.l;
#pragma unroll
          for (int k = 0; k < 4; k++) r.l[k] = __hip_atomic_load(rp + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      sh_type = type;
      sh_r = r;
      if (trace && bx == 0 && inst == 0) trace[0] = wall_clock64();
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // the other workgroups' table writes of the previous round
    const bool tr = trace && bx == 0 && inst == 0 && threadIdx.x == 0;
    if (tr) trace[1] = wall_clock64();
    const uint32_t type = sh_type;
    const Fq r = sh_r;
    if (type != SC_EVAL && type != SC_ROUND && type != SC_FINISH) return;
    Fq e = fq_zero();
    if (type == SC_FINISH) {
      // tables of length 2 -> 1 (the last round, sumcheck.rs:379-393) and the final claims poly[0] (:395-419)
      if (li == 0 && (role == 0 || role == 2 || (role == 4 && I.c_owner))) {
        Fq* p = role == 4 ? cc : ptr_ab[role >> 1];
        Fq x0 = ld_fq(p), x1 = ld_fq(p + 1);
        Fq v = fq_add(x0, fq_mul(r, fq_sub(x1, x0)));
        st_fq(p, v);
        st_fq_sys(&slot->v[role >> 1], v);
      } else if (li == 0 && role == 4) {
        st_fq_sys(&slot->v[2], fq_zero());
      }
    } else {
      const bool do_bind = type == SC_ROUND;
      const size_t span = do_bind ? len / 4 : len / 2;  // indices of this command; element (k, h) of index i sits at h*span + i
      for (size_t base = bx * 32; base < span; base += gx * 32) {
        const size_t i = base + li;
        const bool live = i < span;
        if (role < 6 && live) {
          const int k = role >> 1, h = role & 1;
          const Fq* src = k == 2 ? cc : ptr_ab[k];
          Fq v = ld_fq(src + (size_t)h * span + i);
          if (do_bind) {
            Fq x2 = ld_fq(src + (size_t)(2 + h) * span + i);
            v = fq_add(v, fq_mul(r, fq_sub(x2, v)));
            if (k < 2) st_fq(ptr_ab[k] + (size_t)h * span + i, v);
            else if (I.c_owner) st_fq(cn + (size_t)h * span + i, v);
          }
          bound[li][role] = v;
        }
        __syncthreads();
        if (role < 3 && live) {
          Fq a0 = bound[li][0], a1 = bound[li][1], b0 = bound[li][2], b1 = bound[li][3], c0 = bound[li][4], c1 = bound[li][5];
          Fq a2 = fq_sub(fq_dbl(a1), a0), b2 = fq_sub(fq_dbl(b1), b0), c2 = fq_sub(fq_dbl(c1), c0);
          Fq a3 = fq_sub(fq_add(a2, a1), a0), b3 = fq_sub(fq_add(b2, b1), b0), c3 = fq_sub(fq_add(c2, c1), c0);
          Fq av, bv, cv;
#pragma unroll
          for (int w = 0; w < 4; w++) {
            av.l[w] = role == 0 ? a0.l[w] : (role == 1 ? a2.l[w] : a3.l[w]);
            bv.l[w] = role == 0 ? b0.l[w] : (role == 1 ? b2.l[w] : b3.l[w]);
            cv.l[w] = role == 0 ? c0.l[w] : (role == 1 ? c2.l[w] : c3.l[w]);
          }
          e = fq_add(e, fq_mul(fq_mul(av, bv), cv));
        }
        __syncthreads();
      }
      if (tr) trace[2] = wall_clock64();
      if (role < 3) red[role][li] = e;
      __syncthreads();
      for (int s = 16; s > 0; s >>= 1) {
        if (role < 3 && li < s) red[role][li] = fq_add(red[role][li], red[role][li + s]);
        __syncthreads();
      }
      if (tr) trace[3] = wall_clock64();
      // the 96 bytes of the slot in ONE store instruction (one limb per lane): one PCIe write, not twelve
      if (threadIdx.x < 12)
        __hip_atomic_store(&((uint64_t*)slot->v)[threadIdx.x], red[threadIdx.x >> 2][0].l[threadIdx.x & 3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if (do_bind) {
        len /= 2;
        Fq* t = cc; cc = cn; cn = t;
      }
    }
    // ---- post: table writes and the slot payload first, then the slot's sequence number. The barrier orders every thread's
    // stores before thread 0's system-scope release (one L2 write-back per workgroup, not one per wave).
    if (flags & 1) __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
      if (tr) trace[4] = wall_clock64();
      __hip_atomic_store(&slot->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      if (tr) { trace[5] = wall_clock64(); trace += 6; }
    }
    if (type == SC_FINISH) return;
    seq++;
    // the next command is a ROUND while len >= 4, else the FINISH; a workgroup with nothing left to do leaves
    const uint32_t next = len >= 4 ? (uint32_t)SC_ROUND : (uint32_t)SC_FINISH;
    if (bx >= sess_active(next, len, gx)) return;
  }
}

static uint32_t sess_flags() {
  static const uint32_t v = [] { const char* e = getenv("SPARTAN_SESSION_FLAGS"); return e ? (uint32_t)atoi(e) : 0u; }();
  return v;
}
static bool sess_trace_on() {
  static const bool v = getenv("SPARTAN_SESSION_TRACE") != nullptr;
  return v;
}
static double sess_now() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
}
struct SessTrace { double post_us = 0, done_us = 0; };
struct sp_session {
  std::vector<SessTrace> tr;
  double t_begin = 0;
  sp_ctx* ctx;
  size_t ninst, gx, len, ncs;
  uint64_t seq;            // sequence number of the last command issued
  bool dead;
  std::vector<sp_table*> A, B, C;      // per instance
  std::vector<sp_table*> distinctC;    // in order of first appearance
  std::vector<size_t> c_of_inst;       // instance -> index into distinctC
  std::vector<size_t> owner_inst;      // distinct C -> the instance whose workgroups write it
  std::vector<Fq*> cbuf[2];            // distinct C -> its two buffers; cur tells which one holds the current values
  int cur;
};

static int32_t sess_wait(sp_session* s, uint32_t type, size_t len_before) {
  sp_ctx* c = s->ctx;
  size_t nact = sess_active(type, len_before, s->gx);
  const SessSlot* slots = (const SessSlot*)c->sess_slots;
  for (size_t i = 0; i < s->ninst; i++)
    for (size_t b = 0; b < nact; b++) {
      const uint64_t* p = &slots[i * s->gx + b].seq;
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
// sums of the active workgroups' slots: out[3*i + k]
static void sess_collect(sp_session* s, uint32_t type, size_t len_before, uint64_t* out) {
  size_t nact = sess_active(type, len_before, s->gx);
  const SessSlot* slots = (const SessSlot*)s->ctx->sess_slots;
  Fq* o = (Fq*)out;
  for (size_t i = 0; i < s->ninst; i++)
    for (int k = 0; k < 3; k++) {
      Fq acc = slots[i * s->gx].v[k];
      for (size_t b = 1; b < nact; b++) acc = fq_add(acc, slots[i * s->gx + b].v[k]);
      o[3 * i + k] = acc;
    }
}
// make the table structs describe what the kernel left behind: A, B bound in place; each distinct C in cbuf[cur]
static void sess_sync_tables(sp_session* s) {
  for (size_t i = 0; i < s->ninst; i++) { s->A[i]->len = s->len; s->B[i]->len = s->len; }
  for (size_t k = 0; k < s->ncs; k++) {
    sp_table* t = s->distinctC[k];
    if (t->d != s->cbuf[s->cur][k]) table_swap_to_alt(t, s->len);  // current values sit in what was the alternate buffer
    else t->len = s->len;
  }
}

extern "C" {

int32_t sp_sumcheck_session_begin(sp_ctx* c, sp_table* const* A, sp_table* const* B, sp_table* const* C, size_t ninst, int first_eval, uint64_t* out_evals,
                                  sp_session** out) {
  if (!c || !A || !B || !C || !out || ninst == 0 || ninst > SESS_MAX_INST || (first_eval && !out_evals)) return SP_EINVAL;
  size_t len = A[0] ? A[0]->len : 0;
  if (len < 2 || !is_pow2(len)) return SP_EINVAL;
  for (size_t k = 0; k < ninst; k++)
    if (!A[k] || !B[k] || !C[k] || A[k]->len != len || B[k]->len != len || C[k]->len != len) return SP_EINVAL;
  HIPCHK(hipSetDevice(c->dev));
  if (!c->sess_cmd) {  // mailboxes of this context: allocated on first use, reused by every session
    HIPCHK(hipHostMalloc((void**)&c->sess_cmd, 4096, hipHostMallocCoherent | hipHostMallocMapped));
    HIPCHK(hipHostMalloc((void**)&c->sess_slots, sizeof(SessSlot) * SESS_GX_MAX * SESS_MAX_INST, hipHostMallocCoherent | hipHostMallocMapped));
    HIPCHK(hipMalloc((void**)&c->sess_dev, 32768));  // 64 command copies of 256 B, then the trace area
    memset(c->sess_cmd, 0, 4096);
    memset(c->sess_slots, 0, sizeof(SessSlot) * SESS_GX_MAX * SESS_MAX_INST);
    HIPCHK(hipMemsetAsync(c->sess_dev, 0, 32768, c->stream));
    c->sess_seq = 0;
  }
  sp_session* s = new (std::nothrow) sp_session();
  if (!s) return SP_ENOMEM;
  s->ctx = c; s->ninst = ninst; s->len = len; s->dead = false; s->cur = 0;
  s->t_begin = sess_trace_on() ? sess_now() : 0;
  s->A.assign(A, A + ninst); s->B.assign(B, B + ninst); s->C.assign(C, C + ninst);
  s->c_of_inst.resize(ninst);
  for (size_t k = 0; k < ninst; k++) {
    size_t j = 0;
    while (j < s->distinctC.size() && s->distinctC[j] != C[k]) j++;
    if (j == s->distinctC.size()) { s->distinctC.push_back(C[k]); s->owner_inst.push_back(k); }
    s->c_of_inst[k] = j;
  }
  s->ncs = s->distinctC.size();
  // the second C buffer: half the current length is all a bound table ever needs
  for (size_t k = 0; k < s->ncs; k++) {
    sp_table* t = s->distinctC[k];
    int32_t rc = len >= 4 ? table_ensure_alt(t, len / 2) : SP_OK;
    if (rc != SP_OK) { delete s; return rc; }
    s->cbuf[0].push_back(t->d);
    s->cbuf[1].push_back(len >= 4 ? t->alt : t->d);
  }
  size_t span0 = first_eval ? len / 2 : len / 4;
  s->gx = (span0 + 31) / 32;
  if (s->gx < 1) s->gx = 1;
  if (s->gx > SESS_GX_MAX) s->gx = SESS_GX_MAX;
  std::vector<SessInst> insts(ninst);
  for (size_t k = 0; k < ninst; k++) {
    size_t j = s->c_of_inst[k];
    insts[k] = SessInst{A[k]->d, B[k]->d, s->cbuf[0][j], s->cbuf[1][j], s->owner_inst[j] == k ? 1u : 0u, 0u};
  }
  // the instance list is read by the kernel when it starts: it travels in the session's own corner of the command page
  SessInst* dinst = (SessInst*)((uint8_t*)c->sess_cmd + 256);
  static_assert(256 + sizeof(SessInst) * SESS_MAX_INST <= 4096, "command page layout");
  memcpy(dinst, insts.data(), sizeof(SessInst) * ninst);
  uint64_t seq0 = c->sess_seq + 1;
  {
    ProfScope ps(c, PF_SESSION, 0.0);
    hipLaunchKernelGGL(k_cubic_session, dim3((unsigned)s->gx, (unsigned)ninst), dim3(256), 0, c->stream, (const SessInst*)dinst, len, seq0,
                       (const SessCmd*)c->sess_cmd, (SessSlot*)c->sess_slots, (SessDev*)c->sess_dev, sess_flags(), sess_trace_on() ? (uint64_t*)((uint8_t*)c->sess_dev + 16384) : (uint64_t*)nullptr);
  }
  if (hipGetLastError() != hipSuccess) { delete s; return SP_EHIP; }
  if (first_eval) {
    sess_post(s, SC_EVAL, nullptr);
    int32_t rc = sess_wait(s, SC_EVAL, len);
    if (rc != SP_OK) { delete s; return rc; }
    sess_collect(s, SC_EVAL, len, out_evals);
  }
  *out = s;
  return SP_OK;
}

int32_t sp_sumcheck_session_round(sp_session* s, const uint64_t r[4], uint64_t* out_evals) {
  if (!s || !r || !out_evals || s->dead || s->len < 4) return SP_EINVAL;
  size_t len = s->len;
  sess_post(s, SC_ROUND, r);
  s->len = len / 2;
  s->cur ^= 1;
  for (size_t i = 0; i < s->ninst; i++) { s->A[i]->len = s->len; s->B[i]->len = s->len; }  // sp_table_len() stays truthful inside a session
  for (size_t k = 0; k < s->ncs; k++) s->distinctC[k]->len = s->len;                        // (which C buffer is current is settled at the end)
  SPCHK(sess_wait(s, SC_ROUND, len));
  sess_collect(s, SC_ROUND, len, out_evals);
  return SP_OK;
}

static void sess_release(sp_session* s) {
  if (sess_trace_on() && !s->tr.empty()) {  // diagnostic: where a round's time goes (host clock in us, device wall clock at 100 MHz)
    sp_ctx* c = s->ctx;
    (void)hipStreamSynchronize(c->stream);
    std::vector<uint64_t> dt(6 * s->tr.size());
    (void)hipMemcpy(dt.data(), (uint8_t*)c->sess_dev + 16384, 8 * dt.size(), hipMemcpyDeviceToHost);
    double host_rt = 0, host_gap = 0, d[6] = {0, 0, 0, 0, 0, 0};
    size_t n = s->tr.size(), nr = 0;
    for (size_t k = 0; k < n; k++) {
      host_rt += s->tr[k].done_us - s->tr[k].post_us;
      if (k) host_gap += s->tr[k].post_us - s->tr[k - 1].done_us;
      if (k + 1 == n && s->len == 1) continue;  // the FINISH command takes a different path through the kernel
      nr++;
      for (int j = 0; j < 5; j++) d[j] += (double)(dt[6 * k + j + 1] - dt[6 * k + j]) * 0.01;
      if (k) d[5] += (double)(dt[6 * k] - dt[6 * k - 1]) * 0.01;
    }
    if (!nr) nr = 1;
    fprintf(stderr, "[session] ninst %zu gx %zu cmds %zu | host us/cmd: post->all slots %.1f, think %.1f (begin->first post %.1f) | wg(0,0) us/cmd: seen->acquired %.1f, ->rounds done %.1f, ->tree %.1f, ->slot stored+barrier %.1f, ->seq released %.1f, ->next cmd seen %.1f\n",
            s->ninst, s->gx, n, host_rt / n, n > 1 ? host_gap / (n - 1) : 0.0, s->tr[0].post_us - s->t_begin, d[0] / nr, d[1] / nr, d[2] / nr, d[3] / nr, d[4] / nr,
            d[5] / (nr > 1 ? nr - 1 : 1));
  }
  sess_sync_tables(s);
  delete s;
}

int32_t sp_sumcheck_session_finish(sp_session* s, const uint64_t r[4], uint64_t* out_heads) {
  if (!s || !r || !out_heads || s->dead || s->len != 2) return SP_EINVAL;
  sess_post(s, SC_FINISH, r);
  int32_t rc = sess_wait(s, SC_FINISH, 2);
  if (rc == SP_OK) {
    const SessSlot* slots = (const SessSlot*)s->ctx->sess_slots;
    Fq* o = (Fq*)out_heads;
    for (size_t i = 0; i < s->ninst; i++) { o[2 * i] = slots[i * s->gx].v[0]; o[2 * i + 1] = slots[i * s->gx].v[1]; }
    for (size_t k = 0; k < s->ncs; k++) o[2 * s->ninst + k] = slots[s->owner_inst[k] * s->gx].v[2];
    s->len = 1;
  }
  sess_release(s);
  return rc;
}

void sp_sumcheck_session_abort(sp_session* s) {
  if (!s) return;
  if (!s->dead) {
    sess_post(s, SC_ABORT, nullptr);
    (void)hipStreamSynchronize(s->ctx->stream);  // the resident kernel reads the table buffers until it has seen the command
  }
  sess_release(s);
}

}  // extern "C"
