// spartan_amd: Bulletproofs-style inner-product reduction (BulletReductionProof::prove, src/nizk/bullet.rs:32-132)
// with device-resident a, b and generator-coefficient vector s. See include/spartan_hip.h for the algebra:
// the folded generator vector is never built; every L/R is a fixed-base MSM over the uploaded generators.
#include "internal.hpp"
#include <utility>

struct sp_ipa {
  sp_ctx* ctx;
  const sp_gens* g;
  size_t n0, n_cur, g_off, q_idx, h_idx;
  Fq q_scale;
  Fq *a, *b, *a2, *b2, *s, *s2, *rows;  // device: a, b, a2, b2, s, s2 [n0] each (ping-pong pairs), rows[2][n0+2] (round L/R use [2][n0/2+2])
  uint8_t* base;                       // the one pool allocation behind all of the above (a and a2 swap roles, so `a` is not it)
  bool fold_pending;                   // sp_ipa_round_fold only records (u, u^-1): the next launch that needs a, b, s applies it
  Fq fu, fu_inv;
  uint32_t* idx;              // device: n0+2 generator indices (all generators, Q, H)
  uint32_t* idx_lr;           // device: [2][n0/2+2] per-row generator lists of the current round
  uint32_t* counters;         // device: [2] row tickets of k_ipa_round (zero between launches)
  size_t bytes;
  // one-launch rounds (k_ipa_round, core.hip): c_L, c_R of the first round (k_ipa_init) and the eight quarter dot products the
  // last round left behind, from which the next c_L, c_R follow once the fold challenge is recorded
  Fq c0[2], dots[8];
  bool have_c0, have_dots;
  // sp_ipa_round_prelaunch: the round's kernel is already in flight (it depends on neither the blinds nor Q's scale)
  bool pre = false;
  DoneSig pre_sig;
  unsigned pre_nd = 0;
  Fq pre_cL, pre_cR;
  struct IpaRoundArgs* last_args = nullptr;  // the arguments of the round in flight (owned): a row sum with Z = 0 re-runs them with the unified tree
  // what the LAST round (two entries) left on the host: its vectors a', b' and its two row sums without the c Q + blind H tails,
  //   raw[0] = a'[0] * G'_R,  raw[1] = a'[1] * G'_L   (G'_L, G'_R: the two folded generators the reference holds at that point, bullet.rs:108)
  bool have_fin = false;
  Fq fin_a[2], fin_b[2];
  Pt fin_raw[2];
};
// core.hip
struct IpaRoundArgs {
  const Fq *a, *b, *s;
  Fq *a_new, *b_new, *s_new;
  size_t n_cur, n0, g_off;
  int fold;
  Fq u, u_inv;
  Fq u2, u2_inv;               // u^2, u^-2: the lookups' scalar a'[i] s'[p] in two multiplications (k_ipa_round)
  Pt10* part;
  uint32_t* counters;
  Pt* sums_out;
  Fq* dots_out;
  unsigned nblk, nd;
};
extern "C" int32_t ipa_round_launch(sp_ctx* c, const sp_gens* g, IpaRoundArgs* A, DoneSig* sig_out, int unified);
static unsigned ipa_c0_blocks(size_t n) { return (unsigned)((n / 2 + 511) / 512); }  // k_ipa_init: one workgroup per 512 index pairs
static bool ipa_fused(const sp_ctx* c) { return c->opt.v[OPT_IPA_FUSED] != 0; }  // A/B switch (0): three launches + flag kernel per round (the round-2 path)

// L = <a_L, G_R> + c_L Q + blind_L H and R = <a_R, G_L> + c_R Q + blind_R H (bullet.rs:83-97) over the ORIGINAL generators:
// generator j = p*n_cur + i belongs to L when i >= h (scalar a[i-h]*s[p]) and to R when i < h (scalar a[h+i]*s[p]), so
// each row has m = n0/2 + 2 columns with its own generator list: rows[r][q], idx_lr[r][q], q = p*h + (i mod h), then Q, H.
// Blocks 0..nb-1 fill the generator columns; the extra last block computes c_L, c_R (LDS reduce) and the trailing columns.
// When `fold` is set the vectors are those of the previous round and the fold of bullet.rs:105-109 by (u, u^-1) is applied
// on the way: every block derives the folded entries it needs on the fly (2 multiplications each), and the extra block
// materialises a', b', s' in the ping-pong buffers before it forms c_L, c_R from them — the fold needs no launch of its own.
__device__ __forceinline__ Fq ipa_a(const Fq* __restrict__ a, size_t x, size_t n_cur, int fold, const Fq& u, const Fq& u_inv) {
  Fq v = ld_fq(a + x);
  return fold ? fq_add(fq_mul(v, u), fq_mul(u_inv, ld_fq(a + n_cur + x))) : v;  // a' = a_L u + u^-1 a_R
}
__device__ __forceinline__ Fq ipa_b(const Fq* __restrict__ b, size_t x, size_t n_cur, int fold, const Fq& u, const Fq& u_inv) {
  Fq v = ld_fq(b + x);
  return fold ? fq_add(fq_mul(v, u_inv), fq_mul(u, ld_fq(b + n_cur + x))) : v;  // b' = b_L u^-1 + u b_R
}
__device__ __forceinline__ Fq ipa_s(const Fq* __restrict__ s, size_t p, int fold, const Fq& u, const Fq& u_inv) {
  return fold ? fq_mul(ld_fq(s + p / 2), (p & 1) ? u : u_inv) : ld_fq(s + p);  // s'[2p] = s[p] u^-1, s'[2p+1] = s[p] u
}
// 512-thread blocks: the extra block folds and multiplies up to 2048 entries, 4 + 2 per thread instead of 8 + 4 (1024 threads would spill).
__global__ void __launch_bounds__(512) k_ipa_prepare(const Fq* __restrict__ a, const Fq* __restrict__ b, const Fq* __restrict__ s, size_t n_cur,
                                                     size_t n0, size_t g_off, uint32_t q_idx, uint32_t h_idx, Fq q_scale, Fq blind_L, Fq blind_R,
                                                     Fq* __restrict__ rows, uint32_t* __restrict__ idx_lr, int fold, Fq u, Fq u_inv,
                                                     Fq* __restrict__ a_new, Fq* __restrict__ b_new, Fq* __restrict__ s_new) { SP_FG_PRIO();
  __shared__ Fq sm[2][512];
  size_t h = n_cur / 2, m = n0 / 2 + 2;
  if (blockIdx.x + 1 < gridDim.x) {
    size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n0) return;
    size_t p = j / n_cur, i = j % n_cur;
    Fq sp_ = ipa_s(s, p, fold, u, u_inv);
    bool is_l = i >= h;
    size_t q = p * h + (is_l ? i - h : i);
    Fq v = fq_mul(ipa_a(a, is_l ? i - h : h + i, n_cur, fold, u, u_inv), sp_);  // a_L[i-h] * G_R[i-h]  |  a_R[i] * G_L[i]
    st_fq(rows + (is_l ? 0 : m) + q, v);
    idx_lr[(is_l ? 0 : m) + q] = (uint32_t)(g_off + j);
    return;
  }
  const Fq *ac = a, *bc = b;
  if (fold) {
    for (size_t i = threadIdx.x; i < n_cur; i += blockDim.x) {
      st_fq(a_new + i, ipa_a(a, i, n_cur, 1, u, u_inv));
      st_fq(b_new + i, ipa_b(b, i, n_cur, 1, u, u_inv));
    }
    for (size_t p = threadIdx.x; p < n0 / n_cur; p += blockDim.x) st_fq(s_new + p, ipa_s(s, p, 1, u, u_inv));
    __syncthreads();  // this block reads back what it has just written
    ac = a_new; bc = b_new;
  }
  Fq c[2] = {fq_zero(), fq_zero()};
  for (size_t i = threadIdx.x; i < h; i += blockDim.x) {
    c[0] = fq_add(c[0], fq_mul(ld_fq(ac + i), ld_fq(bc + h + i)));  // c_L = <a_L, b_R>  (bullet.rs:80)
    c[1] = fq_add(c[1], fq_mul(ld_fq(ac + h + i), ld_fq(bc + i)));  // c_R = <a_R, b_L>  (bullet.rs:81)
  }
  sm[0][threadIdx.x] = c[0];
  sm[1][threadIdx.x] = c[1];
  __syncthreads();
  for (unsigned st = blockDim.x / 2; st > 0; st >>= 1) {
    if (threadIdx.x < st) {
      sm[0][threadIdx.x] = fq_add(sm[0][threadIdx.x], sm[0][threadIdx.x + st]);
      sm[1][threadIdx.x] = fq_add(sm[1][threadIdx.x], sm[1][threadIdx.x + st]);
    }
    __syncthreads();
  }
  c[0] = sm[0][0];
  c[1] = sm[1][0];
  if (threadIdx.x == 0) {
    size_t t = n0 / 2;
    st_fq(rows + t, fq_mul(c[0], q_scale));
    st_fq(rows + t + 1, blind_L);
    st_fq(rows + m + t, fq_mul(c[1], q_scale));
    st_fq(rows + m + t + 1, blind_R);
    idx_lr[t] = q_idx; idx_lr[t + 1] = h_idx;
    idx_lr[m + t] = q_idx; idx_lr[m + t + 1] = h_idx;
  }
}
// bullet.rs:105-109 (a, b) and the coefficient update replacing the G fold
__global__ void __launch_bounds__(256) k_ipa_fold(Fq* __restrict__ a, Fq* __restrict__ b, const Fq* __restrict__ s, Fq* __restrict__ s_new,
                                                  size_t n_cur, size_t n0, Fq u, Fq u_inv) { SP_FG_PRIO();
  size_t h = n_cur / 2;
  for (size_t i = threadIdx.x; i < h; i += 256) {
    Fq al = ld_fq(a + i), ar = ld_fq(a + h + i), bl = ld_fq(b + i), br = ld_fq(b + h + i);
    st_fq(a + i, fq_add(fq_mul(al, u), fq_mul(u_inv, ar)));
    st_fq(b + i, fq_add(fq_mul(bl, u_inv), fq_mul(u, br)));
  }
  size_t np = n0 / n_cur;
  for (size_t p = threadIdx.x; p < np; p += 256) {
    Fq sp_ = ld_fq(s + p);
    st_fq(s_new + 2 * p, fq_mul(sp_, u_inv));
    st_fq(s_new + 2 * p + 1, fq_mul(sp_, u));
  }
}
__global__ void __launch_bounds__(256) k_ipa_ghat_row(const Fq* __restrict__ s, size_t n0, Fq d, Fq r, Fq* __restrict__ row) { SP_FG_PRIO();
  for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < n0; j += (size_t)gridDim.x * blockDim.x) st_fq(row + j, fq_mul(ld_fq(s + j), d));
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    st_fq(row + n0, fq_zero());
    st_fq(row + n0 + 1, r);
  }
}

// The end of the argument in one launch: a_hat, b_hat (the vectors at length 1, bullet.rs:121-131) into the host-mapped page and
// the scalar row d * s (+ 0 * Q + r * H) of the commitment under g_hat = sum_p s[p] G[p] (nizk/mod.rs:496-501), with the last
// recorded fold applied on the way (a' = a_0 u + u^-1 a_1, s'[2p] = s[p] u^-1, s'[2p+1] = s[p] u) instead of a launch of its own.
__global__ void __launch_bounds__(256) k_ipa_finish_rows(const Fq* __restrict__ a, const Fq* __restrict__ b, const Fq* __restrict__ s, size_t n0, int fold, Fq u, Fq u_inv,
                                                         Fq d, Fq r, Fq* __restrict__ row, Fq* __restrict__ ab_out) { SP_FG_PRIO();
  for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < n0; j += (size_t)gridDim.x * blockDim.x) st_fq(row + j, fq_mul(ipa_s(s, j, fold, u, u_inv), d));
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    st_fq(row + n0, fq_zero());
    st_fq(row + n0 + 1, r);
    st_fq(ab_out, ipa_a(a, 0, 1, fold, u, u_inv));
    st_fq(ab_out + 1, ipa_b(b, 0, 1, fold, u, u_inv));
  }
}
// a_hat, b_hat (the vectors at length 1) into the host-mapped result page
__global__ void k_ipa_heads(const Fq* __restrict__ a, const Fq* __restrict__ b, Fq* __restrict__ out, DoneSig sig) { SP_FG_PRIO();
  if (threadIdx.x == 0) { st_fq(out, ld_fq(a)); st_fq(out + 1, ld_fq(b)); }
  signal_done(sig);
}

// One launch sets an opening up (was: four copies, a memset and the c_L/c_R kernel): a (a device table, or a pinned host copy the
// kernel reads over PCIe) and b (pinned host copy) into the working buffers, the generator index list, s = [1], the row tickets, and
// the first round's c_L = <a_L, b_R>, c_R = <a_R, b_L> (bullet.rs:80-81) as one partial pair per block of 512 index pairs.
__global__ void __launch_bounds__(256) k_ipa_init(const Fq* __restrict__ a_src, const Fq* __restrict__ b_src, size_t n, size_t g_off, uint32_t q_idx,
                                                  uint32_t h_idx, Fq* __restrict__ a, Fq* __restrict__ b, Fq* __restrict__ s, uint32_t* __restrict__ idx,
                                                  uint32_t* __restrict__ counters, Fq* __restrict__ c0_out) { SP_FG_PRIO();
  __shared__ Fq sm[256];
  const size_t h = n / 2;
  Fq c[2] = {fq_zero(), fq_zero()};
  for (size_t i = (size_t)blockIdx.x * 512 + threadIdx.x, k = 0; k < 2 && i < h; k++, i += 256) {
    const Fq al = ld_fq(a_src + i), ah = ld_fq(a_src + h + i), bl = ld_fq(b_src + i), bh = ld_fq(b_src + h + i);
    st_fq(a + i, al); st_fq(a + h + i, ah); st_fq(b + i, bl); st_fq(b + h + i, bh);
    idx[i] = (uint32_t)(g_off + i); idx[h + i] = (uint32_t)(g_off + h + i);
    c[0] = fq_add(c[0], fq_mul(al, bh));
    c[1] = fq_add(c[1], fq_mul(ah, bl));
  }
  if (blockIdx.x == 0) {
    if (threadIdx.x < 16) counters[threadIdx.x] = 0;
    if (threadIdx.x == 0) {
      if (n == 1) { st_fq(a, ld_fq(a_src)); st_fq(b, ld_fq(b_src)); idx[0] = (uint32_t)g_off; }
      idx[n] = q_idx; idx[n + 1] = h_idx;
      st_fq(s, fq_one());
    }
  }
  if (c0_out) {
    block_sum_fq<2>(c, sm);
    if (threadIdx.x == 0) { st_fq(c0_out + 2 * blockIdx.x, c[0]); st_fq(c0_out + 2 * blockIdx.x + 1, c[1]); }
  }
}

static Fq limbs(const uint64_t* p) {
  Fq x;
  memcpy(x.l, p, 32);
  return x;
}

extern "C" {

void sp_ipa_free(sp_ipa* ipa) {
  if (!ipa) return;
  if (ipa->pre) (void)sig_wait(ipa->ctx, ipa->pre_sig);  // a prelaunched round nobody collected still reads the buffers
  pool_release(ipa->ctx, ipa->base, ipa->bytes);  // one allocation backs a, b, a2, b2, s, s2, rows, idx
  delete ipa->last_args;
  delete ipa;
}
// a: host vector, or (a_dev != nullptr) the first n entries of a device table. commit_a != nullptr: also returns
// commit(a, blind_a) over the same generators (the Cx of DotProductProofLog::prove, nizk/mod.rs:469) computed from the
// device copy — the vector is uploaded once for both uses.
static int32_t ipa_begin(sp_ctx* c, const sp_gens* g, size_t g_off, size_t n, size_t q_idx, size_t h_idx, const uint64_t* q_scale, const uint64_t* a,
                         const sp_table* a_dev, const uint64_t* b, const uint64_t* blind_a, uint8_t* commit_a, sp_ipa** out) {
  if (!c || !g || (!a && !a_dev) || !b || !out || !is_pow2(n) || g_off + n > g->n || q_idx >= g->n || h_idx >= g->n) return SP_EINVAL;
  if (a_dev && a_dev->cap < n) return SP_EINVAL;
  if (commit_a && !blind_a) return SP_EINVAL;
  HIPCHK(hipSetDevice(c->dev));
  sp_ipa* ipa = new (std::nothrow) sp_ipa();
  if (!ipa) return SP_ENOMEM;
  ipa->ctx = c; ipa->g = g; ipa->n0 = ipa->n_cur = n; ipa->g_off = g_off; ipa->q_idx = q_idx; ipa->h_idx = h_idx;
  ipa->q_scale = q_scale ? limbs(q_scale) : fq_one();
  size_t fq_count = 6 * n + 2 * (n + 2);
  uint8_t* base = nullptr;
  ipa->bytes = 32 * fq_count + 4 * (n + 2) + 4 * (n + 4) + 64;
  ipa->a = nullptr;
  ipa->base = nullptr;
  int32_t prc = pool_alloc(c, ipa->bytes, (void**)&base);
  if (prc != SP_OK) { delete ipa; return prc; }
  ipa->base = base;
  ipa->a = (Fq*)base; ipa->b = ipa->a + n; ipa->a2 = ipa->b + n; ipa->b2 = ipa->a2 + n; ipa->s = ipa->b2 + n; ipa->s2 = ipa->s + n;
  ipa->rows = ipa->s2 + n;
  ipa->fold_pending = false;
  ipa->idx = (uint32_t*)(ipa->rows + 2 * (n + 2));
  ipa->idx_lr = ipa->idx + (n + 2);
  ipa->counters = ipa->idx_lr + (n + 4);
  ipa->have_c0 = ipa->have_dots = false;
  // b (and a host-side a) go through the pinned staging buffer, which the set-up kernel reads directly; [0, 4096) of it belongs to stage_in
  const size_t stage_off = 4096;
  int32_t rc = ensure_pinned(c, stage_off + 64 * n);
  if (rc != SP_OK) { sp_ipa_free(ipa); return rc; }
  Fq* pb = (Fq*)(c->pinned + stage_off);
  memcpy(pb, b, 32 * n);
  const Fq* a_src = a_dev ? (const Fq*)a_dev->d : pb + n;
  if (!a_dev) memcpy(pb + n, a, 32 * n);
  // c_L, c_R of the first round ride along with whatever this call waits for: into the result page, clear of the row sums
  Fq* c0_dst = (Fq*)(hres(c) + HOST_SUM_BYTES);  // [30720, 31744): between the partial sums and the row sums of the last KiB
  const unsigned nblk0 = ipa_c0_blocks(n) ? ipa_c0_blocks(n) : 1;
  const bool want_c0 = ipa_fused(c) && n >= 2 && !c->device_encode && nblk0 <= 16;  // 16 partial pairs = 1 KiB of the page
  {
    ProfScope ps(c, PF_IPA, 128.0 * (double)n);
    hipLaunchKernelGGL(k_ipa_init, dim3(nblk0), dim3(256), 0, c->stream, a_src, (const Fq*)pb, n, g_off, (uint32_t)q_idx, (uint32_t)h_idx, ipa->a, ipa->b, ipa->s,
                       ipa->idx, ipa->counters, want_c0 ? c0_dst : (Fq*)nullptr);
  }
  if (hipGetLastError() != hipSuccess) { sp_ipa_free(ipa); return SP_EHIP; }
  if (commit_a) {  // waits for the stream: the host buffers above are released too
    rc = ensure_dstage(c, 32);
    if (rc == SP_OK) rc = stage_in(c, 0, blind_a, 32);
    if (rc == SP_OK) rc = msm_launch(c, g, ipa->a, n, 1, n, g_off, nullptr, (const Fq*)c->dstage, h_idx, commit_a);
  } else if (hipStreamSynchronize(c->stream) != hipSuccess) {
    rc = SP_EHIP;
  }
  if (rc != SP_OK) { sp_ipa_free(ipa); return rc; }
  if (want_c0) {
    ipa->c0[0] = ipa->c0[1] = fq_zero();
    for (unsigned k = 0; k < ipa_c0_blocks(n); k++) { ipa->c0[0] = fq_add(ipa->c0[0], c0_dst[2 * k]); ipa->c0[1] = fq_add(ipa->c0[1], c0_dst[2 * k + 1]); }
    ipa->have_c0 = true;
  }
  *out = ipa;
  return SP_OK;
}
int32_t sp_ipa_begin(sp_ctx* c, const sp_gens* g, size_t g_off, size_t n, size_t q_idx, size_t h_idx, const uint64_t q_scale[4],
                     const uint64_t* a, const uint64_t* b, sp_ipa** out) {
  if (!q_scale || !a) return SP_EINVAL;
  return ipa_begin(c, g, g_off, n, q_idx, h_idx, q_scale, a, nullptr, b, nullptr, nullptr, out);
}
int32_t sp_ipa_begin_dev(sp_ctx* c, const sp_gens* g, size_t g_off, size_t n, size_t q_idx, size_t h_idx, const sp_table* a_dev, const uint64_t* b,
                         const uint64_t blind_a[4], uint8_t commit_a[32], sp_ipa** out) {
  if (!a_dev || !blind_a || !commit_a) return SP_EINVAL;
  return ipa_begin(c, g, g_off, n, q_idx, h_idx, nullptr, nullptr, a_dev, b, blind_a, commit_a, out);
}
int32_t sp_ipa_set_scale(sp_ipa* ipa, const uint64_t q_scale[4]) {
  if (!ipa || !q_scale) return SP_EINVAL;
  ipa->q_scale = limbs(q_scale);
  return SP_OK;
}
// One launch per round: the kernel looks up and sums the generator columns of both rows and prepares the next round, while
// the calling thread forms c_L Q + blind_L H and c_R Q + blind_R H (two terms each) from its host-side window tables.
// first half of a one-launch round: c_L, c_R from what the previous launch left behind, and the launch itself
static int32_t ipa_round_start(sp_ipa* ipa, Fq* cL_out, Fq* cR_out, DoneSig* sig_out, unsigned* nd_out) {
  sp_ctx* c = ipa->ctx;
  Fq cL, cR;
  if (ipa->fold_pending && ipa->have_dots) {
    const Fq u2 = fq_mul(ipa->fu, ipa->fu), ui2 = fq_mul(ipa->fu_inv, ipa->fu_inv);
    const Fq* d = ipa->dots;
    cL = fq_add(fq_add(d[0], fq_mul(u2, d[1])), fq_add(fq_mul(ui2, d[2]), d[3]));
    cR = fq_add(fq_add(d[4], fq_mul(u2, d[5])), fq_add(fq_mul(ui2, d[6]), d[7]));
  } else if (!ipa->fold_pending && ipa->have_c0) {
    cL = ipa->c0[0]; cR = ipa->c0[1];
  } else {
    return SP_EINVAL;  // caller falls back to the three-launch path
  }
  ipa->have_c0 = ipa->have_dots = false;
  IpaRoundArgs A;
  A.a = ipa->a; A.b = ipa->b; A.s = ipa->s; A.a_new = ipa->a2; A.b_new = ipa->b2; A.s_new = ipa->s2;
  A.n_cur = ipa->n_cur; A.n0 = ipa->n0; A.g_off = ipa->g_off;
  A.fold = ipa->fold_pending ? 1 : 0;
  A.u = ipa->fu; A.u_inv = ipa->fu_inv;
  A.u2 = fq_mul(ipa->fu, ipa->fu); A.u2_inv = fq_mul(ipa->fu_inv, ipa->fu_inv);
  A.counters = ipa->counters;
  SPCHK(ipa_round_launch(c, ipa->g, &A, sig_out, 0));
  if (!ipa->last_args) ipa->last_args = new (std::nothrow) IpaRoundArgs();
  if (!ipa->last_args) return SP_ENOMEM;
  *ipa->last_args = A;
  if (ipa->fold_pending) {  // the kernel leaves the folded vectors in the ping-pong buffers
    std::swap(ipa->a, ipa->a2); std::swap(ipa->b, ipa->b2); std::swap(ipa->s, ipa->s2);
    ipa->fold_pending = false;
  }
  *cL_out = cL; *cR_out = cR; *nd_out = A.nd;
  return SP_OK;
}
static bool ipa_round_fusable(const sp_ipa* ipa) {
  const sp_ctx* c = ipa->ctx;
  return ipa_fused(c) && !c->device_encode && ipa->n_cur <= 16384 && ((ipa->fold_pending && ipa->have_dots) || (!ipa->fold_pending && ipa->have_c0));
}
static int32_t ipa_round_fused(sp_ipa* ipa, const uint64_t blind_L[4], const uint64_t blind_R[4], uint8_t L_out[32], uint8_t R_out[32]) {
  sp_ctx* c = ipa->ctx;
  Fq cL, cR;
  DoneSig sig;
  unsigned nd = 0;
  if (ipa->pre) {
    cL = ipa->pre_cL; cR = ipa->pre_cR; sig = ipa->pre_sig; nd = ipa->pre_nd;
    ipa->pre = false;
  } else {
    SPCHK(ipa_round_start(ipa, &cL, &cR, &sig, &nd));
  }
  // meanwhile, on this core: the two-term tails of both rows
  const uint32_t qh[2] = {(uint32_t)ipa->q_idx, (uint32_t)ipa->h_idx};
  uint64_t S[8];
  sp_host_point tails[2];
  int32_t hrc = SP_OK;
  for (int r = 0; r < 2 && hrc == SP_OK; r++) {
    Fq cq = fq_mul(r == 0 ? cL : cR, ipa->q_scale);
    memcpy(S, cq.l, 32);
    memcpy(S + 4, r == 0 ? blind_L : blind_R, 32);
    hrc = sp_host_commit_point(ipa->g, qh, 2, S, &tails[r]);
  }
  SPCHK(sig_wait(c, sig));
  if (hipGetLastError() != hipSuccess) return SP_EHIP;
  SPCHK(hrc);
  const Pt* sums = (const Pt*)hres(c);
  if (fp_is_zero(sums[0].Z) || fp_is_zero(sums[1].Z)) {
    // no valid point has Z = 0: an addition of the two-multiplication tree met one of its exceptional pairs (core.hip, pt10_tree_quad_ded:
    // a generator list that repeats a point). The round is run again with the unified formula: same inputs, same outputs otherwise.
    if (!c->opt.v[OPT_IPA_RERUN_EXCEPTIONAL]) return SP_EHIP;  // test-only: shows that a test input reached this path
    DoneSig sig2;
    SPCHK(ipa_round_launch(c, ipa->g, ipa->last_args, &sig2, 1));
    SPCHK(sig_wait(c, sig2));
    if (hipGetLastError() != hipSuccess) return SP_EHIP;
    if (fp_is_zero(sums[0].Z) || fp_is_zero(sums[1].Z)) return SP_EHIP;
  }
  Pt lr[2];
  for (int r = 0; r < 2; r++) {
    Pt tail;
    memcpy(&tail, &tails[r], sizeof(Pt));
    lr[r] = pt_add(sums[r], tail);
  }
  uint8_t enc[64];
  pt_compress_many(lr, 2, enc);  // L and R together: two interleaved inverse-square-root chains (curve.hpp)
  memcpy(L_out, enc, 32); memcpy(R_out, enc + 32, 32);
  if (ipa->n_cur == 2) {
    const Fq* fin = (const Fq*)(hres(c) + 1024) + 8;
    ipa->fin_a[0] = fin[0]; ipa->fin_a[1] = fin[1]; ipa->fin_b[0] = fin[2]; ipa->fin_b[1] = fin[3];
    memcpy(ipa->fin_raw, sums, 2 * sizeof(Pt));
    ipa->have_fin = true;
  }
  if (ipa->n_cur >= 4) {
    const Fq* dp = (const Fq*)(hres(c) + 1024);
    for (int k = 0; k < 8; k++) {
      Fq acc = dp[k];
      for (unsigned b = 1; b < nd; b++) acc = fq_add(acc, dp[(size_t)b * 8 + k]);
      ipa->dots[k] = acc;
    }
    ipa->have_dots = true;
  }
  return SP_OK;
}
// The kernel of the next round depends on neither the round's blinds nor the scale of Q (those enter the two-term tails the calling
// thread adds): it can be put in flight before the caller has them — DotProductProofLog::prove launches the first round, then absorbs
// Cx, Cy and the 4096 scalars of `a` (0.15 ms of Keccak) and draws r. No other call on the context until sp_ipa_round_lr (the kernel
// writes the result page). Does nothing when the round would not take the one-launch path.
int32_t sp_ipa_round_prelaunch(sp_ipa* ipa) {
  if (!ipa || ipa->n_cur < 2) return SP_EINVAL;
  HIPCHK(hipSetDevice(ipa->ctx->dev));
  if (ipa->pre || !ipa_round_fusable(ipa)) return SP_OK;
  SPCHK(ipa_round_start(ipa, &ipa->pre_cL, &ipa->pre_cR, &ipa->pre_sig, &ipa->pre_nd));
  ipa->pre = true;
  return SP_OK;
}
int32_t sp_ipa_round_lr(sp_ipa* ipa, const uint64_t blind_L[4], const uint64_t blind_R[4], uint8_t L_out[32], uint8_t R_out[32]) {
  if (!ipa || !blind_L || !blind_R || !L_out || !R_out || ipa->n_cur < 2) return SP_EINVAL;
  sp_ctx* c = ipa->ctx;
  HIPCHK(hipSetDevice(c->dev));
  if (ipa->pre || ipa_round_fusable(ipa)) return ipa_round_fused(ipa, blind_L, blind_R, L_out, R_out);
  ipa->have_c0 = ipa->have_dots = false;
  {
    ProfScope ps(c, PF_IPA, 32.0 * 4 * (double)ipa->n0);
    hipLaunchKernelGGL(k_ipa_prepare, dim3((unsigned)((ipa->n0 + 511) / 512 + 1)), dim3(512), 0, c->stream, (const Fq*)ipa->a, (const Fq*)ipa->b, (const Fq*)ipa->s, ipa->n_cur,
                       ipa->n0, ipa->g_off, (uint32_t)ipa->q_idx, (uint32_t)ipa->h_idx, ipa->q_scale, limbs(blind_L), limbs(blind_R), ipa->rows,
                       ipa->idx_lr, ipa->fold_pending ? 1 : 0, ipa->fu, ipa->fu_inv, ipa->a2, ipa->b2, ipa->s2);
  }
  if (ipa->fold_pending) {  // the kernel left the folded vectors in the ping-pong buffers
    std::swap(ipa->a, ipa->a2); std::swap(ipa->b, ipa->b2); std::swap(ipa->s, ipa->s2);
    ipa->fold_pending = false;
  }
  uint8_t lr[64];
  size_t m = ipa->n0 / 2 + 2;
  SPCHK(msm_launch(c, ipa->g, ipa->rows, m, 2, m, 0, ipa->idx_lr, nullptr, 0, lr, m));
  memcpy(L_out, lr, 32);
  memcpy(R_out, lr + 32, 32);
  return SP_OK;
}
// apply a recorded fold now (the vectors are needed by something other than the next round's prepare kernel)
static int32_t ipa_flush_fold(sp_ipa* ipa) {
  if (!ipa->fold_pending) return SP_OK;
  sp_ctx* c = ipa->ctx;
  {
    ProfScope ps(c, PF_IPA, 32.0 * 6 * (double)ipa->n_cur);
    hipLaunchKernelGGL(k_ipa_fold, dim3(1), dim3(256), 0, c->stream, ipa->a, ipa->b, (const Fq*)ipa->s, ipa->s2, 2 * ipa->n_cur, ipa->n0, ipa->fu,
                       ipa->fu_inv);
  }
  std::swap(ipa->s, ipa->s2);
  ipa->fold_pending = false;
  ipa->have_dots = false;  // they described the vectors before this fold
  return hipGetLastError() == hipSuccess ? SP_OK : SP_EHIP;
}
int32_t sp_ipa_round_fold(sp_ipa* ipa, const uint64_t u[4], const uint64_t u_inv[4]) {
  if (!ipa || !u || !u_inv || ipa->n_cur < 2) return SP_EINVAL;
  HIPCHK(hipSetDevice(ipa->ctx->dev));
  SPCHK(ipa_flush_fold(ipa));  // at most one fold is deferred
  ipa->fu = limbs(u);
  ipa->fu_inv = limbs(u_inv);
  ipa->fold_pending = true;
  ipa->n_cur /= 2;
  return SP_OK;
}
int32_t sp_ipa_finish(sp_ipa* ipa, uint64_t a_hat[4], uint64_t b_hat[4], uint8_t* g_hat) {
  if (!ipa || !a_hat || !b_hat || ipa->n_cur != 1) return SP_EINVAL;
  sp_ctx* c = ipa->ctx;
  HIPCHK(hipSetDevice(c->dev));
  SPCHK(ipa_flush_fold(ipa));
  DoneSig sig = sig_make(c, 1);
  hipLaunchKernelGGL(k_ipa_heads, dim3(1), dim3(64), 0, c->stream, (const Fq*)ipa->a, (const Fq*)ipa->b, (Fq*)hres(c), sig);
  SPCHK(sig_wait(c, sig));  // one trip for both
  memcpy(a_hat, hres(c), 32);
  memcpy(b_hat, hres(c) + 32, 32);
  if (g_hat) SPCHK(msm_launch(c, ipa->g, ipa->s, ipa->n0, 1, ipa->n0, ipa->g_off, nullptr, nullptr, 0, g_hat));
  return hipGetLastError() == hipSuccess ? SP_OK : SP_EHIP;
}
// sp_ipa_finish and sp_ipa_commit_ghat in one trip: delta = commit(d, r) under {g_hat, h} does not depend on a_hat, b_hat
// (nizk/mod.rs:498-503 computes y_hat from them only afterwards), so its launches go out first and the two scalars ride along.
// k1 * P1 + k2 * P2 for two points that are NOT fixed generators (host core; Straus with plain 4-bit windows: 252 doublings, <= 128
// additions, ~35 us). The one place the prover multiplies a variable base: see sp_ipa_finish_commit.
static Pt pt_var_msm2(const Pt& P1, const Fq& k1_mont, const Pt& P2, const Fq& k2_mont) {
  const Fq k1 = fq_from_mont(k1_mont), k2 = fq_from_mont(k2_mont);
  Pt T1[16], T2[16];
  T1[0] = T2[0] = pt_identity();
  T1[1] = P1; T2[1] = P2;
  for (int i = 2; i < 16; i++) {
    T1[i] = (i & 1) ? pt_add(T1[i - 1], P1) : pt_dbl(T1[i / 2]);
    T2[i] = (i & 1) ? pt_add(T2[i - 1], P2) : pt_dbl(T2[i / 2]);
  }
  Pt acc = pt_identity();
  bool started = false;
  for (int w = 63; w >= 0; w--) {
    if (started) { acc = pt_dbl(pt_dbl(pt_dbl(pt_dbl(acc)))); }
    const unsigned n1 = (unsigned)(k1.l[w / 16] >> (4 * (w % 16))) & 15, n2 = (unsigned)(k2.l[w / 16] >> (4 * (w % 16))) & 15;
    if (n1) { acc = pt_add(acc, T1[n1]); started = true; }
    if (n2) { acc = pt_add(acc, T2[n2]); started = true; }
  }
  return acc;
}
// The end of the argument on the calling thread's core (round 4). After the last round the reference holds two folded generators G'_L, G'_R and
// forms g_hat = u^-1 G'_L + u G'_R (bullet.rs:108), then commits delta = d g_hat + r_delta h (nizk/mod.rs:496-501). Here the folded generators
// are never built — but the last round's row sums ARE multiples of them: raw[0] = a'[0] G'_R, raw[1] = a'[1] G'_L. So
//     delta = (d u / a'[0]) raw[0] + (d u^-1 / a'[1]) raw[1] + r_delta h,     a_hat = a'[0] u + u^-1 a'[1],   b_hat = b'[0] u^-1 + u b'[1],
// a two-point multi-scalar multiplication over points the host already holds: ~45 us of host arithmetic instead of a scalar-row kernel plus a
// 78 000-lookup commitment kernel and their trip (~140 us), four times per SNARK proof. Same group element, same bytes
// (tests/test_gpu_large.py compares with the device path). Falls back to the device when a'[0] or a'[1] is zero (no such multiple),
// when the round ran unfused, or with SPARTAN_IPA_FINISH_DEVICE=1 (the A/B switch).
static bool ipa_finish_on_host(sp_ipa* ipa, const uint64_t d[4], const uint64_t r[4], uint64_t a_hat[4], uint64_t b_hat[4], uint8_t delta_out[32]) {
  const bool off = ipa->ctx->opt.v[OPT_IPA_FINISH_DEVICE] != 0;
  if (off || !ipa->have_fin || !ipa->fold_pending || ipa->ctx->device_encode) return false;
  const Fq &a0 = ipa->fin_a[0], &a1 = ipa->fin_a[1], &b0 = ipa->fin_b[0], &b1 = ipa->fin_b[1], &u = ipa->fu, &ui = ipa->fu_inv;
  if (fq_is_zero(a0) || fq_is_zero(a1)) return false;
  Fq dd, rr;
  memcpy(dd.l, d, 32); memcpy(rr.l, r, 32);
  const Fq inv01 = fq_invert(fq_mul(a0, a1));               // one inversion for both (the values derive from the witness: the constant chain)
  const Fq inv0 = fq_mul(inv01, a1), inv1 = fq_mul(inv01, a0);
  const Fq k0 = fq_mul(fq_mul(dd, u), inv0), k1 = fq_mul(fq_mul(dd, ui), inv1);
  Pt acc = pt_var_msm2(ipa->fin_raw[0], k0, ipa->fin_raw[1], k1);
  const uint32_t hidx[1] = {(uint32_t)ipa->h_idx};
  sp_host_point tail;
  if (sp_host_commit_point(ipa->g, hidx, 1, rr.l, &tail) != SP_OK) return false;
  Pt th;
  memcpy(&th, &tail, sizeof(Pt));
  pt_compress(pt_add(acc, th), delta_out);
  const Fq ah = fq_add(fq_mul(a0, u), fq_mul(ui, a1)), bh = fq_add(fq_mul(b0, ui), fq_mul(u, b1));
  memcpy(a_hat, ah.l, 32);
  memcpy(b_hat, bh.l, 32);
  return true;
}
// Device-free form of the two-point variable-base multiplication for tests (the arithmetic behind ipa_finish_on_host): k1 P1 + k2 P2, encoded.
int32_t sp_host_msm2_probe(const uint8_t p1[32], const uint64_t k1[4], const uint8_t p2[32], const uint64_t k2[4], uint8_t out[32]) {
  if (!p1 || !k1 || !p2 || !k2 || !out) return SP_EINVAL;
  Pt P1, P2;
  if (!pt_decompress(p1, &P1) || !pt_decompress(p2, &P2)) return SP_EPOINT;
  Fq a, b;
  memcpy(a.l, k1, 32); memcpy(b.l, k2, 32);
  pt_compress(pt_var_msm2(P1, a, P2, b), out);
  return SP_OK;
}
int32_t sp_ipa_finish_commit(sp_ipa* ipa, const uint64_t d[4], const uint64_t r[4], uint64_t a_hat[4], uint64_t b_hat[4], uint8_t delta_out[32]) {
  if (!ipa || !d || !r || !a_hat || !b_hat || !delta_out || ipa->n_cur != 1) return SP_EINVAL;
  if (ipa_finish_on_host(ipa, d, r, a_hat, b_hat, delta_out)) return SP_OK;
  sp_ctx* c = ipa->ctx;
  HIPCHK(hipSetDevice(c->dev));
  // between the partial sums (< HOST_SUM_BYTES) and the row sums (last KiB) of the result page: msm_launch leaves it alone
  Fq* ab = (Fq*)(hres(c) + HOST_SUM_BYTES);
  {
    ProfScope ps(c, PF_IPA, 64.0 * (double)ipa->n0);
    hipLaunchKernelGGL(k_ipa_finish_rows, dim3((unsigned)grid_for(ipa->n0, 64)), dim3(256), 0, c->stream, (const Fq*)ipa->a, (const Fq*)ipa->b, (const Fq*)ipa->s, ipa->n0,
                       ipa->fold_pending ? 1 : 0, ipa->fu, ipa->fu_inv, limbs(d), limbs(r), ipa->rows, ab);
  }
  // (the recorded fold stays recorded: the vectors behind the handle are not needed again — sp_ipa_free follows)
  SPCHK(msm_launch(c, ipa->g, ipa->rows, ipa->n0 + 2, 1, ipa->n0 + 2, 0, ipa->idx, nullptr, 0, delta_out));  // waits for the stream
  memcpy(a_hat, ab, 32);
  memcpy(b_hat, ab + 1, 32);
  return SP_OK;
}
int32_t sp_ipa_commit_ghat(sp_ipa* ipa, const uint64_t d[4], const uint64_t r[4], uint8_t out[32]) {
  if (!ipa || !d || !r || !out || ipa->n_cur != 1) return SP_EINVAL;
  sp_ctx* c = ipa->ctx;
  HIPCHK(hipSetDevice(c->dev));
  SPCHK(ipa_flush_fold(ipa));
  {
    ProfScope ps(c, PF_IPA, 64.0 * (double)ipa->n0);
    hipLaunchKernelGGL(k_ipa_ghat_row, dim3((unsigned)grid_for(ipa->n0, 64)), dim3(256), 0, c->stream, (const Fq*)ipa->s, ipa->n0, limbs(d),
                       limbs(r), ipa->rows);
  }
  return msm_launch(c, ipa->g, ipa->rows, ipa->n0 + 2, 1, ipa->n0 + 2, 0, ipa->idx, nullptr, 0, out);
}

}  // extern "C"
