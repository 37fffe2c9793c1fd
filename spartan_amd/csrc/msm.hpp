// spartan_amd: fixed-base window-table MSM building blocks (host+device so the CPU tests exercise them).
//
// Every multi-scalar multiplication on the prover path is over *fixed* public generators
// (MultiCommitGens, src/commitments.rs:15-33): the batched row commitments of
// DensePolynomial::commit_inner (src/dense_mlpoly.rs:164-177), the 2..5-point Sigma-protocol commitments
// (src/nizk/mod.rs) and — after re-expressing the folded generators of BulletReductionProof::prove
// (src/nizk/bullet.rs:83-109) over the original ones — the inner-product argument too. So the device keeps,
// per generator P and per signed 8-bit window w, the 128 multiples k*2^(8w)*P (k=1..128) in affine Niels
// form; a 253-bit scalar costs at most 32 mixed additions and no doublings.
#pragma once
#include "curve.hpp"

namespace sp {

constexpr int MSM_WBITS = 8;
constexpr int MSM_NWIN = 32;                      // 32 * 8 = 256 bits >= 253 + carry
constexpr int MSM_TENT = 1 << (MSM_WBITS - 1);    // 128 entries per (point, window)
constexpr size_t MSM_PT_ENTRIES = (size_t)MSM_NWIN * MSM_TENT;

// index of entry (point pt, window w, magnitude m in 1..128)
SP_HD size_t msm_tidx(size_t pt, int w, int m) { return (pt * MSM_NWIN + (size_t)w) * MSM_TENT + (size_t)(m - 1); }

// signed 8-bit recoding of a canonical scalar (< 2^253): digits d_w in [-128, 127], sum d_w 2^(8w) = s.
// mag[w] = |d_w| (0..128) packed 4 per word, neg bit w = (d_w < 0).
SP_HD void msm_recode(const Fq& s, uint32_t mag[8], uint32_t* neg) {
  int carry = 0;
  uint32_t ng = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) mag[k] = 0;
#pragma unroll
  for (int w = 0; w < MSM_NWIN; w++) {
    int d = (int)((s.l[w >> 3] >> ((w & 7) * 8)) & 0xff) + carry;
    carry = d > 127;
    d -= carry << 8;
    uint32_t m = (uint32_t)(d < 0 ? -d : d);
    mag[w >> 2] |= m << ((w & 3) * 8);
    ng |= (uint32_t)(d < 0) << w;
  }
  *neg = ng;
}

// acc += s * P[pt] using P's window table. `s` is the reference's Montgomery-form Scalar. The table entry of the
// next window is requested before the current mixed addition so the gather latency overlaps the 7 multiplications.
SP_HD void msm_accumulate(Pt& acc, const Fq& s_mont, const Niels* __restrict__ table, size_t pt) {
  if (fq_is_zero(s_mont)) return;
  Fq s = fq_from_mont(s_mont);  // canonical integer < q < 2^253 (scalar/mod.rs:32-36 does the same for dalek)
  uint32_t mag[8], neg;
  msm_recode(s, mag, &neg);
  const Niels* base = table + pt * MSM_PT_ENTRIES;
  uint32_t m = mag[0] & 0xff;
  Niels cur = base[m ? m - 1 : 0];
#pragma unroll 1
  for (int w = 0; w < MSM_NWIN; w++) {
    uint32_t mn = (w + 1 < MSM_NWIN) ? (mag[(w + 1) >> 2] >> (((w + 1) & 3) * 8)) & 0xff : 0;
    Niels nxt = base[(size_t)((w + 1 < MSM_NWIN) ? w + 1 : w) * MSM_TENT + (mn ? mn - 1 : 0)];
    if (m != 0) acc = pt_madd(acc, cur, (neg >> w) & 1);
    cur = nxt;
    m = mn;
  }
}

}  // namespace sp
