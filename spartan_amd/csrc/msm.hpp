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

// acc += s * P[pt] using P's window table. `s` is the reference's Montgomery-form Scalar.
SP_HD void msm_accumulate(Pt& acc, const Fq& s_mont, const Niels* __restrict__ table, size_t pt) {
  if (fq_is_zero(s_mont)) return;
  Fq s = fq_from_mont(s_mont);  // canonical integer < q < 2^253 (scalar/mod.rs:32-36 does the same for dalek)
  int carry = 0;
#pragma unroll 1
  for (int w = 0; w < MSM_NWIN; w++) {
    int d = (int)((s.l[w >> 3] >> ((w & 7) * 8)) & 0xff) + carry;
    carry = d > 127;
    d -= carry << 8;  // d in [-128, 127]
    if (d != 0) {
      int m = d < 0 ? -d : d;
      acc = pt_madd(acc, table[msm_tidx(pt, w, m)], d < 0);
    }
  }
}

}  // namespace sp
