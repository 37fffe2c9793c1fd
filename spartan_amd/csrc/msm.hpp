// spartan_amd: fixed-base window-table MSM building blocks (host+device so the CPU tests exercise them).
//
// Every multi-scalar multiplication on the prover path is over *fixed* public generators
// (MultiCommitGens, src/commitments.rs:15-33): the batched row commitments of
// DensePolynomial::commit_inner (src/dense_mlpoly.rs:164-177), the 2..5-point Sigma-protocol commitments
// (src/nizk/mod.rs) and — after re-expressing the folded generators of BulletReductionProof::prove
// (src/nizk/bullet.rs:83-109) over the original ones — the inner-product argument too. So the device keeps,
// per generator P and per signed c-bit window w, the 2^(c-1) multiples k*2^(cw)*P in affine Niels form; a
// 253-bit scalar costs at most ceil(254/c) mixed additions and no doublings (c = SP_MSM_WBITS).
#pragma once
#include "curve.hpp"

namespace sp {

#ifndef SP_MSM_WBITS
#define SP_MSM_WBITS 13  // measured on MI355X at 2^20, ms per proof: c = 8 / 10 / 12 / 13 -> 88.5 / 84.3 / 82.1 / 82.4 early in round 1; with the
                         // final code 12 / 13 / 14 -> 50.4 / 49.5 / 48.9. 13 = 8.2 MiB per generator (42 GB at 2^20, 118 GB at 2^22); 14 would not fit 2^22
#endif
constexpr int MSM_WBITS = SP_MSM_WBITS;                        // signed window width c
constexpr int MSM_NWIN = (254 + MSM_WBITS - 1) / MSM_WBITS;     // windows covering a 253-bit scalar plus the recoding carry
constexpr int MSM_TENT = 1 << (MSM_WBITS - 1);                 // entries per (point, window): magnitudes 1..2^(c-1)
constexpr size_t MSM_PT_ENTRIES = (size_t)MSM_NWIN * MSM_TENT;
static_assert(MSM_WBITS >= 4 && MSM_WBITS <= 15, "window width");

// The window width is a property of a generator set (sp_gens), chosen when its tables are built: the widest c whose tables
// fit the HBM budget (core.hip, choose_wbits) — 15 bits (17 additions per scalar) for the generators of a 2^20 instance,
// 13 for a 2^22 one, less for larger sets. The constants above are the default geometry (and the one the host-side
// arithmetic tests use).
struct MsmGeom {
  int wbits, nwin, tent;   // signed window width c, windows per scalar, entries per (point, window) = 2^(c-1)
  size_t pt_entries;       // nwin * tent
};
SP_HD MsmGeom msm_geom(int wbits) {
  MsmGeom g;
  g.wbits = wbits;
  g.nwin = (254 + wbits - 1) / wbits;
  g.tent = 1 << (wbits - 1);
  g.pt_entries = (size_t)g.nwin * (size_t)g.tent;
  return g;
}
// index of entry (point pt, window w, magnitude m in 1..tent)
SP_HD size_t msm_tidx(const MsmGeom& g, size_t pt, int w, int m) { return (pt * g.nwin + (size_t)w) * g.tent + (size_t)(m - 1); }
SP_HD size_t msm_tidx(size_t pt, int w, int m) { return (pt * MSM_NWIN + (size_t)w) * MSM_TENT + (size_t)(m - 1); }

// raw c-bit field of a canonical 256-bit integer at window w
SP_HD uint32_t msm_field(const Fq& s, int w, int wbits) {
  int bit = w * wbits, k = bit >> 6, sh = bit & 63;
  if (k > 3) return 0;
  uint64_t x = s.l[k] >> sh;
  if (sh + wbits > 64 && k < 3) x |= s.l[k + 1] << (64 - sh);
  return (uint32_t)(x & ((1u << wbits) - 1));
}
SP_HD uint32_t msm_field(const Fq& s, int w) { return msm_field(s, w, MSM_WBITS); }
// signed recoding of a canonical scalar (< 2^253): digits d_w in [-2^(c-1), 2^(c-1) - 1], sum d_w 2^(c w) = s.
// mag[w] = |d_w| (0..2^(c-1)), neg bit w = (d_w < 0).
SP_HD void msm_recode(const Fq& s, uint16_t mag[MSM_NWIN], uint32_t* neg) {
  int carry = 0;
  uint32_t ng = 0;
#pragma unroll
  for (int w = 0; w < MSM_NWIN; w++) {
    int d = (int)msm_field(s, w) + carry;
    carry = d >= MSM_TENT;
    d -= carry << MSM_WBITS;
    mag[w] = (uint16_t)(d < 0 ? -d : d);
    ng |= (uint32_t)(d < 0) << w;
  }
  *neg = ng;
}
// digit of window w only (latency-bound kernels: one thread per window)
SP_HD int msm_digit(const Fq& s, int w, const MsmGeom& g) {
  int carry = 0, d = 0;
  for (int k = 0; k <= w; k++) {  // the carry into window w depends on all lower windows
    d = (int)msm_field(s, k, g.wbits) + carry;
    carry = d >= g.tent;
    d -= carry << g.wbits;
  }
  return d;
}
SP_HD int msm_digit(const Fq& s, int w) { return msm_digit(s, w, msm_geom(MSM_WBITS)); }

// acc += s * P[pt] using P's window table. `s` is the reference's Montgomery-form Scalar. The table entry of the
// next window is requested before the current mixed addition so the gather latency overlaps the 7 multiplications.
// PF2: two table entries in flight instead of one (26 more registers: the foreground row MSM has them, the 1024-thread background form has not)
typedef Niels MsmEntry;
SP_HD const Niels& msm_entry_niels(const MsmEntry& e) { return e; }
// the 96 bytes of an entry, field by field (with SP_NIELS_ALIGN=128 a struct copy would also move the 32 bytes of padding)
SP_HD MsmEntry msm_load(const MsmEntry* p) {
  MsmEntry e;
  e.yp = p->yp; e.ym = p->ym; e.t2d = p->t2d;
  return e;
}
template <bool PF2>
SP_HD void msm_accumulate_t(Pt& acc, const Fq& s_mont, const Niels* __restrict__ table, size_t pt, const MsmGeom& g) {
  if (fq_is_zero(s_mont)) return;
  Fq s = fq_from_mont(s_mont);  // canonical integer < q < 2^253 (scalar/mod.rs:32-36 does the same for dalek)
  const MsmEntry* base = reinterpret_cast<const MsmEntry*>(table) + pt * g.pt_entries;
  // digits are produced on the fly by shifting the scalar down one window per step (8 live registers instead of a
  // digit array; the loop stays rolled so the register budget allows a third wave per SIMD)
  const int c = g.wbits;
  const uint32_t mask = (1u << c) - 1;
  int d = (int)(s.l[0] & mask);
  int carry = d >= g.tent;
  d -= carry << c;
  uint32_t m = (uint32_t)(d < 0 ? -d : d);
  bool ng = d < 0;
  MsmEntry cur = msm_load(base + (m ? m - 1 : 0));
  if (PF2) {
  // two table entries in flight (PMC: waves of the row MSM wait on memory 43 % of their cycles at every window width — the gathers
  // are latency-, not bandwidth- or translation-bound: profiles/r3_pmc_msm_translation_fabric.txt)
  auto next_digit = [&](int& dn, uint32_t& mn) {
    s.l[0] = (s.l[0] >> c) | (s.l[1] << (64 - c));
    s.l[1] = (s.l[1] >> c) | (s.l[2] << (64 - c));
    s.l[2] = (s.l[2] >> c) | (s.l[3] << (64 - c));
    s.l[3] >>= c;
    dn = (int)(s.l[0] & mask) + carry;
    carry = dn >= g.tent;
    dn -= carry << c;
    mn = (uint32_t)(dn < 0 ? -dn : dn);
  };
  int d1; uint32_t m1;
  next_digit(d1, m1);
  MsmEntry nx1 = msm_load(base + (size_t)(g.nwin > 1 ? 1 : 0) * g.tent + (m1 ? m1 - 1 : 0));
  bool ng1 = d1 < 0;
#pragma unroll 1
  for (int w = 0; w < g.nwin; w++) {
    int d2; uint32_t m2;
    next_digit(d2, m2);  // window w + 2 (zero past the top: s < 2^253)
    // short scalars (SNARK::encode commits addresses and timestamps, a few bits each; sparse_mlpoly.rs:483-503): nothing is left above the
    // digits in hand — no gathers for the upper windows. (The lanes of a wave are rows of one kind of value, so they leave together.)
    if ((m | m1 | m2) == 0 && (s.l[0] | s.l[1] | s.l[2] | s.l[3]) == 0 && carry == 0) break;
    int w2 = (w + 2 < g.nwin) ? w + 2 : g.nwin - 1;
    MsmEntry nx2 = msm_load(base + (size_t)w2 * g.tent + (m2 ? m2 - 1 : 0));
    if (m != 0) acc = pt_madd(acc, msm_entry_niels(cur), ng);
    cur = nx1; m = m1; ng = ng1;
    nx1 = nx2; m1 = m2; ng1 = d2 < 0;
  }
  } else {
#pragma unroll 1
  for (int w = 0; w < g.nwin; w++) {
    s.l[0] = (s.l[0] >> c) | (s.l[1] << (64 - c));
    s.l[1] = (s.l[1] >> c) | (s.l[2] << (64 - c));
    s.l[2] = (s.l[2] >> c) | (s.l[3] << (64 - c));
    s.l[3] >>= c;
    int dn = (int)(s.l[0] & mask) + carry;  // window w+1 (zero past the top: s < 2^253)
    carry = dn >= g.tent;
    dn -= carry << c;
    uint32_t mn = (uint32_t)(dn < 0 ? -dn : dn);
    if ((m | mn) == 0 && (s.l[0] | s.l[1] | s.l[2] | s.l[3]) == 0 && carry == 0) break;  // short scalar: no gathers for the upper windows
    int wn = (w + 1 < g.nwin) ? w + 1 : w;
    MsmEntry nxt = msm_load(base + (size_t)wn * g.tent + (mn ? mn - 1 : 0));
    if (m != 0) acc = pt_madd(acc, msm_entry_niels(cur), ng);
    cur = nxt;
    m = mn;
    ng = dn < 0;
  }
  }
}
SP_HD void msm_accumulate(Pt& acc, const Fq& s_mont, const Niels* __restrict__ table, size_t pt, const MsmGeom& g) { msm_accumulate_t<false>(acc, s_mont, table, pt, g); }
SP_HD void msm_accumulate(Pt& acc, const Fq& s_mont, const Niels* __restrict__ table, size_t pt) { msm_accumulate(acc, s_mont, table, pt, msm_geom(MSM_WBITS)); }

}  // namespace sp
