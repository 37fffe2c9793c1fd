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
//
// MIXED WIDTHS (round 6). A scalar is < 2^253 and the signed recoding needs one spare bit: the windows of a scalar have to cover 254 bits,
// no more. Uniform c-bit windows cover ceil(254 / c) * c — 266 bits at c = 14: nineteen additions where eighteen windows of 14.1 bits
// would do. So a geometry is (nwin, c, nwide): the nwin - nwide LOW windows are c bits wide (tent = 2^(c-1) entries each), the nwide TOP
// windows c + 1 bits (2 * tent entries), with (nwin - nwide) * c + nwide * (c + 1) >= 254. For a given number of windows the narrowest such
// split is c = floor(254 / nwin), nwide = 254 - nwin * c (msm_geom_windows): 18 windows = 16 x 14 + 2 x 15 bits at 21.0 MB per generator
// (uniform 14-bit: 19 windows, 19.9 MB); 17 = 1 x 14 + 16 x 15 (34.6 MB; uniform 15-bit: 35.7); 20 = 6 x 12 + 14 x 13 (8.9 MB; uniform 12:
// 22 windows, 5.8 MB). The uniform geometries (nwide = 0) remain: forced widths (option msm.wbits), the LDS-staged form, the host tables.
struct MsmGeom {
  int wbits, nwin, tent;   // width c of the narrow (low) windows, windows per scalar, entries of a narrow window = 2^(c-1)
  int nwide;               // top windows of width c + 1 (2 * tent entries each); 0: uniform
  size_t pt_entries;       // (nwin + nwide) * tent
};
SP_HD MsmGeom msm_geom(int wbits) {  // uniform c-bit windows
  MsmGeom g;
  g.wbits = wbits;
  g.nwin = (254 + wbits - 1) / wbits;
  g.tent = 1 << (wbits - 1);
  g.nwide = 0;
  g.pt_entries = (size_t)g.nwin * (size_t)g.tent;
  return g;
}
SP_HD MsmGeom msm_geom_windows(int nwin) {  // exactly nwin windows over 254 bits, as narrow as that allows
  MsmGeom g;
  g.nwin = nwin;
  g.wbits = 254 / nwin;
  g.nwide = 254 - nwin * g.wbits;
  g.tent = 1 << (g.wbits - 1);
  g.pt_entries = (size_t)(g.nwin + g.nwide) * (size_t)g.tent;
  return g;
}
SP_HD int msm_n0(const MsmGeom& g) { return g.nwin - g.nwide; }                                    // first wide window
SP_HD int msm_wbits_of(const MsmGeom& g, int w) { return g.wbits + (w >= msm_n0(g) ? 1 : 0); }      // width of window w
SP_HD int msm_bitpos(const MsmGeom& g, int w) { return w * g.wbits + (w > msm_n0(g) ? w - msm_n0(g) : 0); }  // its lowest bit
SP_HD size_t msm_woff(const MsmGeom& g, int w) { return (size_t)(w + (w > msm_n0(g) ? w - msm_n0(g) : 0)) * (size_t)g.tent; }  // entries of a point before window w
// index of entry (point pt, window w, magnitude m in 1..2^(width-1))
SP_HD size_t msm_tidx(const MsmGeom& g, size_t pt, int w, int m) { return pt * g.pt_entries + msm_woff(g, w) + (size_t)(m - 1); }
SP_HD size_t msm_tidx(size_t pt, int w, int m) { return (pt * MSM_NWIN + (size_t)w) * MSM_TENT + (size_t)(m - 1); }

// raw `wbits`-bit field of a canonical 256-bit integer starting at bit `bit`
SP_HD uint32_t msm_field_at(const Fq& s, int bit, int wbits) {
  int k = bit >> 6, sh = bit & 63;
  if (k > 3) return 0;
  uint64_t x = s.l[k] >> sh;
  if (sh + wbits > 64 && k < 3) x |= s.l[k + 1] << (64 - sh);
  return (uint32_t)(x & ((1u << wbits) - 1));
}
SP_HD uint32_t msm_field(const Fq& s, int w, int wbits) { return msm_field_at(s, w * wbits, wbits); }  // uniform windows
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
    const int c = msm_wbits_of(g, k);
    d = (int)msm_field_at(s, msm_bitpos(g, k), c) + carry;
    carry = d >= (1 << (c - 1));
    d -= carry << c;
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
  // digit array; the loop stays rolled so the register budget allows a third wave per SIMD). Window k is msm_wbits_of(g, k) bits wide:
  // the top nwide windows one bit more than the rest (mixed widths, above).
  const int n0 = msm_n0(g);
  int carry = 0;
  int wnext = 0;        // the next window to take from the shifting scalar
  auto next_digit = [&](int& dn, uint32_t& mn) {
    const int c = g.wbits + (wnext >= n0 ? 1 : 0);
    dn = (int)(s.l[0] & ((1u << c) - 1)) + carry;
    carry = dn >= (1 << (c - 1));
    dn -= carry << c;
    mn = (uint32_t)(dn < 0 ? -dn : dn);
    s.l[0] = (s.l[0] >> c) | (s.l[1] << (64 - c));
    s.l[1] = (s.l[1] >> c) | (s.l[2] << (64 - c));
    s.l[2] = (s.l[2] >> c) | (s.l[3] << (64 - c));
    s.l[3] >>= c;
    wnext++;
  };
  auto entry_of = [&](int w, uint32_t m) { const int wc = w < g.nwin ? w : g.nwin - 1; return base + msm_woff(g, wc) + (m ? m - 1 : 0); };
  int d; uint32_t m;
  next_digit(d, m);
  bool ng = d < 0;
  MsmEntry cur = msm_load(entry_of(0, m));
  if (PF2) {
  // two table entries in flight (PMC: waves of the row MSM wait on memory 43 % of their cycles at every window width — the gathers
  // are latency-, not bandwidth- or translation-bound: profiles/r3_pmc_msm_translation_fabric.txt)
  int d1; uint32_t m1;
  next_digit(d1, m1);
  MsmEntry nx1 = msm_load(entry_of(1, m1));
  bool ng1 = d1 < 0;
#pragma unroll 1
  for (int w = 0; w < g.nwin; w++) {
    int d2; uint32_t m2;
    next_digit(d2, m2);  // window w + 2 (zero past the top: s < 2^253)
    // short scalars (SNARK::encode commits addresses and timestamps, a few bits each; sparse_mlpoly.rs:483-503): nothing is left above the
    // digits in hand — no gathers for the upper windows. (The lanes of a wave are rows of one kind of value, so they leave together.)
    if ((m | m1 | m2) == 0 && (s.l[0] | s.l[1] | s.l[2] | s.l[3]) == 0 && carry == 0) break;
    MsmEntry nx2 = msm_load(entry_of(w + 2, w + 2 < g.nwin ? m2 : 0));
    if (m != 0) acc = pt_madd(acc, msm_entry_niels(cur), ng);
    cur = nx1; m = m1; ng = ng1;
    nx1 = nx2; m1 = m2; ng1 = d2 < 0;
  }
  } else {
#pragma unroll 1
  for (int w = 0; w < g.nwin; w++) {
    int dn; uint32_t mn;
    next_digit(dn, mn);  // window w + 1 (zero past the top: s < 2^253)
    if ((m | mn) == 0 && (s.l[0] | s.l[1] | s.l[2] | s.l[3]) == 0 && carry == 0) break;  // short scalar: no gathers for the upper windows
    MsmEntry nxt = msm_load(entry_of(w + 1, w + 1 < g.nwin ? mn : 0));
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
