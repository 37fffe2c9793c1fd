// TEST INFRASTRUCTURE (not part of the product): compiles spartan_amd/csrc/{field,curve,msm}.hpp for the
// host so the exact arithmetic source that runs on gfx950 can be checked against the oracle on a CPU-only
// machine. Never loaded by spartan_amd itself.
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../spartan_amd/csrc/msm.hpp"
#include "../../spartan_amd/csrc/fe10.hpp"

using namespace sp;
static Fq L(const uint64_t* p) { Fq x; memcpy(x.l, p, 32); return x; }
static void O(const Fq& x, uint64_t* o) { memcpy(o, x.l, 32); }

extern "C" {
void hc_fq_mul(const uint64_t* a, const uint64_t* b, uint64_t* o) { O(fq_mul(L(a), L(b)), o); }
void hc_fq_add(const uint64_t* a, const uint64_t* b, uint64_t* o) { O(fq_add(L(a), L(b)), o); }
void hc_fq_sub(const uint64_t* a, const uint64_t* b, uint64_t* o) { O(fq_sub(L(a), L(b)), o); }
void hc_fq_neg(const uint64_t* a, uint64_t* o) { O(fq_neg(L(a)), o); }
void hc_fq_invert(const uint64_t* a, uint64_t* o) { O(fq_invert(L(a)), o); }
void hc_fq_from_mont(const uint64_t* a, uint64_t* o) { O(fq_from_mont(L(a)), o); }
void hc_fq_from_u512(const uint64_t* w, uint64_t* o) { O(fq_from_u512(w), o); }
void hc_fq_from_u64(uint64_t v, uint64_t* o) { O(fq_from_u64(v), o); }

// Fp: bytes in (any 32 bytes, top bit masked) -> canonical bytes out
void hc_fp_mul(const uint8_t* a, const uint8_t* b, uint8_t* o) { fp_to_bytes(fp_mul(fp_from_bytes(a), fp_from_bytes(b)), o); }
void hc_fp_add(const uint8_t* a, const uint8_t* b, uint8_t* o) { fp_to_bytes(fp_add(fp_from_bytes(a), fp_from_bytes(b)), o); }
void hc_fp_sub(const uint8_t* a, const uint8_t* b, uint8_t* o) { fp_to_bytes(fp_sub(fp_from_bytes(a), fp_from_bytes(b)), o); }
void hc_fp_invert(const uint8_t* a, uint8_t* o) { fp_to_bytes(fp_invert(fp_from_bytes(a)), o); }
// raw 256-bit limbs (exercises weakly-reduced inputs >= p, >= 2^255)
void hc_fp_mul_raw(const uint64_t* a, const uint64_t* b, uint8_t* o) { Fp x, y; memcpy(x.v, a, 32); memcpy(y.v, b, 32); fp_to_bytes(fp_mul(x, y), o); }
void hc_fp_sqr_raw(const uint64_t* a, uint8_t* o) { Fp x; memcpy(x.v, a, 32); fp_to_bytes(fp_sqr(x), o); }
void hc_fe10_mul_raw(const uint64_t* a, const uint64_t* b, uint8_t* o) { Fp x, y; memcpy(x.v, a, 32); memcpy(y.v, b, 32); fp_to_bytes(fe10_to_fp(fe10_mul(fe10_from_fp(x), fe10_from_fp(y))), o); }
void hc_fe10_sqr_chain_raw(const uint64_t* a, int k, uint8_t* o) { Fp x; memcpy(x.v, a, 32); fp_to_bytes(fe10_to_fp(fe10_pow2k(fe10_from_fp(x), k)), o); }
void hc_fp_pow_p58_serial(const uint64_t* a, uint8_t* o) { Fp x; memcpy(x.v, a, 32); fp_to_bytes(fp_pow_p58_serial(x), o); }
void hc_fp_invert_serial(const uint64_t* a, uint8_t* o) { Fp x; memcpy(x.v, a, 32); fp_to_bytes(fp_invert_serial(x), o); }
void hc_fp_add_raw(const uint64_t* a, const uint64_t* b, uint8_t* o) { Fp x, y; memcpy(x.v, a, 32); memcpy(y.v, b, 32); fp_to_bytes(fp_add(x, y), o); }
void hc_fp_sub_raw(const uint64_t* a, const uint64_t* b, uint8_t* o) { Fp x, y; memcpy(x.v, a, 32); memcpy(y.v, b, 32); fp_to_bytes(fp_sub(x, y), o); }

int hc_pt_recompress(const uint8_t* in, uint8_t* out) { Pt p; if (!pt_decompress(in, &p)) return 0; pt_compress(p, out); return 1; }
void hc_pt_from_uniform(const uint8_t* in64, uint8_t* out) { pt_compress(pt_from_uniform_bytes(in64), out); }
int hc_pt_add(const uint8_t* a, const uint8_t* b, uint8_t* out) { Pt p, q; if (!pt_decompress(a, &p) || !pt_decompress(b, &q)) return 0; pt_compress(pt_add(p, q), out); return 1; }
// serial-chain (Fe10) forms: add two points and encode through Pt10
int hc_pt10_add_compress(const uint8_t* a, const uint8_t* b, uint8_t* out) { Pt p, q; if (!pt_decompress(a, &p) || !pt_decompress(b, &q)) return 0; pt10_compress(pt10_add(pt10_load(p), pt10_load(q)), out); return 1; }
int hc_pt10_sum_compress(const uint8_t* pts, size_t n, uint8_t* out) { Pt10 acc = pt10_identity(); for (size_t i = 0; i < n; i++) { Pt p; if (!pt_decompress(pts + 32 * i, &p)) return 0; acc = pt10_add(acc, pt10_load(p)); } pt10_compress(acc, out); return 1; }
// running sums P_i = pts[0] + ... + pts[i] (Z != 1), encoded N at a time (pt_compress_many) into out and one by one into out_single
int hc_pt_running_sums_compress_many(const uint8_t* pts, size_t n, uint8_t* out, uint8_t* out_single) {
  if (n == 0 || n > 64) return 0;
  Pt sums[64], acc = pt_identity();
  for (size_t i = 0; i < n; i++) { Pt p; if (!pt_decompress(pts + 32 * i, &p)) return 0; acc = pt_add(acc, p); sums[i] = acc; }
  pt_compress_many(sums, n, out);
  for (size_t i = 0; i < n; i++) pt_compress(sums[i], out_single + 32 * i);
  return 1;
}
int hc_pt_dbl(const uint8_t* a, uint8_t* out) { Pt p; if (!pt_decompress(a, &p)) return 0; pt_compress(pt_dbl(p), out); return 1; }

// fixed-base table MSM exactly as the device does it: build tables like k_table_build, accumulate like k_msm_rows
int hc_msm_fixed(const uint8_t* pts_comp, size_t n, const uint64_t* scalars, uint8_t* out) {
  std::vector<Niels> table(n * MSM_PT_ENTRIES);
  std::vector<Pt> mults(MSM_TENT);
  for (size_t i = 0; i < n; i++) {
    Pt P;
    if (!pt_decompress(pts_comp + 32 * i, &P)) return 0;
    Pt base = P;
    for (int w = 0; w < MSM_NWIN; w++) {
      // multiples 1..MSM_TENT with one inversion for the whole window (Montgomery's trick) — test helper only
      Pt acc = base;
      std::vector<Fp> pref(MSM_TENT);
      Fp run = fp_one();
      for (int m = 1; m <= MSM_TENT; m++) {
        mults[m - 1] = acc;
        pref[m - 1] = run;
        run = fp_mul(run, acc.Z);
        if (m < MSM_TENT) acc = pt_add(acc, base);
      }
      Fp inv = fp_invert(run);
      for (int m = MSM_TENT; m >= 1; m--) {
        Fp zinv = fp_mul(inv, pref[m - 1]);
        inv = fp_mul(inv, mults[m - 1].Z);
        table[msm_tidx(i, w, m)] = pt_to_niels(mults[m - 1], zinv);
      }
      for (int k = 0; k < MSM_WBITS; k++) base = pt_dbl(base);
    }
  }
  Pt acc = pt_identity();
  for (size_t i = 0; i < n; i++) msm_accumulate(acc, L(scalars + 4 * i), table.data(), i);
  pt_compress(acc, out);
  return 1;
}
// The same over a MIXED-width geometry (msm.hpp, msm_geom_windows: nwin windows over exactly 254 bits, the top ones one bit wider), two ways:
// out_stream = the shifting digit stream of the strip form (msm_accumulate_t, one and two entries in flight), out_digit = one lookup per
// (generator, window) with msm_digit / msm_tidx as the latency kernels do it. nwin = 0: uniform `wbits`-bit windows.
int hc_msm_fixed_geom(const uint8_t* pts_comp, size_t n, const uint64_t* scalars, int nwin, int wbits, uint8_t* out_stream, uint8_t* out_stream2, uint8_t* out_digit) {
  const MsmGeom g = nwin ? msm_geom_windows(nwin) : msm_geom(wbits);
  if (nwin && ((g.nwin - g.nwide) * g.wbits + g.nwide * (g.wbits + 1) != 254 || msm_bitpos(g, g.nwin - 1) + msm_wbits_of(g, g.nwin - 1) != 254)) return 0;
  std::vector<Niels> table(n * g.pt_entries);
  for (size_t i = 0; i < n; i++) {
    Pt P;
    if (!pt_decompress(pts_comp + 32 * i, &P)) return 0;
    for (int w = 0; w < g.nwin; w++) {
      Pt base = P;
      for (int k = 0; k < msm_bitpos(g, w); k++) base = pt_dbl(base);
      Pt acc = base;
      const int tent_w = 1 << (msm_wbits_of(g, w) - 1);
      for (int m = 1; m <= tent_w; m++) {
        table[msm_tidx(g, i, w, m)] = pt_to_niels(acc, fp_invert(acc.Z));
        acc = pt_add(acc, base);
      }
    }
  }
  Pt a1 = pt_identity(), a2 = pt_identity(), a3 = pt_identity();
  for (size_t i = 0; i < n; i++) {
    const Fq sm = L(scalars + 4 * i);
    msm_accumulate_t<false>(a1, sm, table.data(), i, g);
    msm_accumulate_t<true>(a2, sm, table.data(), i, g);
    const Fq s = fq_from_mont(sm);
    for (int w = 0; w < g.nwin; w++) {
      int d = msm_digit(s, w, g);
      if (d != 0) a3 = pt_madd(a3, table[msm_tidx(g, i, w, d < 0 ? -d : d)], d < 0);
    }
  }
  pt_compress(a1, out_stream); pt_compress(a2, out_stream2); pt_compress(a3, out_digit);
  return 1;
}
}
