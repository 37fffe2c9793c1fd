// spartan_amd host driver: inversion of a PUBLIC F_q element (a Fiat-Shamir challenge) by Bernstein-Yang division steps.
//
// BulletReductionProof::prove inverts every round's challenge (bullet.rs:99-100: `u_inv = u.invert().unwrap()`); the reference's
// Scalar::invert is the fixed addition chain a^(q-2) (scalar/ristretto255.rs:541-595: 253 squarings + ~64 multiplications). The
// value is all that matters for the proof bytes, and the inversion sits between two device launches of the inner-product
// argument (45 per SNARK proof), so the proving thread computes it the fastest way a CPU core can: "safegcd"
// (D. J. Bernstein, B.-Y. Yang, "Fast constant-time gcd computation and modular inversion", 2019), 62 division steps per batch on
// 64-bit words, the batch's 2x2 transition matrix applied to (f, g) and, modulo q, to (d, e) — ~10 batches for a 253-bit
// modulus. Variable time (trailing zeros are skipped, up to 6 low bits cancelled per iteration): ONLY for public inputs. Secrets
// (none are inverted on the prover path today) keep fq_invert.
// Measured (bench/host_arith_probe.cc): 10.4 us -> 1.5 us on the build container's Xeon, 5.8 us -> ~1 us on the GPU box's EPYC.
#pragma once
#include <cstdint>
#include <cstring>

#include "../csrc/field.hpp"

namespace spz {
namespace inv62 {

typedef __int128 i128;
struct S62 { int64_t v[5]; };  // signed 62-bit limbs: value = sum v[i] * 2^(62 i); limbs 0..3 in [0, 2^62), limb 4 carries the sign
struct Trans { int64_t u, v, q, r; };
constexpr int64_t M62 = (int64_t)(~0ULL >> 2);

inline S62 from_limbs(const uint64_t l[4]) {
  S62 r;
  r.v[0] = (int64_t)(l[0] & (uint64_t)M62);
  r.v[1] = (int64_t)(((l[0] >> 62) | (l[1] << 2)) & (uint64_t)M62);
  r.v[2] = (int64_t)(((l[1] >> 60) | (l[2] << 4)) & (uint64_t)M62);
  r.v[3] = (int64_t)(((l[2] >> 58) | (l[3] << 6)) & (uint64_t)M62);
  r.v[4] = (int64_t)(l[3] >> 56);
  return r;
}
inline void to_limbs(const S62& a, uint64_t l[4]) {  // a in [0, 2^256)
  const uint64_t a0 = (uint64_t)a.v[0], a1 = (uint64_t)a.v[1], a2 = (uint64_t)a.v[2], a3 = (uint64_t)a.v[3], a4 = (uint64_t)a.v[4];
  l[0] = a0 | (a1 << 62);
  l[1] = (a1 >> 2) | (a2 << 60);
  l[2] = (a2 >> 4) | (a3 << 58);
  l[3] = (a3 >> 6) | (a4 << 56);
}
struct ModInfo {
  S62 m;          // the modulus q
  uint64_t minv;  // q^-1 mod 2^62
};
inline const ModInfo& modinfo() {
  static const ModInfo mi = [] {
    ModInfo x;
    const uint64_t ql[4] = {SP_Q0, SP_Q1, SP_Q2, SP_Q3};
    x.m = from_limbs(ql);
    uint64_t inv = ql[0];  // Newton: correct to 3 bits, each step doubles
    for (int i = 0; i < 6; i++) inv *= 2 - ql[0] * inv;
    x.minv = inv & (uint64_t)M62;
    return x;
  }();
  return mi;
}

// 62 division steps on the low words of (f, g); returns the new eta = -delta and the transition matrix t with
// 2^62 * (f', g') = t * (f, g). f is odd throughout.
inline int64_t divsteps_62_var(int64_t eta, uint64_t f0, uint64_t g0, Trans* t) {
  uint64_t u = 1, v = 0, q = 0, r = 1;
  uint64_t f = f0, g = g0, m;
  uint32_t w;
  int i = 62, limit, zeros;
  for (;;) {
    zeros = __builtin_ctzll(g | (~0ULL << i));  // sentinel: never more than the steps that are left
    g >>= zeros;
    u <<= zeros;
    v <<= zeros;
    eta -= zeros;
    i -= zeros;
    if (i == 0) break;
    // g is odd. delta > 0 (eta < 0): (f, g) <- (g, -f) first; then a multiple of f cancels the low bits of g — as many bits as
    // steps would take the "delta <= 0, g odd" branch in a row (eta + 1 of them), at most 6 (4) by the inverse formula used
    if (eta < 0) {
      uint64_t tmp;
      eta = -eta;
      tmp = f; f = g; g = 0 - tmp;
      tmp = u; u = q; q = 0 - tmp;
      tmp = v; v = r; r = 0 - tmp;
      limit = ((int)eta + 1) > i ? i : ((int)eta + 1);
      m = (~0ULL >> (64 - limit)) & 63U;
      w = (uint32_t)((f * g * (f * f - 2)) & m);  // -g / f mod 2^6: f (f^2 - 2) = -1/f mod 64 for odd f
    } else {
      limit = ((int)eta + 1) > i ? i : ((int)eta + 1);
      m = (~0ULL >> (64 - limit)) & 15U;
      w = (uint32_t)(f + (((f + 1) & 4) << 1));  // 1/f mod 16
      w = (uint32_t)((0 - (uint64_t)w * g) & m);
    }
    g += f * w;
    q += u * w;
    r += v * w;
  }
  t->u = (int64_t)u; t->v = (int64_t)v; t->q = (int64_t)q; t->r = (int64_t)r;
  return eta;
}
// (f, g) <- t * (f, g) / 2^62 (exact)
inline void update_fg(S62& f, S62& g, const Trans& t) {
  const int64_t u = t.u, v = t.v, q = t.q, r = t.r;
  i128 cf = (i128)u * f.v[0] + (i128)v * g.v[0];
  i128 cg = (i128)q * f.v[0] + (i128)r * g.v[0];
  cf >>= 62; cg >>= 62;  // the low 62 bits are zero by construction
  for (int i = 1; i < 5; i++) {
    cf += (i128)u * f.v[i] + (i128)v * g.v[i];
    cg += (i128)q * f.v[i] + (i128)r * g.v[i];
    f.v[i - 1] = (int64_t)((uint64_t)cf & (uint64_t)M62); cf >>= 62;
    g.v[i - 1] = (int64_t)((uint64_t)cg & (uint64_t)M62); cg >>= 62;
  }
  f.v[4] = (int64_t)cf;
  g.v[4] = (int64_t)cg;
}
// (d, e) <- t * (d, e) / 2^62 mod q; d, e stay in (-2q, q)
inline void update_de(S62& d, S62& e, const Trans& t, const ModInfo& mi) {
  const int64_t u = t.u, v = t.v, q = t.q, r = t.r;
  const int64_t sd = d.v[4] >> 63, se = e.v[4] >> 63;  // sign masks
  int64_t md = (u & sd) + (v & se), me = (q & sd) + (r & se);  // multiples of q that bring t * (d, e) back into range
  i128 cd = (i128)u * d.v[0] + (i128)v * e.v[0];
  i128 ce = (i128)q * d.v[0] + (i128)r * e.v[0];
  // ... and that make the low 62 bits vanish
  md -= (int64_t)((mi.minv * (uint64_t)cd + (uint64_t)md) & (uint64_t)M62);
  me -= (int64_t)((mi.minv * (uint64_t)ce + (uint64_t)me) & (uint64_t)M62);
  cd += (i128)mi.m.v[0] * md;
  ce += (i128)mi.m.v[0] * me;
  cd >>= 62; ce >>= 62;
  for (int i = 1; i < 5; i++) {
    cd += (i128)u * d.v[i] + (i128)v * e.v[i] + (i128)mi.m.v[i] * md;
    ce += (i128)q * d.v[i] + (i128)r * e.v[i] + (i128)mi.m.v[i] * me;
    d.v[i - 1] = (int64_t)((uint64_t)cd & (uint64_t)M62); cd >>= 62;
    e.v[i - 1] = (int64_t)((uint64_t)ce & (uint64_t)M62); ce >>= 62;
  }
  d.v[4] = (int64_t)cd;
  e.v[4] = (int64_t)ce;
}
// r in (-2q, q), negated if sign < 0, brought to [0, q)
inline void normalize(S62& r, int64_t sign, const ModInfo& mi) {
  auto carry = [&] {
    for (int i = 0; i < 4; i++) { r.v[i + 1] += r.v[i] >> 62; r.v[i] &= M62; }
  };
  int64_t add = r.v[4] >> 63;  // negative: add q
  for (int i = 0; i < 5; i++) r.v[i] += mi.m.v[i] & add;
  const int64_t neg = sign >> 63;
  for (int i = 0; i < 5; i++) r.v[i] = (r.v[i] ^ neg) - neg;
  carry();
  add = r.v[4] >> 63;  // in (-q, q) now
  for (int i = 0; i < 5; i++) r.v[i] += mi.m.v[i] & add;
  carry();
}
// x^-1 mod q for an integer 0 < x < q (plain, not Montgomery form); 0 -> 0
inline void modinv_var(const uint64_t x[4], uint64_t out[4]) {
  const ModInfo& mi = modinfo();
  S62 d = {{0, 0, 0, 0, 0}}, e = {{1, 0, 0, 0, 0}};
  S62 f = mi.m, g = from_limbs(x);
  int64_t eta = -1;
  while ((g.v[0] | g.v[1] | g.v[2] | g.v[3] | g.v[4]) != 0) {
    Trans t;
    eta = divsteps_62_var(eta, (uint64_t)f.v[0], (uint64_t)g.v[0], &t);
    update_de(d, e, t, mi);
    update_fg(f, g, t);
  }
  // f = +-gcd = +-1 and d * x = f (mod q)
  normalize(d, f.v[4], mi);
  to_limbs(d, out);
}
}  // namespace inv62

// Montgomery form in, Montgomery form out: (aR)^-1 = a^-1 R^-1, times R^3 under a Montgomery multiplication = a^-1 R. Same value as
// sp::fq_invert (0 -> 0 there as well: 0^(q-2) = 0).
inline sp::Fq fq_invert_vartime(const sp::Fq& a_mont) {
  sp::Fq t;
  inv62::modinv_var(a_mont.l, t.l);
  return sp::fq_mul(t, sp::fq_R3());
}

}  // namespace spz
