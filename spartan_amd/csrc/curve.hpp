// spartan_amd: ristretto255 group arithmetic (host+device).
//
// libspartan's GroupElement is curve25519-dalek's RistrettoPoint (reference: src/group.rs:6-7); the only
// stable wire form is the 32-byte CompressedRistretto. This file implements, from RFC 9496 §4, what the
// prover path needs: decode (generators arrive compressed over the C ABI), encode (every commitment leaves
// compressed), the one-way map (MultiCommitGens::new = SHAKE256 stream -> from_uniform_bytes,
// src/commitments.rs:15-33), and twisted-Edwards (a=-1) extended-coordinate addition in the two forms the
// MSM kernels use: full add (add-2008-hwcd-3) and mixed add against a precomputed affine "Niels" entry
// (y+x, y-x, 2dxy) — 7 field multiplications.
#pragma once
#include "fe10.hpp"
#include "field.hpp"

namespace sp {

SP_HD Fp fp_D() { return Fp{{0x75eb4dca135978a3ULL, 0x00700a4d4141d8abULL, 0x8cc740797779e898ULL, 0x52036cee2b6ffe73ULL}}; }
SP_HD Fp fp_D2() { return Fp{{0xebd69b9426b2f159ULL, 0x00e0149a8283b156ULL, 0x198e80f2eef3d130ULL, 0x2406d9dc56dffce7ULL}}; }
SP_HD Fp fp_ONE_MINUS_D_SQ() { return Fp{{0xe27c09c1945fc176ULL, 0x2c81a138cd5e350fULL, 0x9994abddbe70dfe4ULL, 0x029072a8b2b3e0d7ULL}}; }
SP_HD Fp fp_D_MINUS_ONE_SQ() { return Fp{{0x31ad5aaa44ed4d20ULL, 0xd29e4a2cb01e1999ULL, 0x4cdcd32f529b4eebULL, 0x5968b37af66c2241ULL}}; }
SP_HD Fp fp_SQRT_M1() { return Fp{{0xc4ee1b274a0ea0b0ULL, 0x2f431806ad2fe478ULL, 0x2b4d00993dfbd7a7ULL, 0x2b8324804fc1df0bULL}}; }
SP_HD Fp fp_SQRT_AD_MINUS_ONE() { return Fp{{0x7e97f6a0497b2e1bULL, 0xaf9d8e0c1b7854bdULL, 0x0f3cfcc931f5d1fdULL, 0x376931bf2b8348acULL}}; }
SP_HD Fp fp_INVSQRT_A_MINUS_D() { return Fp{{0x99c8fdaa805d40eaULL, 0x9d2f16175a4172beULL, 0x16c27b91fe01d840ULL, 0x786c8905cfaffca2ULL}}; }

struct Pt {  // extended coordinates: x = X/Z, y = Y/Z, T = XY/Z
  Fp X, Y, Z, T;
};
#ifndef SP_NIELS_ALIGN
#define SP_NIELS_ALIGN 128
#endif
// affine point prepared for mixed addition: 96 bytes of values in a 128-byte slot, so that a table gather touches exactly ONE 128-byte
// line. Packed at a 96-byte stride (rounds 1-3; -DSP_NIELS_ALIGN=32 rebuilds that layout) half of the entries straddle two lines.
// Measured in round 4 (bench/gather_probe.hip, profiles/r4_gather_probe.txt: the MSM's access pattern with no arithmetic behind it): 23.5-27.3 G
// entries/s packed against 30.4-33.6 G/s line-aligned, at every table size from 8 to 110 GB (the size of the table set does not matter;
// only a 0.2 GB table, resident in the Infinity Cache, is faster); the 1280 x 4096 launch 6.06 -> 5.61 ms at 14-bit windows, 5.99 -> 5.32 at
// 15; a 2^20 proof 26.3 -> 25.2 ms in the same session (fewer line requests per addition also means less pressure on the latency-bound
// kernels that run next to the background commit). Round 2 had measured the opposite (3.05 -> 3.45 ms) with a struct copy that moved
// the 32 bytes of padding through the registers as well: msm_load (msm.hpp) reads the three fields.
struct alignas(SP_NIELS_ALIGN) Niels {
  Fp yp, ym, t2d;  // y+x, y-x, 2*d*x*y
};

// the same three values packed at 96 bytes: the entry format of the LDS-staged small-window MSM (msm_lds.hip), whose sub-tables are streamed
// into LDS whole (coalesced, no per-lane gather from HBM: no line to align to)
struct NielsP {
  Fp yp, ym, t2d;
};

SP_HD Pt pt_identity() { return Pt{fp_zero(), fp_one(), fp_one(), fp_zero()}; }

// RFC 9496 §4.2 SQRT_RATIO_M1
SP_HD bool fp_sqrt_ratio_m1(const Fp& u, const Fp& v, Fp* out) {
  Fp v3 = fp_mul(fp_sqr(v), v);
  Fp v7 = fp_mul(fp_sqr(v3), v);
#if defined(__HIP_DEVICE_COMPILE__)
  Fp r = fp_mul(fp_mul(u, v3), fp_pow_p58_serial(fp_mul(u, v7)));  // lone-wave chain: radix-2^25.5 ladder (fe10.hpp)
#else
  Fp r = fp_mul(fp_mul(u, v3), fp_pow_p58(fp_mul(u, v7)));  // host core: 64-bit multiplier, 4x64 ladder (~2-4 us)
#endif
  Fp check = fp_mul(v, fp_sqr(r));
  Fp neg_u = fp_neg(u);
  bool correct_sign = fp_eq(check, u);
  bool flipped = fp_eq(check, neg_u);
  bool flipped_i = fp_eq(check, fp_mul(neg_u, fp_SQRT_M1()));
  Fp r_i = fp_mul(r, fp_SQRT_M1());
  r = fp_select(r, r_i, flipped || flipped_i);
  *out = fp_abs(r);
  return correct_sign || flipped;
}

SP_HD Pt pt_add(const Pt& p, const Pt& q) {
  Fp A = fp_mul(fp_sub(p.Y, p.X), fp_sub(q.Y, q.X));
  Fp B = fp_mul(fp_add(p.Y, p.X), fp_add(q.Y, q.X));
  Fp C = fp_mul(fp_mul(p.T, fp_D2()), q.T);
  Fp Dd = fp_mul(fp_add(p.Z, p.Z), q.Z);
  Fp E = fp_sub(B, A), F = fp_sub(Dd, C), G = fp_add(Dd, C), H = fp_add(B, A);
  return Pt{fp_mul(E, F), fp_mul(G, H), fp_mul(F, G), fp_mul(E, H)};
}
// p + n (neg=false) or p - n (neg=true)
SP_HD Pt pt_madd(const Pt& p, const Niels& n, bool neg) {
  Fp yp = fp_select(n.yp, n.ym, neg), ym = fp_select(n.ym, n.yp, neg);
  Fp A = fp_mul(fp_sub(p.Y, p.X), ym);
  Fp B = fp_mul(fp_add(p.Y, p.X), yp);
  Fp C = fp_mul(p.T, n.t2d);
  Fp Dd = fp_add(p.Z, p.Z);
  Fp E = fp_sub(B, A), H = fp_add(B, A);
  Fp F = neg ? fp_add(Dd, C) : fp_sub(Dd, C);
  Fp G = neg ? fp_sub(Dd, C) : fp_add(Dd, C);
  return Pt{fp_mul(E, F), fp_mul(G, H), fp_mul(F, G), fp_mul(E, H)};
}
SP_HD Pt pt_dbl(const Pt& p) {  // dbl-2008-hwcd, a = -1
  Fp A = fp_sqr(p.X), B = fp_sqr(p.Y);
  Fp Zs = fp_sqr(p.Z);
  Fp C = fp_add(Zs, Zs);
  Fp Dd = fp_neg(A);
  Fp E = fp_sub(fp_sub(fp_sqr(fp_add(p.X, p.Y)), A), B);
  Fp G = fp_add(Dd, B), F = fp_sub(G, C), H = fp_sub(Dd, B);
  return Pt{fp_mul(E, F), fp_mul(G, H), fp_mul(F, G), fp_mul(E, H)};
}
// affine Niels form of p given 1/Z
SP_HD Niels pt_to_niels(const Pt& p, const Fp& zinv) {
  Fp x = fp_mul(p.X, zinv), y = fp_mul(p.Y, zinv);
  Niels n;
  n.yp = fp_add(y, x);
  n.ym = fp_sub(y, x);
  n.t2d = fp_mul(fp_mul(x, y), fp_D2());
  return n;
}

// RFC 9496 §4.3.2 Encode
SP_HD void pt_compress(const Pt& p, uint8_t out[32]) {
  Fp u1 = fp_mul(fp_add(p.Z, p.Y), fp_sub(p.Z, p.Y));
  Fp u2 = fp_mul(p.X, p.Y);
  Fp invsqrt;
  fp_sqrt_ratio_m1(fp_one(), fp_mul(u1, fp_sqr(u2)), &invsqrt);
  Fp den1 = fp_mul(invsqrt, u1), den2 = fp_mul(invsqrt, u2);
  Fp z_inv = fp_mul(fp_mul(den1, den2), p.T);
  Fp ix0 = fp_mul(p.X, fp_SQRT_M1()), iy0 = fp_mul(p.Y, fp_SQRT_M1());
  Fp ench = fp_mul(den1, fp_INVSQRT_A_MINUS_D());
  bool rotate = fp_is_negative(fp_mul(p.T, z_inv));
  Fp x = fp_select(p.X, iy0, rotate);
  Fp y = fp_select(p.Y, ix0, rotate);
  Fp den_inv = fp_select(den2, ench, rotate);
  y = fp_cneg(y, fp_is_negative(fp_mul(x, z_inv)));
  Fp s = fp_abs(fp_mul(den_inv, fp_sub(p.Z, y)));
  fp_to_bytes(s, out);
}
#if defined(__HIPCC__)
#define SP_HOST_ONLY __host__ inline
#else
#define SP_HOST_ONLY inline
#endif
// N encodes at once on a host core (N = 2, 3): the ~252 dependent squarings of the inverse square root are a latency chain (one
// squaring every ~14 ns against ~9 ns of multiplier time on the build host), so N independent chains advance in the same loop and the
// out-of-order core overlaps them: 4.1 us for one point, 5.6 us for two, bench/host_arith_probe.cc. Same bytes as pt_compress for every
// point (tests/test_host_arith.py). Used where the proving thread encodes several points back to back: L and R of an inner-product
// round (ipa.hip), the rows of a few-term commitment call (host_commit.hip), commitments of up to 8 rows summed on the device.
template <int N>
SP_HOST_ONLY void fp_pow2k_n(Fp (&a)[N], int k) {
  // the chains in named locals (not array elements): the compiler then keeps all N of them in registers across the loop
  Fp a0 = a[0], a1 = a[N > 1 ? 1 : 0], a2 = a[N > 2 ? 2 : 0], a3 = a[N > 3 ? 3 : 0];
  for (int i = 0; i < k; i++) {
    a0 = fp_sqr(a0);
    if (N > 1) a1 = fp_sqr(a1);
    if (N > 2) a2 = fp_sqr(a2);
    if (N > 3) a3 = fp_sqr(a3);
  }
  a[0] = a0;
  if (N > 1) a[1] = a1;
  if (N > 2) a[2] = a2;
  if (N > 3) a[3] = a3;
}
template <int N>
SP_HOST_ONLY void fp_pow_p58_n(const Fp (&z)[N], Fp (&out)[N]) {  // z^((p-5)/8), the ladder of fp_pow_ladder
  Fp z2[N], z9[N], z11[N], t[N], a[N], b[N], c[N];
  for (int j = 0; j < N; j++) { z2[j] = fp_sqr(z[j]); t[j] = z2[j]; }
  fp_pow2k_n<N>(t, 2);
  for (int j = 0; j < N; j++) { z9[j] = fp_mul(t[j], z[j]); z11[j] = fp_mul(z9[j], z2[j]); t[j] = fp_sqr(z11[j]); }
  for (int j = 0; j < N; j++) { a[j] = fp_mul(t[j], z9[j]); t[j] = a[j]; }  // a = z^(2^5 - 1)
  fp_pow2k_n<N>(t, 5);
  for (int j = 0; j < N; j++) { b[j] = fp_mul(t[j], a[j]); t[j] = b[j]; }   // b = z^(2^10 - 1)
  fp_pow2k_n<N>(t, 10);
  for (int j = 0; j < N; j++) { c[j] = fp_mul(t[j], b[j]); t[j] = c[j]; }   // c = z^(2^20 - 1)
  fp_pow2k_n<N>(t, 20);
  for (int j = 0; j < N; j++) t[j] = fp_mul(t[j], c[j]);                    // 2^40 - 1
  fp_pow2k_n<N>(t, 10);
  for (int j = 0; j < N; j++) { a[j] = fp_mul(t[j], b[j]); t[j] = a[j]; }   // a = z^(2^50 - 1)
  fp_pow2k_n<N>(t, 50);
  for (int j = 0; j < N; j++) { c[j] = fp_mul(t[j], a[j]); t[j] = c[j]; }   // c = z^(2^100 - 1)
  fp_pow2k_n<N>(t, 100);
  for (int j = 0; j < N; j++) t[j] = fp_mul(t[j], c[j]);                    // 2^200 - 1
  fp_pow2k_n<N>(t, 50);
  for (int j = 0; j < N; j++) t[j] = fp_mul(t[j], a[j]);                    // 2^250 - 1
  fp_pow2k_n<N>(t, 2);
  for (int j = 0; j < N; j++) out[j] = fp_mul(t[j], z[j]);
}
template <int N>
SP_HOST_ONLY void pt_compress_n(const Pt* p, uint8_t* out) {  // out: N x 32 bytes
  Fp u1[N], u2[N], v[N], v3[N], w[N], r[N];
  for (int j = 0; j < N; j++) {
    u1[j] = fp_mul(fp_add(p[j].Z, p[j].Y), fp_sub(p[j].Z, p[j].Y));
    u2[j] = fp_mul(p[j].X, p[j].Y);
    v[j] = fp_mul(u1[j], fp_sqr(u2[j]));
    // SQRT_RATIO_M1(1, v): r = v^3 (v^7)^((p-5)/8)
    v3[j] = fp_mul(fp_sqr(v[j]), v[j]);
    w[j] = fp_mul(fp_sqr(v3[j]), v[j]);
  }
  fp_pow_p58_n<N>(w, r);
  for (int j = 0; j < N; j++) {
    Fp rr = fp_mul(v3[j], r[j]);
    Fp check = fp_mul(v[j], fp_sqr(rr));
    Fp one = fp_one(), neg_u = fp_neg(one);
    bool flipped = fp_eq(check, neg_u);
    bool flipped_i = fp_eq(check, fp_mul(neg_u, fp_SQRT_M1()));
    Fp r_i = fp_mul(rr, fp_SQRT_M1());
    Fp invsqrt = fp_abs(fp_select(rr, r_i, flipped || flipped_i));
    Fp den1 = fp_mul(invsqrt, u1[j]), den2 = fp_mul(invsqrt, u2[j]);
    Fp z_inv = fp_mul(fp_mul(den1, den2), p[j].T);
    Fp ix0 = fp_mul(p[j].X, fp_SQRT_M1()), iy0 = fp_mul(p[j].Y, fp_SQRT_M1());
    Fp ench = fp_mul(den1, fp_INVSQRT_A_MINUS_D());
    bool rotate = fp_is_negative(fp_mul(p[j].T, z_inv));
    Fp x = fp_select(p[j].X, iy0, rotate);
    Fp y = fp_select(p[j].Y, ix0, rotate);
    Fp den_inv = fp_select(den2, ench, rotate);
    y = fp_cneg(y, fp_is_negative(fp_mul(x, z_inv)));
    Fp s = fp_abs(fp_mul(den_inv, fp_sub(p[j].Z, y)));
    fp_to_bytes(s, out + 32 * j);
  }
}
// n points, encoded in pairs (5.6 us per pair against 4.1 us for one point and 8.4 / 11.4 us for three / four at once: two chains
// fill the multiplier; bench/host_arith_probe.cc), a last odd three together
SP_HOST_ONLY void pt_compress_many(const Pt* p, size_t n, uint8_t* out) {
  size_t i = 0;
  for (; n - i >= 4; i += 2) pt_compress_n<2>(p + i, out + 32 * i);
  if (n - i == 3) pt_compress_n<3>(p + i, out + 32 * i);
  else if (n - i == 2) pt_compress_n<2>(p + i, out + 32 * i);
  else if (n - i == 1) pt_compress(p[i], out + 32 * i);
}
// RFC 9496 §4.3.1 Decode
SP_HD bool pt_decompress(const uint8_t in[32], Pt* out) {
  Fp s = fp_from_bytes(in);
  uint8_t chk[32];
  fp_to_bytes(s, chk);
  bool canonical = true;
  for (int i = 0; i < 32; i++) canonical = canonical && (chk[i] == in[i]);
  if (!canonical || (in[0] & 1)) return false;
  Fp ss = fp_sqr(s);
  Fp u1 = fp_sub(fp_one(), ss), u2 = fp_add(fp_one(), ss);
  Fp u2s = fp_sqr(u2);
  Fp v = fp_sub(fp_neg(fp_mul(fp_D(), fp_sqr(u1))), u2s);
  Fp invsqrt;
  bool was_square = fp_sqrt_ratio_m1(fp_one(), fp_mul(v, u2s), &invsqrt);
  Fp den_x = fp_mul(invsqrt, u2);
  Fp den_y = fp_mul(fp_mul(invsqrt, den_x), v);
  Fp x = fp_abs(fp_mul(fp_add(s, s), den_x));
  Fp y = fp_mul(u1, den_y);
  Fp t = fp_mul(x, y);
  if (!was_square || fp_is_negative(t) || fp_is_zero(y)) return false;
  *out = Pt{x, y, fp_one(), t};
  return true;
}
// RFC 9496 §4.3.4 MAP
SP_HD Pt pt_elligator(const Fp& t) {
  Fp one = fp_one();
  Fp r = fp_mul(fp_SQRT_M1(), fp_sqr(t));
  Fp u = fp_mul(fp_add(r, one), fp_ONE_MINUS_D_SQ());
  Fp v = fp_mul(fp_sub(fp_neg(one), fp_mul(r, fp_D())), fp_add(r, fp_D()));
  Fp s;
  bool was_square = fp_sqrt_ratio_m1(u, v, &s);
  Fp s_prime = fp_neg(fp_abs(fp_mul(s, t)));
  s = fp_select(s_prime, s, was_square);
  Fp c = fp_select(r, fp_neg(one), was_square);
  Fp N = fp_sub(fp_mul(fp_mul(c, fp_sub(r, one)), fp_D_MINUS_ONE_SQ()), v);
  Fp w0 = fp_mul(fp_add(s, s), v);
  Fp w1 = fp_mul(N, fp_SQRT_AD_MINUS_ONE());
  Fp s2 = fp_sqr(s);
  Fp w2 = fp_sub(one, s2);
  Fp w3 = fp_add(one, s2);
  return Pt{fp_mul(w0, w3), fp_mul(w2, w1), fp_mul(w1, w3), fp_mul(w0, w2)};
}
// dalek RistrettoPoint::from_uniform_bytes (= RFC 9496 one-way map), src/commitments.rs:25
SP_HD Pt pt_from_uniform_bytes(const uint8_t b[64]) {
  Fp t1 = fp_from_bytes(b), t2 = fp_from_bytes(b + 32);
  return pt_add(pt_elligator(t1), pt_elligator(t2));
}

// ---------------------------------------------------------------- serial-chain forms (fe10.hpp): LDS tree additions
// and the ristretto encode executed by a handful of waves on the commit critical path. Same group elements, same bytes.
struct Pt10 {
  Fe10 X, Y, Z, T;
};
SP_HD Pt10 pt10_identity() { return Pt10{Fe10{{0, 0, 0, 0, 0, 0, 0, 0, 0, 0}}, fe10_one(), fe10_one(), Fe10{{0, 0, 0, 0, 0, 0, 0, 0, 0, 0}}}; }
SP_HD Pt10 pt10_load(const Pt& p) { return Pt10{fe10_load(p.X), fe10_load(p.Y), fe10_load(p.Z), fe10_load(p.T)}; }
SP_HD Pt10 pt10_add(const Pt10& p, const Pt10& q) {
  Fe10 A = fe10_mul(fe10_sub(p.Y, p.X), fe10_sub(q.Y, q.X));
  Fe10 B = fe10_mul(fe10_add(p.Y, p.X), fe10_add(q.Y, q.X));
  Fe10 C = fe10_mul(fe10_mul(p.T, fe10_load(fp_D2())), q.T);
  Fe10 Dd = fe10_mul(fe10_add(p.Z, p.Z), q.Z);
  Fe10 E = fe10_sub(B, A), F = fe10_sub(Dd, C), G = fe10_add(Dd, C), H = fe10_add(B, A);
  return Pt10{fe10_mul(E, F), fe10_mul(G, H), fe10_mul(F, G), fe10_mul(E, H)};
}
SP_HD bool fe10_is_negative(const Fe10& a) { return fe10_to_fp(a).v[0] & 1; }
SP_HD Fe10 fe10_abs(const Fe10& a) { return fe10_select(a, fe10_neg(a), fe10_is_negative(a)); }
// RFC 9496 §4.3.2 Encode on radix-2^25.5 limbs
SP_HD void pt10_compress(const Pt10& p, uint8_t out[32]) {
  Fe10 sqrt_m1 = fe10_load(fp_SQRT_M1());
  Fe10 u1 = fe10_mul(fe10_add(p.Z, p.Y), fe10_sub(p.Z, p.Y));
  Fe10 u2 = fe10_mul(p.X, p.Y);
  // SQRT_RATIO_M1(1, v) with v = u1 * u2^2
  Fe10 v = fe10_mul(u1, fe10_sqr(u2));
  Fe10 v3 = fe10_mul(fe10_sqr(v), v);
  Fe10 v7 = fe10_mul(fe10_sqr(v3), v);
  Fe10 r = fe10_mul(v3, fe10_pow_p58(v7));
  Fp check = fe10_to_fp(fe10_mul(v, fe10_sqr(r)));
  bool flipped = fp_eq(check, fp_neg(fp_one()));
  bool flipped_i = fp_eq(check, fp_neg(fp_SQRT_M1()));
  r = fe10_select(r, fe10_mul(r, sqrt_m1), flipped || flipped_i);
  Fe10 invsqrt = fe10_abs(r);
  Fe10 den1 = fe10_mul(invsqrt, u1), den2 = fe10_mul(invsqrt, u2);
  Fe10 z_inv = fe10_mul(fe10_mul(den1, den2), p.T);
  Fe10 ix0 = fe10_mul(p.X, sqrt_m1), iy0 = fe10_mul(p.Y, sqrt_m1);
  Fe10 ench = fe10_mul(den1, fe10_load(fp_INVSQRT_A_MINUS_D()));
  bool rotate = fe10_is_negative(fe10_mul(p.T, z_inv));
  Fe10 x = fe10_select(p.X, iy0, rotate);
  Fe10 y = fe10_select(p.Y, ix0, rotate);
  Fe10 den_inv = fe10_select(den2, ench, rotate);
  y = fe10_select(y, fe10_neg(y), fe10_is_negative(fe10_mul(x, z_inv)));
  Fe10 s = fe10_abs(fe10_mul(den_inv, fe10_sub(p.Z, y)));
  fp_to_bytes(fe10_to_fp(s), out);
}

}  // namespace sp
