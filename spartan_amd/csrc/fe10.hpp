// spartan_amd: radix-2^25.5 arithmetic mod p = 2^255-19 for SERIAL chains executed by a lone wavefront.
//
// A single wave issues roughly one instruction every 5 cycles whatever its type, so the time of the
// 254-squaring inverse-square-root ladder inside every ristretto encode (RFC 9496 §4.3.2; one per commitment,
// ~130 of them strictly sequential per proof) is its instruction count. With ten unsigned limbs of 26/25 bits a
// partial product plus accumulation is ONE v_mad_u64_u32, there are ten independent accumulators (no dependent
// stalls), and reduction is a shift/mask carry chain: ~130 VALU instructions per squaring, against ~360 scalar
// instructions when the compiler scalarises the 4x64 form (it does, for wave-uniform data) or ~250 vector ones.
// Values are identical to field.hpp's; tests/test_host_arith.py checks both against Python integers.
#pragma once
#include "field.hpp"

namespace sp {

struct Fe10 {
  uint32_t v[10];  // limb i has weight 2^ceil(25.5 i); even limbs < 2^26 + eps, odd limbs < 2^25 + eps
};

// keep the limbs in vector registers: inline-asm outputs are treated as divergent, which stops the compiler from
// moving the chain to the scalar ALU (no 64-bit multiply-add there).
SP_HD void fe10_pin(Fe10& a) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
  for (int i = 0; i < 10; i++) asm volatile("" : "+v"(a.v[i]));
#else
  (void)a;
#endif
}

SP_HD Fe10 fe10_from_fp(const Fp& x) {
  Fp c = fp_canon(x);  // < 2^255
  const int off[10] = {0, 26, 51, 77, 102, 128, 153, 179, 204, 230};
  Fe10 r;
#pragma unroll
  for (int i = 0; i < 10; i++) {
    int w = off[i] >> 6, s = off[i] & 63, bits = (i & 1) ? 25 : 26;
    uint64_t lo = c.v[w] >> s;
    if (s + bits > 64 && w < 3) lo |= c.v[w + 1] << (64 - s);
    r.v[i] = (uint32_t)(lo & (((uint64_t)1 << bits) - 1));
  }
  return r;
}
SP_HD Fp fe10_to_fp(const Fe10& a) {
  // limbs may slightly exceed their width: accumulate with carries into 4x64 (+ overflow folded by 38)
  const int off[10] = {0, 26, 51, 77, 102, 128, 153, 179, 204, 230};
  uint64_t w[5] = {0, 0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < 10; i++) {
    int k = off[i] >> 6, s = off[i] & 63;
    u128 t = (u128)a.v[i] << s;
    u128 c = (u128)w[k] + (uint64_t)t;
    w[k] = (uint64_t)c;
    c = (c >> 64) + (uint64_t)(t >> 64);
#pragma unroll
    for (int m = k + 1; m < 5; m++) {
      c += w[m];
      w[m] = (uint64_t)c;
      c >>= 64;
    }
  }
  uint64_t t8[8] = {w[0], w[1], w[2], w[3], w[4], 0, 0, 0};
  return fp_reduce512(t8);
}

SP_HD Fe10 fe10_carry(uint64_t h[10]) {
  const uint64_t M26 = (1u << 26) - 1, M25 = (1u << 25) - 1;
  uint64_t c;
  // two interleaved chains (as in the ref10 code) to shorten the dependent path
  c = h[0] >> 26; h[1] += c; h[0] &= M26;
  c = h[4] >> 26; h[5] += c; h[4] &= M26;
  c = h[1] >> 25; h[2] += c; h[1] &= M25;
  c = h[5] >> 25; h[6] += c; h[5] &= M25;
  c = h[2] >> 26; h[3] += c; h[2] &= M26;
  c = h[6] >> 26; h[7] += c; h[6] &= M26;
  c = h[3] >> 25; h[4] += c; h[3] &= M25;
  c = h[7] >> 25; h[8] += c; h[7] &= M25;
  c = h[4] >> 26; h[5] += c; h[4] &= M26;
  c = h[8] >> 26; h[9] += c; h[8] &= M26;
  c = h[9] >> 25; h[0] += 19 * c; h[9] &= M25;
  c = h[0] >> 26; h[1] += c; h[0] &= M26;
  Fe10 r;
#pragma unroll
  for (int i = 0; i < 10; i++) r.v[i] = (uint32_t)h[i];
  return r;
}
SP_HD Fe10 fe10_mul(const Fe10& f, const Fe10& g) {
  uint32_t g19[10], f2[10];
#pragma unroll
  for (int i = 0; i < 10; i++) {
    g19[i] = 19 * g.v[i];
    f2[i] = 2 * f.v[i];
  }
  uint64_t h[10];
#pragma unroll
  for (int k = 0; k < 10; k++) h[k] = 0;
#pragma unroll
  for (int i = 0; i < 10; i++)
#pragma unroll
    for (int j = 0; j < 10; j++) {
      uint32_t a = ((i & 1) && (j & 1)) ? f2[i] : f.v[i];
      uint32_t b = (i + j >= 10) ? g19[j] : g.v[j];
      h[(i + j) % 10] += (uint64_t)a * b;
    }
  return fe10_carry(h);
}
SP_HD Fe10 fe10_sqr(const Fe10& f) {
  uint32_t f19[10], f2[10], f4[10];
#pragma unroll
  for (int i = 0; i < 10; i++) {
    f19[i] = 19 * f.v[i];
    f2[i] = 2 * f.v[i];
    f4[i] = 4 * f.v[i];
  }
  uint64_t h[10];
#pragma unroll
  for (int k = 0; k < 10; k++) h[k] = 0;
#pragma unroll
  for (int i = 0; i < 10; i++) {
    {  // diagonal term f_i^2 (x2 when i is odd)
      uint32_t a = (i & 1) ? f2[i] : f.v[i];
      uint32_t b = (2 * i >= 10) ? f19[i] : f.v[i];
      h[(2 * i) % 10] += (uint64_t)a * b;
    }
#pragma unroll
    for (int j = i + 1; j < 10; j++) {  // cross terms counted twice (x4 when both odd)
      uint32_t a = ((i & 1) && (j & 1)) ? f4[i] : f2[i];
      uint32_t b = (i + j >= 10) ? f19[j] : f.v[j];
      h[(i + j) % 10] += (uint64_t)a * b;
    }
  }
  return fe10_carry(h);
}
SP_HD Fe10 fe10_pow2k(Fe10 a, int k) {
  for (int i = 0; i < k; i++) a = fe10_sqr(a);
  return a;
}
// z^(2^250-1) and z^11
SP_HD void fe10_pow_ladder(const Fe10& z, Fe10* z2_250_0, Fe10* z11) {
  Fe10 z2 = fe10_sqr(z);
  Fe10 z9 = fe10_mul(fe10_pow2k(z2, 2), z);
  *z11 = fe10_mul(z9, z2);
  Fe10 z2_5_0 = fe10_mul(fe10_sqr(*z11), z9);
  Fe10 z2_10_0 = fe10_mul(fe10_pow2k(z2_5_0, 5), z2_5_0);
  Fe10 z2_20_0 = fe10_mul(fe10_pow2k(z2_10_0, 10), z2_10_0);
  Fe10 z2_40_0 = fe10_mul(fe10_pow2k(z2_20_0, 20), z2_20_0);
  Fe10 z2_50_0 = fe10_mul(fe10_pow2k(z2_40_0, 10), z2_10_0);
  Fe10 z2_100_0 = fe10_mul(fe10_pow2k(z2_50_0, 50), z2_50_0);
  Fe10 z2_200_0 = fe10_mul(fe10_pow2k(z2_100_0, 100), z2_100_0);
  *z2_250_0 = fe10_mul(fe10_pow2k(z2_200_0, 50), z2_50_0);
}
SP_HD Fp fp_pow_p58_serial(const Fp& z) {  // z^((p-5)/8), for lone-wave callers
  Fe10 x = fe10_from_fp(z);
  fe10_pin(x);
  Fe10 t, z11;
  fe10_pow_ladder(x, &t, &z11);
  return fe10_to_fp(fe10_mul(fe10_pow2k(t, 2), x));
}
SP_HD Fp fp_invert_serial(const Fp& z) {  // z^(p-2)
  Fe10 x = fe10_from_fp(z);
  fe10_pin(x);
  Fe10 t, z11;
  fe10_pow_ladder(x, &t, &z11);
  return fe10_to_fp(fe10_mul(fe10_pow2k(t, 5), z11));
}

}  // namespace sp
