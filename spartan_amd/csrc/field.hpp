// spartan_amd: field arithmetic shared by the HIP kernels and the host-side prover driver.
//
//   Fq  scalar field of ristretto255, q = 2^252 + 27742317777372353535851937790883648493.
//       In-memory form is exactly libspartan's `Scalar([u64;4])`: little-endian limbs of the Montgomery
//       residue x*2^256 mod q, always fully reduced to [0,q) (reference: src/scalar/ristretto255.rs:199,
//       248-328, 642-760), so device buffers are bit-compatible with a Rust `&[Scalar]`.
//   Fp  base field of edwards25519, p = 2^255 - 19. 4x64 saturated limbs, weakly reduced (any value in
//       [0,2^256) congruent to the element); 2^256 = 38 (mod p) folds carries. bench/ubench_fpmul.hip on
//       MI355X, full-width operands: 4x64 via __int128 = 156 Gmul/s, 8x32 = 153, 10x25.5 = 154 — all within 5 %
//       (the 32x32 multiplier rate is what binds), so the layout that is also the natural host form was kept.
//
// Everything is SP_HD (host+device) so the very same source is unit-tested on the CPU against the oracle
// (tests/csrc/hostcheck.cc) and then runs on gfx950.
#pragma once
#include <cstdint>
#include <cstring>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define SP_HD __host__ __device__ __forceinline__
#define SP_HD_NOINLINE __host__ __device__ __noinline__
#else
#define SP_HD inline
#define SP_HD_NOINLINE
#endif

namespace sp {

typedef unsigned __int128 u128;

// ------------------------------------------------------------------ Fq
struct Fq {
  uint64_t l[4];
};

#define SP_Q0 0x5812631a5cf5d3edULL
#define SP_Q1 0x14def9dea2f79cd6ULL
#define SP_Q2 0x0000000000000000ULL
#define SP_Q3 0x1000000000000000ULL
#define SP_QINV 0xd2b51da312547e1bULL /* -(q^-1) mod 2^64, ristretto255.rs:304 */

SP_HD Fq fq_zero() { return Fq{{0, 0, 0, 0}}; }
SP_HD Fq fq_one() {  // R mod q, ristretto255.rs:307-312
  return Fq{{0xd6ec31748d98951dULL, 0xc6ef5bf4737dcf70ULL, 0xfffffffffffffffeULL, 0x0fffffffffffffffULL}};
}
SP_HD Fq fq_R2() {  // ristretto255.rs:315-320
  return Fq{{0xa40611e3449c0f01ULL, 0xd00e1ba768859347ULL, 0xceec73d217f5be65ULL, 0x0399411b7c309a3dULL}};
}
SP_HD Fq fq_R3() {  // ristretto255.rs:323-328
  return Fq{{0x2a9e49687b83a2dbULL, 0x278324e6aef7f3ecULL, 0x8065dc6c04ec5b65ULL, 0x0e530b773599cec7ULL}};
}
SP_HD bool fq_is_zero(const Fq& a) { return (a.l[0] | a.l[1] | a.l[2] | a.l[3]) == 0; }
SP_HD bool fq_eq(const Fq& a, const Fq& b) {
  return ((a.l[0] ^ b.l[0]) | (a.l[1] ^ b.l[1]) | (a.l[2] ^ b.l[2]) | (a.l[3] ^ b.l[3])) == 0;
}

#if defined(__HIP_DEVICE_COMPILE__) && !defined(SP_FIELD_ADD_GENERIC)
// ---- device add/sub: two interleaved 8-word carry chains + one select, in inline asm.
// The u128 formulations below compile to 48-66 VALU instructions plus ~28 wait-state s_nops per field addition (on gfx9 two
// wait states separate a VALU write of a carry mask from the VALU read of it, and the compiler serialises every chain): a
// quarter of the cubic sum-check kernels and of a mixed point addition. Here chain 1 is (a op b) word by word, chain 2
// applies the correction to chain 1's words one step behind it (t - q, d + q, s + 38, d - 38), each chain carrying through
// its own SGPR pair, so the other chain's instruction and one s_nop 0 are all the padding a carry needs; the final carry
// of one of the chains selects, per lane, which of the two results is the reduced one. 8 + 8 + 8 instructions.
template <int OP>  // OP 0: a + b then - k (fq_add) | 1: a - b then + k (fq_sub) | 2: a + b then + k (fp_add) | 3: a - b then - k (fp_sub)
__device__ __forceinline__ void sp_chain2(const uint32_t (&a)[8], const uint32_t (&b)[8], const uint32_t (&k)[8], uint32_t (&s)[8], uint32_t (&d)[8],
                                          uint64_t& cA, uint64_t& cB) {
  // low four words: the chains start (no carry-in)
  if (OP == 0)
    asm("v_add_co_u32_e64 %0, %8, %10, %14\n\tv_sub_co_u32_e64 %4, %9, %0, %18\n\ts_nop 0\n\t"
        "v_addc_co_u32_e64 %1, %8, %11, %15, %8\n\tv_subb_co_u32_e64 %5, %9, %1, %19, %9\n\ts_nop 0\n\t"
        "v_addc_co_u32_e64 %2, %8, %12, %16, %8\n\tv_subb_co_u32_e64 %6, %9, %2, %20, %9\n\ts_nop 0\n\t"
        "v_addc_co_u32_e64 %3, %8, %13, %17, %8\n\tv_subb_co_u32_e64 %7, %9, %3, %21, %9\n\ts_nop 0"
        : "=&v"(s[0]), "=&v"(s[1]), "=&v"(s[2]), "=&v"(s[3]), "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&s"(cA), "=&s"(cB)
        : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(k[0]), "v"(k[1]), "v"(k[2]), "v"(k[3]));
  else if (OP == 1)
    asm("v_sub_co_u32_e64 %0, %8, %10, %14\n\tv_add_co_u32_e64 %4, %9, %0, %18\n\ts_nop 0\n\t"
        "v_subb_co_u32_e64 %1, %8, %11, %15, %8\n\tv_addc_co_u32_e64 %5, %9, %1, %19, %9\n\ts_nop 0\n\t"
        "v_subb_co_u32_e64 %2, %8, %12, %16, %8\n\tv_addc_co_u32_e64 %6, %9, %2, %20, %9\n\ts_nop 0\n\t"
        "v_subb_co_u32_e64 %3, %8, %13, %17, %8\n\tv_addc_co_u32_e64 %7, %9, %3, %21, %9\n\ts_nop 0"
        : "=&v"(s[0]), "=&v"(s[1]), "=&v"(s[2]), "=&v"(s[3]), "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&s"(cA), "=&s"(cB)
        : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(k[0]), "v"(k[1]), "v"(k[2]), "v"(k[3]));
  else if (OP == 2)
    asm("v_add_co_u32_e64 %0, %8, %10, %14\n\tv_add_co_u32_e64 %4, %9, %0, %18\n\ts_nop 0\n\t"
        "v_addc_co_u32_e64 %1, %8, %11, %15, %8\n\tv_addc_co_u32_e64 %5, %9, %1, %19, %9\n\ts_nop 0\n\t"
        "v_addc_co_u32_e64 %2, %8, %12, %16, %8\n\tv_addc_co_u32_e64 %6, %9, %2, %20, %9\n\ts_nop 0\n\t"
        "v_addc_co_u32_e64 %3, %8, %13, %17, %8\n\tv_addc_co_u32_e64 %7, %9, %3, %21, %9\n\ts_nop 0"
        : "=&v"(s[0]), "=&v"(s[1]), "=&v"(s[2]), "=&v"(s[3]), "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&s"(cA), "=&s"(cB)
        : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(k[0]), "v"(k[1]), "v"(k[2]), "v"(k[3]));
  else
    asm("v_sub_co_u32_e64 %0, %8, %10, %14\n\tv_sub_co_u32_e64 %4, %9, %0, %18\n\ts_nop 0\n\t"
        "v_subb_co_u32_e64 %1, %8, %11, %15, %8\n\tv_subb_co_u32_e64 %5, %9, %1, %19, %9\n\ts_nop 0\n\t"
        "v_subb_co_u32_e64 %2, %8, %12, %16, %8\n\tv_subb_co_u32_e64 %6, %9, %2, %20, %9\n\ts_nop 0\n\t"
        "v_subb_co_u32_e64 %3, %8, %13, %17, %8\n\tv_subb_co_u32_e64 %7, %9, %3, %21, %9\n\ts_nop 0"
        : "=&v"(s[0]), "=&v"(s[1]), "=&v"(s[2]), "=&v"(s[3]), "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&s"(cA), "=&s"(cB)
        : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(k[0]), "v"(k[1]), "v"(k[2]), "v"(k[3]));
  // high four words: both chains continue from their carries
  if (OP == 0)
    asm("v_addc_co_u32_e64 %0, %8, %10, %14, %8\n\tv_subb_co_u32_e64 %4, %9, %0, %18, %9\n\ts_nop 0\n\t"
        "v_addc_co_u32_e64 %1, %8, %11, %15, %8\n\tv_subb_co_u32_e64 %5, %9, %1, %19, %9\n\ts_nop 0\n\t"
        "v_addc_co_u32_e64 %2, %8, %12, %16, %8\n\tv_subb_co_u32_e64 %6, %9, %2, %20, %9\n\ts_nop 0\n\t"
        "v_addc_co_u32_e64 %3, %8, %13, %17, %8\n\tv_subb_co_u32_e64 %7, %9, %3, %21, %9\n\ts_nop 0"
        : "=&v"(s[4]), "=&v"(s[5]), "=&v"(s[6]), "=&v"(s[7]), "=&v"(d[4]), "=&v"(d[5]), "=&v"(d[6]), "=&v"(d[7]), "+s"(cA), "+s"(cB)
        : "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]), "v"(k[4]), "v"(k[5]), "v"(k[6]), "v"(k[7]));
  else if (OP == 1)
    asm("v_subb_co_u32_e64 %0, %8, %10, %14, %8\n\tv_addc_co_u32_e64 %4, %9, %0, %18, %9\n\ts_nop 0\n\t"
        "v_subb_co_u32_e64 %1, %8, %11, %15, %8\n\tv_addc_co_u32_e64 %5, %9, %1, %19, %9\n\ts_nop 0\n\t"
        "v_subb_co_u32_e64 %2, %8, %12, %16, %8\n\tv_addc_co_u32_e64 %6, %9, %2, %20, %9\n\ts_nop 0\n\t"
        "v_subb_co_u32_e64 %3, %8, %13, %17, %8\n\tv_addc_co_u32_e64 %7, %9, %3, %21, %9\n\ts_nop 0"
        : "=&v"(s[4]), "=&v"(s[5]), "=&v"(s[6]), "=&v"(s[7]), "=&v"(d[4]), "=&v"(d[5]), "=&v"(d[6]), "=&v"(d[7]), "+s"(cA), "+s"(cB)
        : "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]), "v"(k[4]), "v"(k[5]), "v"(k[6]), "v"(k[7]));
  else if (OP == 2)
    asm("v_addc_co_u32_e64 %0, %8, %10, %14, %8\n\tv_addc_co_u32_e64 %4, %9, %0, %18, %9\n\ts_nop 0\n\t"
        "v_addc_co_u32_e64 %1, %8, %11, %15, %8\n\tv_addc_co_u32_e64 %5, %9, %1, %19, %9\n\ts_nop 0\n\t"
        "v_addc_co_u32_e64 %2, %8, %12, %16, %8\n\tv_addc_co_u32_e64 %6, %9, %2, %20, %9\n\ts_nop 0\n\t"
        "v_addc_co_u32_e64 %3, %8, %13, %17, %8\n\tv_addc_co_u32_e64 %7, %9, %3, %21, %9\n\ts_nop 0"
        : "=&v"(s[4]), "=&v"(s[5]), "=&v"(s[6]), "=&v"(s[7]), "=&v"(d[4]), "=&v"(d[5]), "=&v"(d[6]), "=&v"(d[7]), "+s"(cA), "+s"(cB)
        : "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]), "v"(k[4]), "v"(k[5]), "v"(k[6]), "v"(k[7]));
  else
    asm("v_subb_co_u32_e64 %0, %8, %10, %14, %8\n\tv_subb_co_u32_e64 %4, %9, %0, %18, %9\n\ts_nop 0\n\t"
        "v_subb_co_u32_e64 %1, %8, %11, %15, %8\n\tv_subb_co_u32_e64 %5, %9, %1, %19, %9\n\ts_nop 0\n\t"
        "v_subb_co_u32_e64 %2, %8, %12, %16, %8\n\tv_subb_co_u32_e64 %6, %9, %2, %20, %9\n\ts_nop 0\n\t"
        "v_subb_co_u32_e64 %3, %8, %13, %17, %8\n\tv_subb_co_u32_e64 %7, %9, %3, %21, %9\n\ts_nop 0"
        : "=&v"(s[4]), "=&v"(s[5]), "=&v"(s[6]), "=&v"(s[7]), "=&v"(d[4]), "=&v"(d[5]), "=&v"(d[6]), "=&v"(d[7]), "+s"(cA), "+s"(cB)
        : "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]), "v"(k[4]), "v"(k[5]), "v"(k[6]), "v"(k[7]));
}
// r[i] = sel ? x[i] : y[i] per lane; `both` = 1 where selA and selB are both set (the rare second wrap of the Fp forms)
__device__ __forceinline__ void sp_select8(uint32_t (&r)[8], const uint32_t (&x)[8], const uint32_t (&y)[8], uint64_t sel) {
  asm("s_nop 1\n\t"
      "v_cndmask_b32_e64 %0, %16, %8, %24\n\tv_cndmask_b32_e64 %1, %17, %9, %24\n\tv_cndmask_b32_e64 %2, %18, %10, %24\n\t"
      "v_cndmask_b32_e64 %3, %19, %11, %24\n\tv_cndmask_b32_e64 %4, %20, %12, %24\n\tv_cndmask_b32_e64 %5, %21, %13, %24\n\t"
      "v_cndmask_b32_e64 %6, %22, %14, %24\n\tv_cndmask_b32_e64 %7, %23, %15, %24"
      : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7])
      : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]), "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]), "v"(y[4]),
        "v"(y[5]), "v"(y[6]), "v"(y[7]), "s"(sel));
}
__device__ __forceinline__ uint32_t sp_both(uint64_t selA, uint64_t selB) {
  uint32_t f;
  uint64_t t;
  asm("s_and_b64 %1, %2, %3\n\ts_nop 1\n\tv_cndmask_b32_e64 %0, 0, 1, %1" : "=v"(f), "=&s"(t) : "s"(selA), "s"(selB) : "scc");
  return f;
}
template <typename F>
__device__ __forceinline__ void sp_split8(const F& v, uint32_t (&w)[8]) {
  const uint64_t* p = reinterpret_cast<const uint64_t*>(&v);
#pragma unroll
  for (int i = 0; i < 4; i++) { w[2 * i] = (uint32_t)p[i]; w[2 * i + 1] = (uint32_t)(p[i] >> 32); }
}
template <typename F>
__device__ __forceinline__ F sp_join8(const uint32_t (&w)[8]) {
  F r;
  uint64_t* p = reinterpret_cast<uint64_t*>(&r);
#pragma unroll
  for (int i = 0; i < 4; i++) p[i] = (uint64_t)w[2 * i] | ((uint64_t)w[2 * i + 1] << 32);
  return r;
}
#define SP_Q_WORDS {(uint32_t)SP_Q0, (uint32_t)(SP_Q0 >> 32), (uint32_t)SP_Q1, (uint32_t)(SP_Q1 >> 32), 0u, 0u, 0u, (uint32_t)(SP_Q3 >> 32)}
#endif

// t - q if t >= q else t   (t < 2q)
SP_HD Fq fq_csub(const Fq& t) {
  const uint64_t Q[4] = {SP_Q0, SP_Q1, SP_Q2, SP_Q3};
  Fq d;
  uint64_t borrow = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    u128 x = (u128)t.l[i] - Q[i] - borrow;
    d.l[i] = (uint64_t)x;
    borrow = (uint64_t)(x >> 64) & 1;
  }
  Fq r;
#pragma unroll
  for (int i = 0; i < 4; i++) r.l[i] = borrow ? t.l[i] : d.l[i];
  return r;
}
#if defined(__HIP_DEVICE_COMPILE__) && !defined(SP_FIELD_ADD_GENERIC)
__device__ __forceinline__ Fq fq_add(const Fq& a, const Fq& b) {  // ristretto255.rs:736-745: s = a + b, then s - q unless that borrows
  uint32_t aw[8], bw[8], s[8], d[8], r[8];
  const uint32_t q[8] = SP_Q_WORDS;
  sp_split8(a, aw); sp_split8(b, bw);
  uint64_t cA, cB;
  sp_chain2<0>(aw, bw, q, s, d, cA, cB);
  sp_select8(r, s, d, cB);  // borrow: s < q, keep s
  return sp_join8<Fq>(r);
}
__device__ __forceinline__ Fq fq_sub(const Fq& a, const Fq& b) {  // ristretto255.rs:718-733: d = a - b, plus q if that borrowed
  uint32_t aw[8], bw[8], d[8], e[8], r[8];
  const uint32_t q[8] = SP_Q_WORDS;
  sp_split8(a, aw); sp_split8(b, bw);
  uint64_t cA, cB;
  sp_chain2<1>(aw, bw, q, d, e, cA, cB);
  sp_select8(r, e, d, cA);
  return sp_join8<Fq>(r);
}
__host__ inline Fq fq_add(const Fq& a, const Fq& b) {
#else
SP_HD Fq fq_add(const Fq& a, const Fq& b) {  // ristretto255.rs:736-745
#endif
  Fq s;
  u128 c = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    c += (u128)a.l[i] + b.l[i];
    s.l[i] = (uint64_t)c;
    c >>= 64;
  }
  return fq_csub(s);  // a,b < q < 2^253: no carry out
}
#if defined(__HIP_DEVICE_COMPILE__) && !defined(SP_FIELD_ADD_GENERIC)
__host__ inline Fq fq_sub(const Fq& a, const Fq& b) {
#else
SP_HD Fq fq_sub(const Fq& a, const Fq& b) {  // ristretto255.rs:718-733
#endif
  const uint64_t Q[4] = {SP_Q0, SP_Q1, SP_Q2, SP_Q3};
  Fq d;
  uint64_t borrow = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    u128 x = (u128)a.l[i] - b.l[i] - borrow;
    d.l[i] = (uint64_t)x;
    borrow = (uint64_t)(x >> 64) & 1;
  }
  uint64_t mask = 0 - borrow;
  u128 c = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    c += (u128)d.l[i] + (Q[i] & mask);
    d.l[i] = (uint64_t)c;
    c >>= 64;
  }
  return d;
}
SP_HD Fq fq_neg(const Fq& a) { return fq_sub(fq_zero(), a); }  // 0-0 = 0 stays canonical
SP_HD Fq fq_dbl(const Fq& a) { return fq_add(a, a); }

// Montgomery reduction of a 512-bit value (< q*2^256): returns t/2^256 mod q in [0,q). ristretto255.rs:642-686
SP_HD Fq fq_mont_reduce(const uint64_t tin[8]) {
  const uint64_t Q[4] = {SP_Q0, SP_Q1, SP_Q2, SP_Q3};
  uint64_t r[8];
#pragma unroll
  for (int i = 0; i < 8; i++) r[i] = tin[i];
  uint64_t carry2 = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    uint64_t k = r[i] * SP_QINV;
    u128 c = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      c += (u128)k * Q[j] + r[i + j];
      r[i + j] = (uint64_t)c;
      c >>= 64;
    }
    c += (u128)r[i + 4] + carry2;
    r[i + 4] = (uint64_t)c;
    carry2 = (uint64_t)(c >> 64);
  }
  Fq t = {{r[4], r[5], r[6], r[7]}};
  return fq_csub(t);
}
#if defined(__HIP_DEVICE_COMPILE__) && !defined(SP_FQ_MUL_GENERIC)
// Device form of fq_mul: product-scanning (Comba) Montgomery multiplication on 8 x 32-bit words with a 96-bit column
// accumulator, written around the one instruction the hardware has for this — v_mad_u64_u32 (32 x 32 + 64 -> 64, carry-out
// into an SGPR pair). A term of a column is mad + add-with-carry; the compiler's own code for the u128 formulation below
// never uses the carry-out and spends 350 of its 517 instructions (100 of them multiplies) shuffling 32-bit halves so that
// no addition can overflow. Here: 64 + 40 terms (q = 2^252 + c has three zero words, and its top word is a single bit),
// 8 word inverses, 15 column shifts — ~330 issue slots. Two wait states separate a VALU write of VCC from a VALU read of it
// on gfx9: the s_nop sits inside the asm statement, where the hazard recognizer does not look.
// Same value as the generic form (both return the canonical residue), checked by every parity test.
// lo:hi (64 + 32 bits) += x*y, for 1, 2 or 4 terms per statement. In the 4-term form the carries travel through three SGPR
// pairs (VCC and two the compiler picks) and every add-with-carry sits at least two instructions behind the mad whose
// carry it consumes, so no wait-state padding is needed: two instructions per term.
__device__ __forceinline__ void fq_term1(uint64_t& lo, uint32_t& hi, uint32_t x, uint32_t y) {
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\ts_nop 1\n\tv_addc_co_u32_e32 %1, vcc, 0, %1, vcc" : "+v"(lo), "+v"(hi) : "v"(x), "v"(y) : "vcc");
}
__device__ __forceinline__ void fq_term2(uint64_t& lo, uint32_t& hi, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1) {
  uint64_t c1;
  asm("v_mad_u64_u32 %0, vcc, %3, %4, %0\n\t"
      "v_mad_u64_u32 %0, %2, %5, %6, %0\n\t"
      "s_nop 0\n\t"
      "v_addc_co_u32_e32 %1, vcc, 0, %1, vcc\n\t"
      "v_addc_co_u32_e64 %1, %2, 0, %1, %2"
      : "+v"(lo), "+v"(hi), "=&s"(c1)
      : "v"(x0), "v"(y0), "v"(x1), "v"(y1)
      : "vcc");
}
__device__ __forceinline__ void fq_term4(uint64_t& lo, uint32_t& hi, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1, uint32_t x2, uint32_t y2,
                                         uint32_t x3, uint32_t y3) {
  uint64_t c1, c2;
  asm("v_mad_u64_u32 %0, vcc, %4, %5, %0\n\t"
      "v_mad_u64_u32 %0, %2, %6, %7, %0\n\t"
      "v_mad_u64_u32 %0, %3, %8, %9, %0\n\t"
      "v_addc_co_u32_e32 %1, vcc, 0, %1, vcc\n\t"
      "v_mad_u64_u32 %0, vcc, %10, %11, %0\n\t"
      "v_addc_co_u32_e64 %1, %2, 0, %1, %2\n\t"
      "v_addc_co_u32_e64 %1, %3, 0, %1, %3\n\t"
      "v_addc_co_u32_e32 %1, vcc, 0, %1, vcc"
      : "+v"(lo), "+v"(hi), "=&s"(c1), "=&s"(c2)
      : "v"(x0), "v"(y0), "v"(x1), "v"(y1), "v"(x2), "v"(y2), "v"(x3), "v"(y3)
      : "vcc");
}
template <int N>
__device__ __forceinline__ void fq_terms(uint64_t& lo, uint32_t& hi, const uint32_t (&x)[16], const uint32_t (&y)[16]) {
  static_assert(N >= 0 && N <= 16, "terms per column");
  constexpr int n4 = N / 4 * 4;
#pragma unroll
  for (int t = 0; t < n4; t += 4) fq_term4(lo, hi, x[t], y[t], x[t + 1], y[t + 1], x[t + 2], y[t + 2], x[t + 3], y[t + 3]);
  if (N - n4 >= 2) fq_term2(lo, hi, x[n4], y[n4], x[n4 + 1], y[n4 + 1]);
  if ((N - n4) & 1) fq_term1(lo, hi, x[N - 1], y[N - 1]);
}
// column k of the product-scanning Montgomery multiplication: sum of a_i b_(k-i) and of m_i q_(k-i) over the words that exist
template <int K>
__device__ __forceinline__ void fq_column(uint64_t& lo, uint32_t& hi, const uint32_t (&a)[8], const uint32_t (&b)[8], uint32_t (&m)[8],
                                          const uint32_t (&q)[8], uint32_t (&r)[8]) {
  constexpr int i0 = K < 8 ? 0 : K - 7, i1 = K < 8 ? K : 7;  // a_i b_(K-i) for i0 <= i <= i1
  // reduction terms m_i q_(K-i): i < K for the low columns (m_K is made below), q has words 0..3 and 7 only
  constexpr bool has7 = K >= 7;               // m_(K-7) q_7
  constexpr int j0 = K < 8 ? (K >= 3 ? K - 3 : 0) : (K - 3 > K - 7 ? K - 3 : K - 7);  // m_i q_(K-i) with K-i in 1..3 (and 0 for high columns)
  uint32_t x[16], y[16];
  int n = 0;
#pragma unroll
  for (int i = i0; i <= i1; i++) { x[n] = a[i]; y[n] = b[K - i]; n++; }
  if (has7) { x[n] = m[K - 7]; y[n] = q[7]; n++; }
#pragma unroll
  for (int i = j0; i <= 7; i++) {
    const int w = K - i;  // word of q
    if (w >= (K < 8 ? 1 : 0) && w <= 3) { x[n] = m[i]; y[n] = q[w]; n++; }
  }
  constexpr int nprod = i1 - i0 + 1;
  constexpr int nred = (has7 ? 1 : 0) + (K < 8 ? (K < 3 ? K : 3) : (K <= 10 ? 11 - K : 0));
  fq_terms<nprod + nred>(lo, hi, x, y);
  if (K < 8) {
    uint64_t t;
    asm("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(t) : "v"((uint32_t)lo), "v"((uint32_t)SP_QINV) : "vcc");  // low word of lo * (-q^-1)
    m[K] = (uint32_t)t;
    fq_term1(lo, hi, m[K], q[0]);  // clears the column's low word
  } else {
    r[K - 8] = (uint32_t)lo;
  }
  lo = (lo >> 32) | ((uint64_t)hi << 32);
  hi = 0;
}
__device__ __forceinline__ Fq fq_mul(const Fq& A, const Fq& B) {  // ristretto255.rs:690-714
  uint32_t a[8], b[8], m[8], r[8];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    a[2 * i] = (uint32_t)A.l[i]; a[2 * i + 1] = (uint32_t)(A.l[i] >> 32);
    b[2 * i] = (uint32_t)B.l[i]; b[2 * i + 1] = (uint32_t)(B.l[i] >> 32);
  }
  const uint32_t q[8] = {(uint32_t)SP_Q0, (uint32_t)(SP_Q0 >> 32), (uint32_t)SP_Q1, (uint32_t)(SP_Q1 >> 32), 0, 0, 0, (uint32_t)(SP_Q3 >> 32)};
  uint64_t lo = 0;
  uint32_t hi = 0;
  fq_column<0>(lo, hi, a, b, m, q, r); fq_column<1>(lo, hi, a, b, m, q, r); fq_column<2>(lo, hi, a, b, m, q, r); fq_column<3>(lo, hi, a, b, m, q, r);
  fq_column<4>(lo, hi, a, b, m, q, r); fq_column<5>(lo, hi, a, b, m, q, r); fq_column<6>(lo, hi, a, b, m, q, r); fq_column<7>(lo, hi, a, b, m, q, r);
  fq_column<8>(lo, hi, a, b, m, q, r); fq_column<9>(lo, hi, a, b, m, q, r); fq_column<10>(lo, hi, a, b, m, q, r); fq_column<11>(lo, hi, a, b, m, q, r);
  fq_column<12>(lo, hi, a, b, m, q, r); fq_column<13>(lo, hi, a, b, m, q, r); fq_column<14>(lo, hi, a, b, m, q, r);
  r[7] = (uint32_t)lo;
  Fq t = {{(uint64_t)r[0] | ((uint64_t)r[1] << 32), (uint64_t)r[2] | ((uint64_t)r[3] << 32), (uint64_t)r[4] | ((uint64_t)r[5] << 32),
           (uint64_t)r[6] | ((uint64_t)r[7] << 32)}};
  return fq_csub(t);
}
__host__ inline Fq fq_mul(const Fq& a, const Fq& b) {
#else
SP_HD Fq fq_mul(const Fq& a, const Fq& b) {  // ristretto255.rs:690-714
#endif
  uint64_t r[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < 4; i++) {
    u128 c = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      c += (u128)a.l[i] * b.l[j] + r[i + j];
      r[i + j] = (uint64_t)c;
      c >>= 64;
    }
    r[i + 4] = (uint64_t)c;
  }
  return fq_mont_reduce(r);
}
SP_HD Fq fq_sqr(const Fq& a) { return fq_mul(a, a); }
// Montgomery form -> canonical integer limbs (Scalar::to_bytes, ristretto255.rs:419-431)
SP_HD Fq fq_from_mont(const Fq& a) {
  uint64_t r[8] = {a.l[0], a.l[1], a.l[2], a.l[3], 0, 0, 0, 0};
  return fq_mont_reduce(r);
}
SP_HD Fq fq_to_mont(const Fq& canonical) { return fq_mul(canonical, fq_R2()); }
SP_HD Fq fq_from_u64(uint64_t v) { return fq_mul(Fq{{v, 0, 0, 0}}, fq_R2()); }  // ristretto255.rs:222-226
SP_HD Fq fq_from_u512(const uint64_t w[8]) {                                      // ristretto255.rs:448-466
  Fq d0 = {{w[0], w[1], w[2], w[3]}}, d1 = {{w[4], w[5], w[6], w[7]}};
  return fq_add(fq_mul(d0, fq_R2()), fq_mul(d1, fq_R3()));
}
SP_HD Fq fq_pow(const Fq& a, const uint64_t e[4]) {
  Fq res = fq_one();
  for (int w = 3; w >= 0; w--)
    for (int i = 63; i >= 0; i--) {
      res = fq_sqr(res);
      if ((e[w] >> i) & 1) res = fq_mul(res, a);
    }
  return res;
}
SP_HD Fq fq_invert(const Fq& a) {  // value of ristretto255.rs:541-595: a^(q-2)
  const uint64_t e[4] = {SP_Q0 - 2, SP_Q1, SP_Q2, SP_Q3};
  return fq_pow(a, e);
}

SP_HD Fq operator+(const Fq& a, const Fq& b) { return fq_add(a, b); }
SP_HD Fq operator-(const Fq& a, const Fq& b) { return fq_sub(a, b); }
SP_HD Fq operator*(const Fq& a, const Fq& b) { return fq_mul(a, b); }
SP_HD Fq operator-(const Fq& a) { return fq_neg(a); }
SP_HD bool operator==(const Fq& a, const Fq& b) { return fq_eq(a, b); }
SP_HD bool operator!=(const Fq& a, const Fq& b) { return !fq_eq(a, b); }
SP_HD Fq& operator+=(Fq& a, const Fq& b) { a = fq_add(a, b); return a; }
SP_HD Fq& operator-=(Fq& a, const Fq& b) { a = fq_sub(a, b); return a; }
SP_HD Fq& operator*=(Fq& a, const Fq& b) { a = fq_mul(a, b); return a; }

// ------------------------------------------------------------------ Fp
struct Fp {
  uint64_t v[4];
};

SP_HD Fp fp_zero() { return Fp{{0, 0, 0, 0}}; }
SP_HD Fp fp_one() { return Fp{{1, 0, 0, 0}}; }

// add `c` (< 2^64) * 1 into limb 0 with full carry propagation; returns carry out of limb 3
SP_HD uint64_t fp_add_small(Fp& r, uint64_t x) {
  u128 c = (u128)r.v[0] + x;
  r.v[0] = (uint64_t)c;
  c >>= 64;
#pragma unroll
  for (int i = 1; i < 4; i++) {
    c += r.v[i];
    r.v[i] = (uint64_t)c;
    c >>= 64;
  }
  return (uint64_t)c;
}
#if defined(__HIP_DEVICE_COMPILE__) && !defined(SP_FIELD_ADD_GENERIC)
__device__ __forceinline__ Fp fp_add(const Fp& a, const Fp& b) {  // s = a + b; if it left 2^256: s + 38 (2^256 = 38 mod p); a second wrap adds 38 once more
  uint32_t aw[8], bw[8], s[8], t[8], r[8];
  const uint32_t k[8] = {38u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
  sp_split8(a, aw); sp_split8(b, bw);
  uint64_t cA, cB;
  sp_chain2<2>(aw, bw, k, s, t, cA, cB);
  sp_select8(r, t, s, cA);
  Fp o = sp_join8<Fp>(r);
  o.v[0] += 38 * (uint64_t)sp_both(cA, cB);  // after a second wrap o < 38: cannot carry
  return o;
}
__device__ __forceinline__ Fp fp_sub(const Fp& a, const Fp& b) {  // d = a - b; if it borrowed: d - 38; a second borrow takes 38 once more
  uint32_t aw[8], bw[8], d[8], e[8], r[8];
  const uint32_t k[8] = {38u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
  sp_split8(a, aw); sp_split8(b, bw);
  uint64_t cA, cB;
  sp_chain2<3>(aw, bw, k, d, e, cA, cB);
  sp_select8(r, e, d, cA);
  Fp o = sp_join8<Fp>(r);
  o.v[0] -= 38 * (uint64_t)sp_both(cA, cB);  // after a second wrap o >= 2^256 - 38: cannot borrow
  return o;
}
__host__ inline Fp fp_add(const Fp& a, const Fp& b) {
#else
SP_HD Fp fp_add(const Fp& a, const Fp& b) {
#endif
  Fp r;
  u128 c = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    c += (u128)a.v[i] + b.v[i];
    r.v[i] = (uint64_t)c;
    c >>= 64;
  }
  uint64_t c2 = fp_add_small(r, 38 * (uint64_t)c);  // 2^256 = 38
  r.v[0] += 38 * c2;                                // second wrap leaves r < 38: cannot carry
  return r;
}
#if defined(__HIP_DEVICE_COMPILE__) && !defined(SP_FIELD_ADD_GENERIC)
__host__ inline Fp fp_sub(const Fp& a, const Fp& b) {
#else
SP_HD Fp fp_sub(const Fp& a, const Fp& b) {
#endif
  Fp r;
  uint64_t borrow = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    u128 x = (u128)a.v[i] - b.v[i] - borrow;
    r.v[i] = (uint64_t)x;
    borrow = (uint64_t)(x >> 64) & 1;
  }
  // a - b + 2^256 = a - b + 38 (mod p): take 38 back out
  uint64_t s = 38 * borrow, b2 = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    u128 x = (u128)r.v[i] - (i == 0 ? s : 0) - b2;
    r.v[i] = (uint64_t)x;
    b2 = (uint64_t)(x >> 64) & 1;
  }
  r.v[0] -= 38 * b2;  // r wrapped to >= 2^256-38: subtracting 38 again cannot borrow
  return r;
}
SP_HD Fp fp_neg(const Fp& a) { return fp_sub(fp_zero(), a); }

SP_HD Fp fp_reduce512(const uint64_t t[8]) {
  // lo + 38*hi ; written with a 64-bit running carry (this exact shape measured fastest in bench/ubench_fpmul.hip)
  Fp r;
  uint64_t c = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    u128 x = (u128)t[i + 4] * 38u + t[i] + c;
    r.v[i] = (uint64_t)x;
    c = (uint64_t)(x >> 64);
  }
  u128 x = (u128)c * 38u + r.v[0];  // c < 39
  r.v[0] = (uint64_t)x;
  uint64_t cc = (uint64_t)(x >> 64);
#pragma unroll
  for (int i = 1; i < 4; i++) {
    u128 s = (u128)r.v[i] + cc;
    r.v[i] = (uint64_t)s;
    cc = (uint64_t)(s >> 64);
  }
  r.v[0] += cc * 38u;  // a second wrap leaves r < 2^64: cannot carry
  return r;
}
#if defined(__HIP_DEVICE_COMPILE__) && !defined(SP_FP_MUL_GENERIC)
// Device form of fp_mul, same construction as fq_mul above (32-bit words, 96-bit column accumulator, v_mad_u64_u32 with its
// carry-out, two instructions per term): the seven high columns of the 8 x 8 product first (words h[0..7], exact: they carry
// no carries of the low half), then the eight low columns with the extra term 38 * h[k] (2^256 = 38 mod p), then the
// carry word of the low half folded in the same way. 72 terms, ~210 instructions against 317 (+89 s_nop) compiled from the
// u128 form; a mixed point addition is seven of these. The result is a weakly reduced representative (< 2^256) of the same
// residue; every consumer reduces before it compares or encodes.
template <int K>
__device__ __forceinline__ void fp_column(uint64_t& lo, uint32_t& hi, const uint32_t (&a)[8], const uint32_t (&b)[8], const uint32_t (&h)[8]) {
  constexpr int i0 = K < 8 ? 0 : K - 7, i1 = K < 8 ? K : 7;
  uint32_t x[16], y[16];
  int n = 0;
#pragma unroll
  for (int i = i0; i <= i1; i++) { x[n] = a[i]; y[n] = b[K - i]; n++; }
  if (K < 8) { x[n] = h[K]; y[n] = 38u; n++; }
  fq_terms<(i1 - i0 + 1) + (K < 8 ? 1 : 0)>(lo, hi, x, y);
}
__device__ __forceinline__ Fp fp_mul(const Fp& A, const Fp& B) {
  uint32_t a[8], b[8], h[8], r[8];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    a[2 * i] = (uint32_t)A.v[i]; a[2 * i + 1] = (uint32_t)(A.v[i] >> 32);
    b[2 * i] = (uint32_t)B.v[i]; b[2 * i + 1] = (uint32_t)(B.v[i] >> 32);
  }
  uint64_t lo = 0;
  uint32_t hi = 0;
#define SP_FP_HI(K)                              \
  fp_column<K>(lo, hi, a, b, h);                 \
  h[K - 8] = (uint32_t)lo;                       \
  lo = (lo >> 32) | ((uint64_t)hi << 32);        \
  hi = 0;
  SP_FP_HI(8) SP_FP_HI(9) SP_FP_HI(10) SP_FP_HI(11) SP_FP_HI(12) SP_FP_HI(13) SP_FP_HI(14)
#undef SP_FP_HI
  h[7] = (uint32_t)lo;
  lo = 0;
#define SP_FP_LO(K)                              \
  fp_column<K>(lo, hi, a, b, h);                 \
  r[K] = (uint32_t)lo;                           \
  lo = (lo >> 32) | ((uint64_t)hi << 32);        \
  hi = 0;
  SP_FP_LO(0) SP_FP_LO(1) SP_FP_LO(2) SP_FP_LO(3) SP_FP_LO(4) SP_FP_LO(5) SP_FP_LO(6) SP_FP_LO(7)
#undef SP_FP_LO
  Fp o = {{(uint64_t)r[0] | ((uint64_t)r[1] << 32), (uint64_t)r[2] | ((uint64_t)r[3] << 32), (uint64_t)r[4] | ((uint64_t)r[5] << 32),
           (uint64_t)r[6] | ((uint64_t)r[7] << 32)}};
  // the low half overflowed 2^256 by lo (< 2^40): fold 38 * lo back in; a second wrap leaves o < 2^64, so it cannot carry again
  uint64_t c2 = fp_add_small(o, 38 * lo);
  o.v[0] += 38 * c2;
  return o;
}
__host__ inline Fp fp_mul(const Fp& a, const Fp& b) {
#else
SP_HD Fp fp_mul(const Fp& a, const Fp& b) {
#endif
  uint64_t t[8];
#pragma unroll
  for (int i = 0; i < 8; i++) t[i] = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    uint64_t c = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      u128 x = (u128)a.v[i] * b.v[j] + t[i + j] + c;
      t[i + j] = (uint64_t)x;
      c = (uint64_t)(x >> 64);
    }
    t[i + 4] = c;
  }
  return fp_reduce512(t);
}
// dedicated squaring: 6 doubled cross products + 4 squares (10 wide multiplies instead of 16). The ~254-squaring
// inverse-square-root chain of the ristretto encode is latency-critical on the commit path.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(SP_FP_MUL_GENERIC)
__device__ __forceinline__ Fp fp_sqr(const Fp& a) { return fp_mul(a, a); }  // ~210 instructions; the u128 squaring below compiles to 251 (+61 s_nop)
__host__ inline Fp fp_sqr(const Fp& a) {
#else
SP_HD Fp fp_sqr(const Fp& a) {
#endif
  uint64_t t[8];
  u128 c;
  c = (u128)a.v[0] * a.v[1];
  t[1] = (uint64_t)c; c >>= 64;
  c += (u128)a.v[0] * a.v[2];
  t[2] = (uint64_t)c; c >>= 64;
  c += (u128)a.v[0] * a.v[3];
  t[3] = (uint64_t)c; t[4] = (uint64_t)(c >> 64);
  c = (u128)a.v[1] * a.v[2] + t[3];
  t[3] = (uint64_t)c; c >>= 64;
  c += (u128)a.v[1] * a.v[3] + t[4];
  t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
  c = (u128)a.v[2] * a.v[3] + t[5];
  t[5] = (uint64_t)c; t[6] = (uint64_t)(c >> 64);
  // double the cross terms
  t[7] = t[6] >> 63;
  t[6] = (t[6] << 1) | (t[5] >> 63);
  t[5] = (t[5] << 1) | (t[4] >> 63);
  t[4] = (t[4] << 1) | (t[3] >> 63);
  t[3] = (t[3] << 1) | (t[2] >> 63);
  t[2] = (t[2] << 1) | (t[1] >> 63);
  t[1] = t[1] << 1;
  // add the squares a_i^2 at limb 2i
  c = (u128)a.v[0] * a.v[0];
  t[0] = (uint64_t)c; c >>= 64;
  c += t[1]; t[1] = (uint64_t)c; c >>= 64;
  c += (u128)a.v[1] * a.v[1] + t[2];
  t[2] = (uint64_t)c; c >>= 64;
  c += t[3]; t[3] = (uint64_t)c; c >>= 64;
  c += (u128)a.v[2] * a.v[2] + t[4];
  t[4] = (uint64_t)c; c >>= 64;
  c += t[5]; t[5] = (uint64_t)c; c >>= 64;
  c += (u128)a.v[3] * a.v[3] + t[6];
  t[6] = (uint64_t)c; c >>= 64;
  t[7] += (uint64_t)c;
  return fp_reduce512(t);
}
// canonical representative in [0,p)
SP_HD Fp fp_canon(const Fp& a) {
  Fp t = a;
  // fold bit 255: t = (t mod 2^255) + 19*(t >> 255)   -> t < 2^255 + 19
  uint64_t top = t.v[3] >> 63;
  t.v[3] &= 0x7fffffffffffffffULL;
  fp_add_small(t, 19 * top);
  // now t < 2^255 + 19 ; subtract p if t >= p  <=>  (t + 19) has bit 255 set
  Fp u = t;
  fp_add_small(u, 19);
  uint64_t ge = u.v[3] >> 63;
  u.v[3] &= 0x7fffffffffffffffULL;
  Fp r;
#pragma unroll
  for (int i = 0; i < 4; i++) r.v[i] = ge ? u.v[i] : t.v[i];
  return r;
}
SP_HD void fp_to_bytes(const Fp& a, uint8_t out[32]) {
  Fp c = fp_canon(a);
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int k = 0; k < 8; k++) out[8 * i + k] = (uint8_t)(c.v[i] >> (8 * k));
}
SP_HD Fp fp_from_bytes(const uint8_t b[32]) {  // top bit masked (value mod 2^255), as dalek/RFC 9496
  Fp r;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    uint64_t w = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) w |= (uint64_t)b[8 * i + k] << (8 * k);
    r.v[i] = w;
  }
  r.v[3] &= 0x7fffffffffffffffULL;
  return r;
}
SP_HD bool fp_is_zero(const Fp& a) {
  Fp c = fp_canon(a);
  return (c.v[0] | c.v[1] | c.v[2] | c.v[3]) == 0;
}
SP_HD bool fp_eq(const Fp& a, const Fp& b) { return fp_is_zero(fp_sub(a, b)); }
SP_HD bool fp_is_negative(const Fp& a) { return fp_canon(a).v[0] & 1; }  // RFC 9496 §4.1
SP_HD Fp fp_cneg(const Fp& a, bool neg) {
  Fp n = fp_neg(a), r;
#pragma unroll
  for (int i = 0; i < 4; i++) r.v[i] = neg ? n.v[i] : a.v[i];
  return r;
}
SP_HD Fp fp_abs(const Fp& a) { return fp_cneg(a, fp_is_negative(a)); }
SP_HD Fp fp_select(const Fp& a, const Fp& b, bool take_b) {
  Fp r;
#pragma unroll
  for (int i = 0; i < 4; i++) r.v[i] = take_b ? b.v[i] : a.v[i];
  return r;
}
SP_HD Fp fp_pow2k(Fp a, int k) {
  for (int i = 0; i < k; i++) a = fp_sqr(a);
  return a;
}
// z^(2^250-1) and z^11, shared by inversion and the (p-5)/8 power
SP_HD void fp_pow_ladder(const Fp& z, Fp* z2_250_0, Fp* z11) {
  Fp z2 = fp_sqr(z);
  Fp z9 = fp_mul(fp_pow2k(z2, 2), z);
  *z11 = fp_mul(z9, z2);
  Fp z2_5_0 = fp_mul(fp_sqr(*z11), z9);
  Fp z2_10_0 = fp_mul(fp_pow2k(z2_5_0, 5), z2_5_0);
  Fp z2_20_0 = fp_mul(fp_pow2k(z2_10_0, 10), z2_10_0);
  Fp z2_40_0 = fp_mul(fp_pow2k(z2_20_0, 20), z2_20_0);
  Fp z2_50_0 = fp_mul(fp_pow2k(z2_40_0, 10), z2_10_0);
  Fp z2_100_0 = fp_mul(fp_pow2k(z2_50_0, 50), z2_50_0);
  Fp z2_200_0 = fp_mul(fp_pow2k(z2_100_0, 100), z2_100_0);
  *z2_250_0 = fp_mul(fp_pow2k(z2_200_0, 50), z2_50_0);
}
SP_HD Fp fp_invert(const Fp& z) {  // z^(p-2)
  Fp t, z11;
  fp_pow_ladder(z, &t, &z11);
  return fp_mul(fp_pow2k(t, 5), z11);
}
SP_HD Fp fp_pow_p58(const Fp& z) {  // z^((p-5)/8)
  Fp t, z11;
  fp_pow_ladder(z, &t, &z11);
  return fp_mul(fp_pow2k(t, 2), z);
}

}  // namespace sp
