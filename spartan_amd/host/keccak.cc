// spartan_amd host driver: Keccak-f[1600] (FIPS 202 §3.2-3.4), the permutation under Merlin's STROBE-128 and SHAKE256.
// Own translation unit so it can be built with g++ while the field-arithmetic-heavy driver is built with clang++
// (see transcript.hpp). State kept in 25 locals (A[x + 5y]); theta, rho+pi and chi written out per lane.
#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <cstring>

namespace spz {

static inline uint64_t rotl64(uint64_t x, int n) { return (x << n) | (x >> (64 - n)); }  // n in 1..63

static const uint64_t RC[24] = {0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
                               0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
                               0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
                               0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
                               0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
                               0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};

// The permutation is compiled twice and the variant is picked once, BY FEATURE: ANDN / RORX (BMI1/2) make chi and rho cheaper
// (418 -> 300 ns per permutation on the build host's Xeon). Rounds 2-3 used GCC's target_clones("default", "arch=haswell") for this;
// its resolver matches the CPU *model* (__builtin_cpu_is), so on anything that is not literally a Haswell — the GPU box's EPYC
// included — the plain clone ran. A proof runs ~9 500 permutations on the proving thread's critical path.
__attribute__((always_inline)) static inline void keccak_f1600_body(uint64_t A[25]) {
  uint64_t a00 = A[0], a10 = A[1], a20 = A[2], a30 = A[3], a40 = A[4], a01 = A[5], a11 = A[6], a21 = A[7], a31 = A[8], a41 = A[9], a02 = A[10], a12 = A[11], a22 = A[12], a32 = A[13], a42 = A[14], a03 = A[15], a13 = A[16], a23 = A[17], a33 = A[18], a43 = A[19], a04 = A[20], a14 = A[21], a24 = A[22], a34 = A[23], a44 = A[24];
  for (int round = 0; round < 24; round++) {
    uint64_t c0 = a00 ^ a01 ^ a02 ^ a03 ^ a04, c1 = a10 ^ a11 ^ a12 ^ a13 ^ a14, c2 = a20 ^ a21 ^ a22 ^ a23 ^ a24, c3 = a30 ^ a31 ^ a32 ^ a33 ^ a34, c4 = a40 ^ a41 ^ a42 ^ a43 ^ a44;
    uint64_t d0 = c4 ^ rotl64(c1, 1), d1 = c0 ^ rotl64(c2, 1), d2 = c1 ^ rotl64(c3, 1), d3 = c2 ^ rotl64(c4, 1), d4 = c3 ^ rotl64(c0, 1);
    uint64_t b00 = (a00 ^ d0), b10 = rotl64((a11 ^ d1), 44), b20 = rotl64((a22 ^ d2), 43), b30 = rotl64((a33 ^ d3), 21), b40 = rotl64((a44 ^ d4), 14), b01 = rotl64((a30 ^ d3), 28), b11 = rotl64((a41 ^ d4), 20), b21 = rotl64((a02 ^ d0), 3), b31 = rotl64((a13 ^ d1), 45), b41 = rotl64((a24 ^ d2), 61), b02 = rotl64((a10 ^ d1), 1), b12 = rotl64((a21 ^ d2), 6), b22 = rotl64((a32 ^ d3), 25), b32 = rotl64((a43 ^ d4), 8), b42 = rotl64((a04 ^ d0), 18), b03 = rotl64((a40 ^ d4), 27), b13 = rotl64((a01 ^ d0), 36), b23 = rotl64((a12 ^ d1), 10), b33 = rotl64((a23 ^ d2), 15), b43 = rotl64((a34 ^ d3), 56), b04 = rotl64((a20 ^ d2), 62), b14 = rotl64((a31 ^ d3), 55), b24 = rotl64((a42 ^ d4), 39), b34 = rotl64((a03 ^ d0), 41), b44 = rotl64((a14 ^ d1), 2);
    a00 = b00 ^ (~b10 & b20);
    a10 = b10 ^ (~b20 & b30);
    a20 = b20 ^ (~b30 & b40);
    a30 = b30 ^ (~b40 & b00);
    a40 = b40 ^ (~b00 & b10);
    a01 = b01 ^ (~b11 & b21);
    a11 = b11 ^ (~b21 & b31);
    a21 = b21 ^ (~b31 & b41);
    a31 = b31 ^ (~b41 & b01);
    a41 = b41 ^ (~b01 & b11);
    a02 = b02 ^ (~b12 & b22);
    a12 = b12 ^ (~b22 & b32);
    a22 = b22 ^ (~b32 & b42);
    a32 = b32 ^ (~b42 & b02);
    a42 = b42 ^ (~b02 & b12);
    a03 = b03 ^ (~b13 & b23);
    a13 = b13 ^ (~b23 & b33);
    a23 = b23 ^ (~b33 & b43);
    a33 = b33 ^ (~b43 & b03);
    a43 = b43 ^ (~b03 & b13);
    a04 = b04 ^ (~b14 & b24);
    a14 = b14 ^ (~b24 & b34);
    a24 = b24 ^ (~b34 & b44);
    a34 = b34 ^ (~b44 & b04);
    a44 = b44 ^ (~b04 & b14);
    a00 ^= RC[round];
  }
  A[0] = a00; A[1] = a10; A[2] = a20; A[3] = a30; A[4] = a40;
  A[5] = a01; A[6] = a11; A[7] = a21; A[8] = a31; A[9] = a41;
  A[10] = a02; A[11] = a12; A[12] = a22; A[13] = a32; A[14] = a42;
  A[15] = a03; A[16] = a13; A[17] = a23; A[18] = a33; A[19] = a43;
  A[20] = a04; A[21] = a14; A[22] = a24; A[23] = a34; A[24] = a44;
}
#if defined(__x86_64__) && defined(__GNUC__)
#define SPZ_KECCAK_X86 1
#include <immintrin.h>
__attribute__((target("bmi,bmi2"))) static void keccak_f1600_bmi(uint64_t A[25]) { keccak_f1600_body(A); }

// AVX-512 form: one 512-bit register per plane (register y holds lanes A[x + 5y] at elements x = 0..4).
//   theta   two three-way XORs for the column parities, two lane rotations of the parity register, one XOR3 per plane
//   rho     one variable rotate per plane
//   pi      within a plane: B[X][Y] with X = y, Y = 2x + 3y comes from plane X, element (3Y + X) mod 5 — ONE in-register permutation per
//           plane turns plane X into "column register" X (element Y = B[X][Y])
//   chi     across registers: N_X = C_X ^ (~C_{X+1} & C_{X+2}), one ternary-logic instruction per column register; iota on N_0[0]
//   back to planes: a 5 x 5 transposition of the column registers (4 unpacks, 5 two-source permutations, 5 masked permutations)
// ~45 instructions per round against ~150 for the scalar form. Which form is faster depends on the core (shuffle ports against
// scalar ALUs): keccak_pick() times them once.
alignas(64) static const uint64_t K_PREV[8] = {4, 0, 1, 2, 3, 5, 6, 7}, K_NEXT[8] = {1, 2, 3, 4, 0, 5, 6, 7};
alignas(64) static const uint64_t K_RHO[5][8] = {{0, 1, 62, 28, 27, 0, 0, 0}, {36, 44, 6, 55, 20, 0, 0, 0}, {3, 10, 43, 25, 39, 0, 0, 0},
                                                 {41, 45, 15, 21, 8, 0, 0, 0}, {18, 2, 61, 56, 14, 0, 0, 0}};
// K_PI[X][Y] = (3Y + X) mod 5
alignas(64) static const uint64_t K_PI[5][8] = {{0, 3, 1, 4, 2, 5, 6, 7}, {1, 4, 2, 0, 3, 5, 6, 7}, {2, 0, 3, 1, 4, 5, 6, 7},
                                                {3, 1, 4, 2, 0, 5, 6, 7}, {4, 2, 0, 3, 1, 5, 6, 7}};
// transposition: U01 = unpacklo(N0, N1) = [N0[0], N1[0], N0[2], N1[2], N0[4], N1[4], ..], V01 = unpackhi = [N0[1], N1[1], N0[3], N1[3], ..]
alignas(64) static const uint64_t K_T0[8] = {0, 1, 8, 9, 0, 0, 0, 0}, K_T1[8] = {2, 3, 10, 11, 0, 0, 0, 0}, K_T2[8] = {4, 5, 12, 13, 0, 0, 0, 0};
alignas(64) static const uint64_t K_E[5][8] = {{0, 0, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 1, 0, 0, 0}, {0, 0, 0, 0, 2, 0, 0, 0}, {0, 0, 0, 0, 3, 0, 0, 0},
                                               {0, 0, 0, 0, 4, 0, 0, 0}};
__attribute__((target("avx512f"))) static void keccak_f1600_avx512(uint64_t A[25]) {
  const __m512i prev = _mm512_load_si512(K_PREV), next = _mm512_load_si512(K_NEXT);
  const __m512i rho0 = _mm512_load_si512(K_RHO[0]), rho1 = _mm512_load_si512(K_RHO[1]), rho2 = _mm512_load_si512(K_RHO[2]),
                rho3 = _mm512_load_si512(K_RHO[3]), rho4 = _mm512_load_si512(K_RHO[4]);
  const __m512i pi0 = _mm512_load_si512(K_PI[0]), pi1 = _mm512_load_si512(K_PI[1]), pi2 = _mm512_load_si512(K_PI[2]), pi3 = _mm512_load_si512(K_PI[3]),
                pi4 = _mm512_load_si512(K_PI[4]);
  const __m512i t0 = _mm512_load_si512(K_T0), t1 = _mm512_load_si512(K_T1), t2 = _mm512_load_si512(K_T2);
  const __m512i e0 = _mm512_load_si512(K_E[0]), e1 = _mm512_load_si512(K_E[1]), e2 = _mm512_load_si512(K_E[2]), e3 = _mm512_load_si512(K_E[3]),
                e4 = _mm512_load_si512(K_E[4]);
  __m512i r0 = _mm512_maskz_loadu_epi64(0x1f, A), r1 = _mm512_maskz_loadu_epi64(0x1f, A + 5), r2 = _mm512_maskz_loadu_epi64(0x1f, A + 10),
          r3 = _mm512_maskz_loadu_epi64(0x1f, A + 15), r4 = _mm512_maskz_loadu_epi64(0x1f, A + 20);
  for (int round = 0; round < 24; round++) {
    // theta
    __m512i c = _mm512_ternarylogic_epi64(_mm512_ternarylogic_epi64(r0, r1, r2, 0x96), r3, r4, 0x96);
    __m512i cm = _mm512_permutexvar_epi64(prev, c);                          // C[x-1]
    __m512i cp = _mm512_rol_epi64(_mm512_permutexvar_epi64(next, c), 1);     // rot(C[x+1], 1)
    r0 = _mm512_ternarylogic_epi64(r0, cm, cp, 0x96);
    r1 = _mm512_ternarylogic_epi64(r1, cm, cp, 0x96);
    r2 = _mm512_ternarylogic_epi64(r2, cm, cp, 0x96);
    r3 = _mm512_ternarylogic_epi64(r3, cm, cp, 0x96);
    r4 = _mm512_ternarylogic_epi64(r4, cm, cp, 0x96);
    // rho, then pi inside each plane: column registers c_X[Y] = B[X][Y]
    __m512i c0 = _mm512_permutexvar_epi64(pi0, _mm512_rolv_epi64(r0, rho0));
    __m512i c1 = _mm512_permutexvar_epi64(pi1, _mm512_rolv_epi64(r1, rho1));
    __m512i c2 = _mm512_permutexvar_epi64(pi2, _mm512_rolv_epi64(r2, rho2));
    __m512i c3 = _mm512_permutexvar_epi64(pi3, _mm512_rolv_epi64(r3, rho3));
    __m512i c4 = _mm512_permutexvar_epi64(pi4, _mm512_rolv_epi64(r4, rho4));
    // chi across the column registers, iota
    __m512i n0 = _mm512_ternarylogic_epi64(c0, c1, c2, 0xD2);
    __m512i n1 = _mm512_ternarylogic_epi64(c1, c2, c3, 0xD2);
    __m512i n2 = _mm512_ternarylogic_epi64(c2, c3, c4, 0xD2);
    __m512i n3 = _mm512_ternarylogic_epi64(c3, c4, c0, 0xD2);
    __m512i n4 = _mm512_ternarylogic_epi64(c4, c0, c1, 0xD2);
    n0 = _mm512_xor_si512(n0, _mm512_maskz_set1_epi64(0x01, (long long)RC[round]));
    // planes again: r_Y[X] = n_X[Y]
    __m512i u01 = _mm512_unpacklo_epi64(n0, n1), v01 = _mm512_unpackhi_epi64(n0, n1);
    __m512i u23 = _mm512_unpacklo_epi64(n2, n3), v23 = _mm512_unpackhi_epi64(n2, n3);
    r0 = _mm512_mask_permutexvar_epi64(_mm512_permutex2var_epi64(u01, t0, u23), 0x10, e0, n4);
    r1 = _mm512_mask_permutexvar_epi64(_mm512_permutex2var_epi64(v01, t0, v23), 0x10, e1, n4);
    r2 = _mm512_mask_permutexvar_epi64(_mm512_permutex2var_epi64(u01, t1, u23), 0x10, e2, n4);
    r3 = _mm512_mask_permutexvar_epi64(_mm512_permutex2var_epi64(v01, t1, v23), 0x10, e3, n4);
    r4 = _mm512_mask_permutexvar_epi64(_mm512_permutex2var_epi64(u01, t2, u23), 0x10, e4, n4);
  }
  _mm512_mask_storeu_epi64(A, 0x1f, r0); _mm512_mask_storeu_epi64(A + 5, 0x1f, r1); _mm512_mask_storeu_epi64(A + 10, 0x1f, r2);
  _mm512_mask_storeu_epi64(A + 15, 0x1f, r3); _mm512_mask_storeu_epi64(A + 20, 0x1f, r4);
}
#endif
static void keccak_f1600_plain(uint64_t A[25]) { keccak_f1600_body(A); }
typedef void (*keccak_fn)(uint64_t[25]);
struct KeccakChoice { keccak_fn fn; const char* name; };
static double keccak_time(keccak_fn f) {  // ns per permutation, best of three short runs
  uint64_t st[25];
  for (int i = 0; i < 25; i++) st[i] = 0x9e3779b97f4a7c15ULL * (uint64_t)(i + 1);
  double best = 1e30;
  for (int rep = 0; rep < 3; rep++) {
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < 400; i++) f(st);
    double ns = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e9 / 400;
    if (ns < best) best = ns;
  }
  return best + (st[0] == 1 ? 1e-9 : 0);  // keep the result alive
}
struct sp_ctx;
#include "../../include/spartan_hip.h"  // sp_ctx_get_option
static KeccakChoice keccak_pick() {
  KeccakChoice cands[3];
  int n = 0;
  cands[n++] = {keccak_f1600_plain, "plain"};
#ifdef SPZ_KECCAK_X86
  __builtin_cpu_init();
  if (__builtin_cpu_supports("bmi") && __builtin_cpu_supports("bmi2")) cands[n++] = {keccak_f1600_bmi, "bmi2"};
  if (__builtin_cpu_supports("avx512f")) cands[n++] = {keccak_f1600_avx512, "avx512"};
#endif
  int64_t want = 0;  // process-wide library option host.keccak: 0 = by calibration, 1 plain, 2 bmi2, 3 avx512 (ignored if this CPU cannot run it)
  if (sp_ctx_get_option(nullptr, "host.keccak", &want) == 0 && want >= 1 && want <= 3) {
    static const char* names[4] = {"", "plain", "bmi2", "avx512"};
    for (int i = 0; i < n; i++) if (!strcmp(names[want], cands[i].name)) return cands[i];
  }
  int best = 0;
  double bt = 1e30;
  for (int i = 0; i < n; i++) {
    double t = keccak_time(cands[i].fn);
    if (t < bt) { bt = t; best = i; }
  }
  return cands[best];
}
static const KeccakChoice& keccak_chosen() {
  static const KeccakChoice c = keccak_pick();  // function-local: safe whatever the order of static initialisers across translation units
  return c;
}
void keccak_f1600_impl(uint64_t A[25]) { keccak_chosen().fn(A); }
// which variant runs (tests, bench.py's host line)
const char* keccak_f1600_variant() { return keccak_chosen().name; }
// one named variant on a caller's state (tests): 0 if this CPU cannot run it
int keccak_f1600_run_variant(const char* name, uint64_t A[25]) {
  if (!strcmp(name, "plain")) { keccak_f1600_plain(A); return 1; }
#ifdef SPZ_KECCAK_X86
  __builtin_cpu_init();
  if (!strcmp(name, "bmi2") && __builtin_cpu_supports("bmi") && __builtin_cpu_supports("bmi2")) { keccak_f1600_bmi(A); return 1; }
  if (!strcmp(name, "avx512") && __builtin_cpu_supports("avx512f")) { keccak_f1600_avx512(A); return 1; }
#endif
  return 0;
}

}  // namespace spz
