// spartan_amd host driver: Keccak-f[1600] (FIPS 202 §3.2-3.4), the permutation under Merlin's STROBE-128 and SHAKE256.
// Own translation unit so it can be built with g++ while the field-arithmetic-heavy driver is built with clang++
// (see transcript.hpp). State kept in 25 locals (A[x + 5y]); theta, rho+pi and chi written out per lane.
#include <cstdint>

namespace spz {

static inline uint64_t rotl64(uint64_t x, int n) { return (x << n) | (x >> (64 - n)); }  // n in 1..63

// two clones, picked once at load time by the dynamic linker: ANDN/RORX (BMI1/2) make chi and rho cheaper
#if defined(__GNUC__) && !defined(__clang__)
__attribute__((target_clones("default", "arch=haswell")))
#endif
void keccak_f1600_impl(uint64_t A[25]) {
  static const uint64_t RC[24] = {0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
                                  0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
                                  0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
                                  0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
                                  0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
                                  0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
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

}  // namespace spz
