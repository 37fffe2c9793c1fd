// spartan_amd host driver: Fiat–Shamir layer.
// Mirrors src/transcript.rs:13-63 (ProofTranscript, AppendToTranscript) and src/random.rs:10-28 (RandomTape)
// over Merlin 1.0 (STROBE-128 / Keccak-f[1600]); SHAKE256 for MultiCommitGens::new (src/commitments.rs:16-24).
// The transcript stays on the host exactly as it stays in Rust in the drop-in design.
#pragma once
#include <chrono>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../csrc/field.hpp"

namespace spz {
using sp::Fq;

inline uint64_t rotl64(uint64_t x, int n) { return (x << n) | (x >> (64 - n)); }  // n in 1..63

// Keccak-f[1600], state kept in 25 locals (A[x + 5y]); theta, rho+pi and chi written out per lane.
#ifdef SPZ_HOSTPROF
struct KeccakProf { uint64_t n = 0; double t = 0; };
inline KeccakProf& keccak_prof() { static KeccakProf p; return p; }
inline void keccak_f1600_impl(uint64_t A[25]);
inline void keccak_f1600(uint64_t A[25]) {
  auto t0 = std::chrono::steady_clock::now();
  keccak_f1600_impl(A);
  keccak_prof().t += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  keccak_prof().n++;
}
inline void keccak_f1600_impl(uint64_t A[25]) {
#else
inline void keccak_f1600(uint64_t A[25]) {
#endif
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

class Shake256 {  // FIPS 202, rate 136, suffix 0x1f
 public:
  Shake256() : pos_(0), squeezing_(false) { memset(st_, 0, sizeof st_); }
  void absorb(const void* data, size_t n) {
    const uint8_t* d = (const uint8_t*)data;
    uint8_t* s = (uint8_t*)st_;
    for (size_t i = 0; i < n; i++) {
      s[pos_++] ^= d[i];
      if (pos_ == 136) { keccak_f1600(st_); pos_ = 0; }
    }
  }
  void squeeze(uint8_t* out, size_t n) {
    uint8_t* s = (uint8_t*)st_;
    if (!squeezing_) {
      s[pos_] ^= 0x1f;
      s[135] ^= 0x80;
      keccak_f1600(st_);
      pos_ = 0;
      squeezing_ = true;
    }
    for (size_t i = 0; i < n; i++) {
      if (pos_ == 136) { keccak_f1600(st_); pos_ = 0; }
      out[i] = s[pos_++];
    }
  }

 private:
  uint64_t st_[25];
  size_t pos_;
  bool squeezing_;
};

class Strobe128 {  // STROBE v1.0.2, the subset Merlin uses (AD, meta-AD, PRF), R = 166
 public:
  explicit Strobe128(const char* protocol) : pos_(0), pos_begin_(0), cur_flags_(0) {
    memset(st_, 0, sizeof st_);
    uint8_t* s = (uint8_t*)st_;
    const uint8_t init[6] = {1, 168, 1, 0, 1, 96};
    memcpy(s, init, 6);
    memcpy(s + 6, "STROBEv1.0.2", 12);
    keccak_f1600(st_);
    meta_ad((const uint8_t*)protocol, strlen(protocol), false);
  }
  void meta_ad(const uint8_t* d, size_t n, bool more) { begin_op(FLAG_M | FLAG_A, more); absorb(d, n); }
  void ad(const uint8_t* d, size_t n, bool more) { begin_op(FLAG_A, more); absorb(d, n); }
  void prf(uint8_t* out, size_t n, bool more) { begin_op(FLAG_I | FLAG_A | FLAG_C, more); squeeze(out, n); }

 private:
  static constexpr uint8_t FLAG_I = 1, FLAG_A = 2, FLAG_C = 4, FLAG_T = 8, FLAG_M = 16, FLAG_K = 32;
  static constexpr int R = 166;
  void run_f() {
    uint8_t* s = (uint8_t*)st_;
    s[pos_] ^= pos_begin_;
    s[pos_ + 1] ^= 0x04;
    s[R + 1] ^= 0x80;
    keccak_f1600(st_);
    pos_ = 0;
    pos_begin_ = 0;
  }
  // duplex in runs of up to R - pos bytes (a proof absorbs ~1.3 MB: every commitment share and the vectors of the
  // inner-product arguments), the permutation runs exactly where the byte-at-a-time definition runs it
  void absorb(const uint8_t* d, size_t n) {
    uint8_t* s = (uint8_t*)st_;
    while (n) {
      size_t k = (size_t)R - pos_;
      if (k > n) k = n;
      uint8_t* dst = s + pos_;
      for (size_t i = 0; i < k; i++) dst[i] ^= d[i];
      pos_ = (uint8_t)(pos_ + k);
      d += k;
      n -= k;
      if (pos_ == R) run_f();
    }
  }
  void squeeze(uint8_t* out, size_t n) {
    uint8_t* s = (uint8_t*)st_;
    while (n) {
      size_t k = (size_t)R - pos_;
      if (k > n) k = n;
      memcpy(out, s + pos_, k);
      memset(s + pos_, 0, k);
      pos_ = (uint8_t)(pos_ + k);
      out += k;
      n -= k;
      if (pos_ == R) run_f();
    }
  }
  void begin_op(uint8_t flags, bool more) {
    if (more) return;  // continuation of the current operation (flags must match; Merlin guarantees it)
    uint8_t old_begin = pos_begin_;
    pos_begin_ = pos_ + 1;
    cur_flags_ = flags;
    uint8_t hdr[2] = {old_begin, flags};
    absorb(hdr, 2);
    bool force_f = (flags & (FLAG_C | FLAG_K)) != 0;
    if (force_f && pos_ != 0) run_f();
  }
  uint64_t st_[25];
  uint8_t pos_, pos_begin_, cur_flags_;
};

class Transcript {  // merlin::Transcript + libspartan's ProofTranscript trait
 public:
  explicit Transcript(const char* label) : s_("Merlin v1.0") { append_message("dom-sep", (const uint8_t*)label, strlen(label)); }
  void append_message(const char* label, const uint8_t* msg, size_t n) {
    uint8_t len[4] = {(uint8_t)n, (uint8_t)(n >> 8), (uint8_t)(n >> 16), (uint8_t)(n >> 24)};
    s_.meta_ad((const uint8_t*)label, strlen(label), false);
    s_.meta_ad(len, 4, true);
    s_.ad(msg, n, false);
  }
  void append_message(const char* label, const char* msg) { append_message(label, (const uint8_t*)msg, strlen(msg)); }
  void append_u64(const char* label, uint64_t x) {
    uint8_t b[8];
    for (int i = 0; i < 8; i++) b[i] = (uint8_t)(x >> (8 * i));
    append_message(label, b, 8);
  }
  void challenge_bytes(const char* label, uint8_t* out, size_t n) {
    uint8_t len[4] = {(uint8_t)n, (uint8_t)(n >> 8), (uint8_t)(n >> 16), (uint8_t)(n >> 24)};
    s_.meta_ad((const uint8_t*)label, strlen(label), false);
    s_.meta_ad(len, 4, true);
    s_.prf(out, n, false);
  }
  // transcript.rs:14-36
  void append_protocol_name(const char* name) { append_message("protocol-name", name); }
  void append_scalar(const char* label, const Fq& x) {
    Fq c = sp::fq_from_mont(x);  // Scalar::to_bytes: canonical little-endian
    uint8_t b[32];
    memcpy(b, c.l, 32);
    append_message(label, b, 32);
  }
  void append_point(const char* label, const uint8_t pt[32]) { append_message(label, pt, 32); }
  Fq challenge_scalar(const char* label) {
    uint8_t buf[64];
    challenge_bytes(label, buf, 64);
    uint64_t w[8];
    memcpy(w, buf, 64);
    return sp::fq_from_u512(w);  // Scalar::from_bytes_wide
  }
  std::vector<Fq> challenge_vector(const char* label, size_t len) {
    std::vector<Fq> v(len);
    for (size_t i = 0; i < len; i++) v[i] = challenge_scalar(label);
    return v;
  }
  // transcript.rs:49-57
  void append_scalars(const char* label, const Fq* v, size_t n) {
    append_message(label, "begin_append_vector");
    for (size_t i = 0; i < n; i++) append_scalar(label, v[i]);
    append_message(label, "end_append_vector");
  }
  void append_scalars(const char* label, const std::vector<Fq>& v) { append_scalars(label, v.data(), v.size()); }

 private:
  Strobe128 s_;
};

// random.rs:10-28. The reference seeds from OsRng (random.rs:13-15); `new_with_seed` is the determinism hook
// the parity contract needs on both sides.
class RandomTape {
 public:
  RandomTape(const char* name, const Fq& seed) : tape_(name) { tape_.append_scalar("init_randomness", seed); }
  Fq random_scalar(const char* label) { return tape_.challenge_scalar(label); }
  std::vector<Fq> random_vector(const char* label, size_t len) { return tape_.challenge_vector(label, len); }

 private:
  Transcript tape_;
};

}  // namespace spz
