// spartan_amd host driver: Fiat–Shamir layer.
// Mirrors src/transcript.rs:13-63 (ProofTranscript, AppendToTranscript) and src/random.rs:10-28 (RandomTape)
// over Merlin 1.0 (STROBE-128 / Keccak-f[1600]); SHAKE256 for MultiCommitGens::new (src/commitments.rs:16-24).
// The transcript stays on the host exactly as it stays in Rust in the drop-in design.
#pragma once
#include <sys/random.h>

#include <cerrno>
#include <chrono>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../csrc/field.hpp"

namespace spz {
using sp::Fq;

// Keccak-f[1600] lives in keccak.cc: that file is built with g++ (its scheduling of the permutation is ~25% faster than
// clang's here, and it carries a BMI clone), the rest of the driver with clang++ (whose 4x64 Montgomery code is 1.5-2x
// faster than g++'s). SPZ_HOSTPROF (diagnostic build) counts and times the calls.
void keccak_f1600_impl(uint64_t A[25]);
const char* keccak_f1600_variant();                            // "plain" | "bmi2" | "avx512": the form picked for this CPU (keccak.cc)
int keccak_f1600_run_variant(const char* name, uint64_t A[25]);  // tests: 0 if this CPU cannot run it
#ifdef SPZ_HOSTPROF
struct KeccakProf { uint64_t n = 0; double t = 0; };
inline KeccakProf& keccak_prof() { static KeccakProf p; return p; }
inline void keccak_f1600(uint64_t A[25]) {
  auto t0 = std::chrono::steady_clock::now();
  keccak_f1600_impl(A);
  keccak_prof().t += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  keccak_prof().n++;
}
#else
inline void keccak_f1600(uint64_t A[25]) { keccak_f1600_impl(A); }
#endif

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
  // The whole state of a merlin::Transcript is its Strobe128 { state: [u8; 200], pos, pos_begin, cur_flags } (merlin 3.0.0,
  // src/strobe.rs): 203 bytes. Exchanging them is how a caller-owned `&mut Transcript` (src/lib.rs:339-347) is continued by
  // the library and handed back (spz_snark_prove_t / spz_nizk_prove_t, host_capi.cc).
  static constexpr size_t STATE_BYTES = 203;
  struct FromState {};
  Strobe128(FromState, const uint8_t in[STATE_BYTES]) : pos_(in[200]), pos_begin_(in[201]), cur_flags_(in[202]) {
    memcpy(st_, in, 200);
    if (pos_ >= R || pos_begin_ > R) throw std::runtime_error("transcript state: position out of range");
  }
  void export_state(uint8_t out[STATE_BYTES]) const { memcpy(out, st_, 200); out[200] = pos_; out[201] = pos_begin_; out[202] = cur_flags_; }

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
  static constexpr size_t STATE_BYTES = Strobe128::STATE_BYTES;
  struct FromState {};
  Transcript(FromState, const uint8_t state[STATE_BYTES]) : s_(Strobe128::FromState(), state) {}  // continue a caller-owned transcript
  void export_state(uint8_t out[STATE_BYTES]) const { s_.export_state(out); }
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

// random.rs:10-28. RandomTape::new seeds the tape with Scalar::random(&mut OsRng) (random.rs:13-15): the one-argument
// constructor does the same (64 bytes from getrandom(2) -> from_bytes_wide), and is what the provers use when the caller
// passes no seed. The seeded constructor is `new_with_seed`, the determinism hook the parity tests need on both sides
// (SURVEY.md fact 1): a seed fixes every blind of the proof, so outside tests it must be secret, carry
// >= 256 bits of entropy and never be used for two proofs — otherwise zero-knowledge is lost.
class RandomTape {
 public:
  explicit RandomTape(const char* name) : tape_(name) { tape_.append_scalar("init_randomness", os_random_scalar()); }
  RandomTape(const char* name, const Fq& seed) : tape_(name) { tape_.append_scalar("init_randomness", seed); }
  static Fq os_random_scalar() {  // Scalar::random (ristretto255.rs:374-380) over the OS entropy source
    uint8_t buf[64];
    size_t got = 0;
    while (got < sizeof buf) {
      ssize_t n = getrandom(buf + got, sizeof buf - got, 0);
      if (n < 0) {
        if (errno == EINTR) continue;
        throw std::runtime_error("getrandom failed: no OS entropy for the RandomTape");
      }
      got += (size_t)n;
    }
    uint64_t w[8];
    memcpy(w, buf, 64);
    return sp::fq_from_u512(w);
  }
  Fq random_scalar(const char* label) { return tape_.challenge_scalar(label); }
  std::vector<Fq> random_vector(const char* label, size_t len) { return tape_.challenge_vector(label, len); }

 private:
  Transcript tape_;
};

}  // namespace spz
