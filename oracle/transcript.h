// ORACLE (test infrastructure only — never linked into the product path).
// Fiat–Shamir layer of the reference: src/transcript.rs:13-63 (ProofTranscript / AppendToTranscript) and
// src/random.rs:10-28 (RandomTape), on top of the third-party crates merlin 3.0.0 (STROBE-128 over
// Keccak-f[1600]) and sha3 0.8.2 (SHAKE256, commitments.rs:16-24) — neither is under /root/reference.
// Restated from the published constructions (Merlin v1.0 spec, STROBE v1.0.2, FIPS 202) and pinned against
// the Merlin "equivalence_simple" test vector and hashlib.shake_256 (tests/test_oracle_transcript.py).
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "fq.h"

namespace orc {

void keccak_f1600(uint64_t st[25]);

struct Shake256 {  // FIPS 202 XOF, rate 136, domain suffix 0x1f
  uint8_t st[200];
  size_t pos;
  bool squeezing;
  Shake256() : pos(0), squeezing(false) { memset(st, 0, 200); }
  void absorb(const uint8_t* d, size_t n);
  void squeeze(uint8_t* out, size_t n);
};

struct Strobe128 {
  uint8_t st[200];
  uint8_t pos, pos_begin, cur_flags;
  explicit Strobe128(const char* proto);
  void meta_ad(const uint8_t* d, size_t n, bool more);
  void ad(const uint8_t* d, size_t n, bool more);
  void prf(uint8_t* out, size_t n, bool more);
  void run_f();
  void absorb(const uint8_t* d, size_t n);
  void squeeze(uint8_t* out, size_t n);
  void begin_op(uint8_t flags, bool more);
};

struct Transcript {  // merlin::Transcript
  Strobe128 s;
  explicit Transcript(const char* label) : s("Merlin v1.0") { append_message("dom-sep", (const uint8_t*)label, strlen(label)); }
  // merlin 3.0.0: Transcript { strobe: Strobe128 { state: [u8; 200], pos, pos_begin, cur_flags } } — 203 bytes say everything.
  // The reference's provers take `transcript: &mut Transcript` (src/lib.rs:339-347): a caller may have absorbed anything
  // before; these two let a test start the oracle from such a state and read the state the proof leaves behind.
  void import_state(const uint8_t in[203]) { memcpy(s.st, in, 200); s.pos = in[200]; s.pos_begin = in[201]; s.cur_flags = in[202]; }
  void export_state(uint8_t out[203]) const { memcpy(out, s.st, 200); out[200] = s.pos; out[201] = s.pos_begin; out[202] = s.cur_flags; }
  void append_message(const char* label, const uint8_t* msg, size_t n) {
    uint8_t len[4] = {(uint8_t)n, (uint8_t)(n >> 8), (uint8_t)(n >> 16), (uint8_t)(n >> 24)};
    s.meta_ad((const uint8_t*)label, strlen(label), false);
    s.meta_ad(len, 4, true);
    s.ad(msg, n, false);
  }
  void append_message(const char* label, const char* msg) { append_message(label, (const uint8_t*)msg, strlen(msg)); }
  void append_u64(const char* label, uint64_t x) {
    uint8_t b[8];
    memcpy(b, &x, 8);
    append_message(label, b, 8);
  }
  void challenge_bytes(const char* label, uint8_t* out, size_t n) {
    uint8_t len[4] = {(uint8_t)n, (uint8_t)(n >> 8), (uint8_t)(n >> 16), (uint8_t)(n >> 24)};
    s.meta_ad((const uint8_t*)label, strlen(label), false);
    s.meta_ad(len, 4, true);
    s.prf(out, n, false);
  }
  // --- transcript.rs:13-37 ProofTranscript ---
  void append_protocol_name(const char* name) { append_message("protocol-name", name); }
  void append_scalar(const char* label, const Fq& x) {
    uint8_t b[32];
    fq_to_bytes(x, b);
    append_message(label, b, 32);
  }
  void append_point(const char* label, const uint8_t pt[32]) { append_message(label, pt, 32); }
  Fq challenge_scalar(const char* label) {
    uint8_t buf[64];
    challenge_bytes(label, buf, 64);
    return fq_from_bytes_wide(buf);
  }
  std::vector<Fq> challenge_vector(const char* label, size_t len) {
    std::vector<Fq> v(len);
    for (size_t i = 0; i < len; i++) v[i] = challenge_scalar(label);
    return v;
  }
  // transcript.rs:49-57  AppendToTranscript for [Scalar]
  void append_scalars(const char* label, const Fq* v, size_t n) {
    append_message(label, "begin_append_vector");
    for (size_t i = 0; i < n; i++) append_scalar(label, v[i]);
    append_message(label, "end_append_vector");
  }
  void append_scalars(const char* label, const std::vector<Fq>& v) { append_scalars(label, v.data(), v.size()); }
};

// random.rs:10-28. The reference seeds the tape from OsRng (random.rs:13-15); the oracle takes the seed
// scalar explicitly — this is the determinism hook of SURVEY.md fact 1 / §7.3-1.
struct RandomTape {
  Transcript tape;
  RandomTape(const char* name, const Fq& seed) : tape(name) { tape.append_scalar("init_randomness", seed); }
  Fq random_scalar(const char* label) { return tape.challenge_scalar(label); }
  std::vector<Fq> random_vector(const char* label, size_t len) { return tape.challenge_vector(label, len); }
};

}  // namespace orc
