// ORACLE (test infrastructure only). Keccak-f[1600], SHAKE256, STROBE-128 as used by Merlin. See transcript.h.
#include "transcript.h"

namespace orc {

static const uint64_t KECCAK_RC[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
    0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
    0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
    0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
    0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
static const int KECCAK_ROT[24] = {1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14, 27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44};
static const int KECCAK_PIL[24] = {10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1};

static inline uint64_t rotl64(uint64_t x, int n) { return (x << n) | (x >> (64 - n)); }

void keccak_f1600(uint64_t st[25]) {
  uint64_t bc[5], t;
  for (int round = 0; round < 24; round++) {
    for (int i = 0; i < 5; i++) bc[i] = st[i] ^ st[i + 5] ^ st[i + 10] ^ st[i + 15] ^ st[i + 20];
    for (int i = 0; i < 5; i++) {
      t = bc[(i + 4) % 5] ^ rotl64(bc[(i + 1) % 5], 1);
      for (int j = 0; j < 25; j += 5) st[j + i] ^= t;
    }
    t = st[1];
    for (int i = 0; i < 24; i++) {
      int j = KECCAK_PIL[i];
      uint64_t b = st[j];
      st[j] = rotl64(t, KECCAK_ROT[i]);
      t = b;
    }
    for (int j = 0; j < 25; j += 5) {
      for (int i = 0; i < 5; i++) bc[i] = st[j + i];
      for (int i = 0; i < 5; i++) st[j + i] ^= (~bc[(i + 1) % 5]) & bc[(i + 2) % 5];
    }
    st[0] ^= KECCAK_RC[round];
  }
}

static inline void permute_bytes(uint8_t st[200]) {
  uint64_t w[25];
  memcpy(w, st, 200);
  keccak_f1600(w);
  memcpy(st, w, 200);
}

void Shake256::absorb(const uint8_t* d, size_t n) {
  for (size_t i = 0; i < n; i++) {
    st[pos++] ^= d[i];
    if (pos == 136) { permute_bytes(st); pos = 0; }
  }
}
void Shake256::squeeze(uint8_t* out, size_t n) {
  if (!squeezing) {
    st[pos] ^= 0x1f;
    st[135] ^= 0x80;
    permute_bytes(st);
    pos = 0;
    squeezing = true;
  }
  for (size_t i = 0; i < n; i++) {
    if (pos == 136) { permute_bytes(st); pos = 0; }
    out[i] = st[pos++];
  }
}

// ---- STROBE-128 subset used by Merlin (merlin 3.0.0 src/strobe.rs, restated from the STROBE v1.0.2 spec) ----
static const uint8_t STROBE_R = 166;
enum { FLAG_I = 1, FLAG_A = 2, FLAG_C = 4, FLAG_T = 8, FLAG_M = 16, FLAG_K = 32 };

Strobe128::Strobe128(const char* proto) : pos(0), pos_begin(0), cur_flags(0) {
  memset(st, 0, 200);
  const uint8_t hdr[6] = {1, (uint8_t)(STROBE_R + 2), 1, 0, 1, 96};
  memcpy(st, hdr, 6);
  memcpy(st + 6, "STROBEv1.0.2", 12);
  permute_bytes(st);
  meta_ad((const uint8_t*)proto, strlen(proto), false);
}
void Strobe128::run_f() {
  st[pos] ^= pos_begin;
  st[pos + 1] ^= 0x04;
  st[STROBE_R + 1] ^= 0x80;
  permute_bytes(st);
  pos = 0;
  pos_begin = 0;
}
void Strobe128::absorb(const uint8_t* d, size_t n) {
  for (size_t i = 0; i < n; i++) {
    st[pos] ^= d[i];
    pos++;
    if (pos == STROBE_R) run_f();
  }
}
void Strobe128::squeeze(uint8_t* out, size_t n) {
  for (size_t i = 0; i < n; i++) {
    out[i] = st[pos];
    st[pos] = 0;
    pos++;
    if (pos == STROBE_R) run_f();
  }
}
void Strobe128::begin_op(uint8_t flags, bool more) {
  if (more) return;  // continuation of the same operation (flags must match cur_flags)
  uint8_t old_begin = pos_begin;
  pos_begin = pos + 1;
  cur_flags = flags;
  uint8_t hdr[2] = {old_begin, flags};
  absorb(hdr, 2);
  bool force_f = (flags & (FLAG_C | FLAG_K)) != 0;
  if (force_f && pos != 0) run_f();
}
void Strobe128::meta_ad(const uint8_t* d, size_t n, bool more) {
  begin_op(FLAG_M | FLAG_A, more);
  absorb(d, n);
}
void Strobe128::ad(const uint8_t* d, size_t n, bool more) {
  begin_op(FLAG_A, more);
  absorb(d, n);
}
void Strobe128::prf(uint8_t* out, size_t n, bool more) {
  begin_op(FLAG_I | FLAG_A | FLAG_C, more);
  squeeze(out, n);
}

}  // namespace orc
