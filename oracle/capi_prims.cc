// ORACLE (test infrastructure only). C entry points over the primitive layers, used by tests/ via ctypes.
#include <cstdlib>

#include "fq.h"
#include "ristretto.h"
#include "transcript.h"

using namespace orc;

extern "C" {

// ---- F_q (limbs are the reference's in-memory Scalar: 4 x u64 Montgomery, ristretto255.rs:199) ----
void orc_fq_mul(const uint64_t a[4], const uint64_t b[4], uint64_t out[4]) { Fq x, y; memcpy(x.l, a, 32); memcpy(y.l, b, 32); Fq r = fq_mul(x, y); memcpy(out, r.l, 32); }
void orc_fq_add(const uint64_t a[4], const uint64_t b[4], uint64_t out[4]) { Fq x, y; memcpy(x.l, a, 32); memcpy(y.l, b, 32); Fq r = fq_add(x, y); memcpy(out, r.l, 32); }
void orc_fq_sub(const uint64_t a[4], const uint64_t b[4], uint64_t out[4]) { Fq x, y; memcpy(x.l, a, 32); memcpy(y.l, b, 32); Fq r = fq_sub(x, y); memcpy(out, r.l, 32); }
void orc_fq_neg(const uint64_t a[4], uint64_t out[4]) { Fq x; memcpy(x.l, a, 32); Fq r = fq_neg(x); memcpy(out, r.l, 32); }
void orc_fq_invert(const uint64_t a[4], uint64_t out[4]) { Fq x; memcpy(x.l, a, 32); Fq r = fq_invert(x); memcpy(out, r.l, 32); }
void orc_fq_from_bytes_wide(const uint8_t b[64], uint64_t out[4]) { Fq r = fq_from_bytes_wide(b); memcpy(out, r.l, 32); }
void orc_fq_from_u64(uint64_t v, uint64_t out[4]) { Fq r = fq_from_u64(v); memcpy(out, r.l, 32); }
void orc_fq_to_bytes(const uint64_t a[4], uint8_t out[32]) { Fq x; memcpy(x.l, a, 32); fq_to_bytes(x, out); }
int orc_fq_from_bytes(const uint8_t b[32], uint64_t out[4]) { Fq r; bool ok = fq_from_bytes(b, &r); memcpy(out, r.l, 32); return ok ? 1 : 0; }

// ---- group ----
void orc_pt_from_uniform_bytes(const uint8_t b[64], uint8_t out[32]) { Pt p = pt_from_uniform_bytes(b); pt_compress(p, out); }
int orc_pt_recompress(const uint8_t in[32], uint8_t out[32]) { Pt p; if (!pt_decompress(in, &p)) return 0; pt_compress(p, out); return 1; }
int orc_pt_add(const uint8_t a[32], const uint8_t b[32], uint8_t out[32]) { Pt p, q; if (!pt_decompress(a, &p) || !pt_decompress(b, &q)) return 0; pt_compress(pt_add(p, q), out); return 1; }
int orc_pt_dbl(const uint8_t a[32], uint8_t out[32]) { Pt p; if (!pt_decompress(a, &p)) return 0; pt_compress(pt_dbl(p), out); return 1; }
// scalar given as canonical 32 little-endian bytes (< q)
int orc_pt_mul_bytes(const uint8_t s[32], const uint8_t a[32], uint8_t out[32]) {
  Pt p; Fq x; if (!pt_decompress(a, &p)) return 0; if (!fq_from_bytes(s, &x)) return 0;
  pt_compress(pt_mul(x, p), out); return 1;
}
// scalars: n x 4 u64 Montgomery limbs; points: n x 32 compressed
int orc_pt_msm(const uint64_t* scalars, const uint8_t* points, size_t n, uint8_t out[32]) {
  std::vector<Fq> s(n); std::vector<Pt> p(n);
  for (size_t i = 0; i < n; i++) { memcpy(s[i].l, scalars + 4 * i, 32); if (!pt_decompress(points + 32 * i, &p[i])) return 0; }
  pt_compress(pt_msm(s.data(), p.data(), n), out); return 1;
}
void orc_basepoint(uint8_t out[32]) { pt_compress(pt_basepoint(), out); }

// ---- hashing / transcript ----
void orc_shake256(const uint8_t* in, size_t n, uint8_t* out, size_t outlen) { Shake256 s; s.absorb(in, n); s.squeeze(out, outlen); }
// Transcript::new(tlabel); append_message(mlabel, msg); challenge_bytes(clabel, out)
void orc_merlin_simple(const char* tlabel, const char* mlabel, const uint8_t* msg, size_t n, const char* clabel, uint8_t* out, size_t outlen) {
  Transcript t(tlabel); t.append_message(mlabel, msg, n); t.challenge_bytes(clabel, out, outlen);
}
// scripted transcript for cross-checks: ops is a sequence of (kind, label, data) ; kind 0 = append_message,
// 1 = challenge_bytes(len = datalen, output appended to out), 2 = append_u64 (data = 8 bytes LE)
size_t orc_merlin_script(const char* tlabel, size_t nops, const int* kinds, const char* const* labels, const uint8_t* const* datas, const size_t* lens, uint8_t* out) {
  Transcript t(tlabel); size_t o = 0;
  for (size_t i = 0; i < nops; i++) {
    if (kinds[i] == 0) t.append_message(labels[i], datas[i], lens[i]);
    else if (kinds[i] == 1) { t.challenge_bytes(labels[i], out + o, lens[i]); o += lens[i]; }
    else { uint64_t x; memcpy(&x, datas[i], 8); t.append_u64(labels[i], x); }
  }
  return o;
}
// the 203-byte transcript state after the same kind of script (challenges are drawn and dropped)
void orc_merlin_state(const char* tlabel, size_t nops, const int* kinds, const char* const* labels, const uint8_t* const* datas, const size_t* lens,
                      uint8_t out_state[203]) {
  Transcript t(tlabel);
  uint8_t sink[256];
  for (size_t i = 0; i < nops; i++) {
    if (kinds[i] == 0) t.append_message(labels[i], datas[i], lens[i]);
    else if (kinds[i] == 1) t.challenge_bytes(labels[i], sink, lens[i] < sizeof sink ? lens[i] : sizeof sink);
    else { uint64_t x; memcpy(&x, datas[i], 8); t.append_u64(labels[i], x); }
  }
  t.export_state(out_state);
}
void orc_merlin_challenge_from_state(uint8_t state[203], const char* label, uint8_t* out, size_t n) {
  Transcript t("x");
  t.import_state(state);
  t.challenge_bytes(label, out, n);
  t.export_state(state);
}
}
