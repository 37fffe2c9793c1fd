// spartan_amd host driver: C entry points over libspartan.hpp for ctypes (tests/, bench.py). Exceptions are
// turned into NULL / error strings here; nothing throws across the boundary.
#include <cstring>
#include <string>

#include "libspartan.hpp"
#include "fq_inv.hpp"

using namespace spz;

namespace {
thread_local std::string g_err;
struct EncH { ComputationCommitment comm; ComputationDecommitment decomm; };
struct ProofH { std::vector<uint8_t> bytes; };
FqVec limbs_vec(const uint64_t* p, size_t n) { FqVec v(n); if (n) memcpy(v[0].l, p, 32 * n); return v; }
void fill_times(const ProveTimes& t, double* o) {
  if (!o) return;
  o[0] = t.polycommit; o[1] = t.sc_phase_one; o[2] = t.sc_phase_two; o[3] = t.polyeval; o[4] = t.r1cs_sat; o[5] = t.eval_sparse_polys;
  o[6] = t.commit_nondet_witness; o[7] = t.build_layered_network; o[8] = t.evalproof_layered_network; o[9] = t.total;
}
template <typename F>
auto guard(F f) -> decltype(f()) {
  try {
    g_err.clear();
    return f();
  } catch (const std::exception& e) {
    g_err = e.what();
    return nullptr;
  }
}
}  // namespace

extern "C" {
const char* spz_last_error() { return g_err.c_str(); }
void* spz_ctx_new(int device) { return guard([&]() -> void* { return new Ctx(device); }); }
void spz_ctx_free(void* c) { delete (Ctx*)c; }
sp_ctx* spz_ctx_raw(void* c) { return ((Ctx*)c)->h; }
// row-sharded commitments across `world` lock-step ranks (libspartan.hpp: set_commit_shard); 0 = ok
int spz_ctx_set_commit_shard(void* c, int rank, int world, CommitGatherFn gather, void* user) {
  try {
    set_commit_shard(*(Ctx*)c, rank, world, gather, user);
    return 0;
  } catch (const std::exception& e) {
    g_err = e.what();
    return -1;
  }
}

// RCCL inside the library (shard.cc): rank 0 draws the id, the caller distributes it, every rank joins; 0 = ok
int spz_rccl_unique_id(uint8_t out[128]) {
  try { rccl_unique_id(out); return 0; } catch (const std::exception& e) { g_err = e.what(); return -1; }
}
int spz_ctx_set_commit_shard_rccl(void* c, int rank, int world, const uint8_t unique_id[128]) {
  try { set_commit_shard_rccl(*(Ctx*)c, rank, world, unique_id); return 0; } catch (const std::exception& e) { g_err = e.what(); return -1; }
}
// W shards on one GPU, in-process gather (the partitioning logic without a second GPU); nshards <= 1 clears it
int spz_ctx_set_commit_shard_virtual(void* c, int nshards) {
  try { set_commit_shard_virtual(*(Ctx*)c, nshards); return 0; } catch (const std::exception& e) { g_err = e.what(); return -1; }
}
double spz_rccl_allgather_probe(void* c, size_t bytes, int iters) {
  try { return rccl_allgather_probe(((Ctx*)c)->h, bytes, iters); } catch (const std::exception& e) { g_err = e.what(); return -2.0; }
}
void spz_ctx_shard_stats(void* c, int reset, uint64_t out[2]) {
  ShardStats s = commit_shard_stats(*(Ctx*)c, reset != 0);
  out[0] = s.gathers; out[1] = s.bytes;
}

// Instance::new (lib.rs:121-128): entries of A, B, C back to back as (row, col, [u8;32] canonical little-endian value).
// Errors mirror R1CSError: "InvalidIndex", "InvalidScalar" (spz_last_error()).
void* spz_instance_new(void* ctx, size_t num_cons, size_t num_vars, size_t num_inputs, const size_t nnz[3], const uint64_t* rows,
                       const uint64_t* cols, const uint8_t* vals) {
  return guard([&]() -> void* {
    std::vector<SparseEntry> m[3];
    // lib.rs:164-186: an entry is checked for InvalidIndex first, then its scalar for InvalidScalar
    size_t off = 0;
    for (int k = 0; k < 3; k++)
      for (size_t i = 0; i < nnz[k]; i++, off++) {
        SparseEntry e;
        e.row = rows[off]; e.col = cols[off];
        if (e.row >= num_cons || e.col >= num_vars + 1 + num_inputs) throw Error("InvalidIndex");
        sp::Fq raw;
        memcpy(raw.l, vals + 32 * off, 32);
        // Scalar::from_bytes (ristretto255.rs:390-416): reject encodings >= q
        static const uint64_t Q[4] = {SP_Q0, SP_Q1, SP_Q2, SP_Q3};
        bool lt = false;
        for (int w = 3; w >= 0; w--) {
          if (raw.l[w] != Q[w]) { lt = raw.l[w] < Q[w]; break; }
        }
        if (!lt) throw Error("InvalidScalar");
        e.val = sp::fq_to_mont(raw);
        m[k].push_back(e);
      }
    return new Instance(*(Ctx*)ctx, num_cons, num_vars, num_inputs, m[0], m[1], m[2]);
  });
}
// Instance::produce_synthetic_r1cs with a seed; vars (num_vars) and inputs (num_inputs) are written out
void* spz_instance_synthetic(void* ctx, size_t num_cons, size_t num_vars, size_t num_inputs, uint64_t seed, uint64_t* vars_out, uint64_t* inputs_out) {
  return guard([&]() -> void* {
    FqVec v, in;
    std::unique_ptr<Instance> p = Instance::produce_synthetic_r1cs(*(Ctx*)ctx, num_cons, num_vars, num_inputs, seed, &v, &in);
    if (vars_out) memcpy(vars_out, v[0].l, 32 * v.size());
    if (inputs_out && !in.empty()) memcpy(inputs_out, in[0].l, 32 * in.size());
    return p.release();
  });
}
// UniPoly::from_evals / compress / evaluate (unipoly.rs) on n = 3 or 4 evaluations; scalars as Montgomery limbs. Test hook.
// host arithmetic of the batched cubic sum-check (spark.inc): the cubic in the next challenge, and the last rounds on short tables
void spz_cubic_coeffs_probe(const uint64_t S[48], const uint64_t r[4], uint64_t ev[12]) {
  Fq s[12], rr, e[3];
  memcpy(s, S, sizeof s); memcpy(rr.l, r, 32);
  cubic_coeffs_probe(s, rr, e);
  memcpy(ev, e, sizeof e);
}
// tab: [ni][3][m] in, bound in place; evs: 3 per round (log2 m rounds); returns the number of rounds
int spz_cubic_tail_probe(uint64_t* tab, size_t ni, size_t m, const uint64_t* coeffs, const uint64_t* challenges, uint64_t* evs) {
  size_t rounds = 0;
  for (size_t x = m; x > 1; x /= 2) rounds++;
  FqVec t = limbs_vec(tab, ni * 3 * m), cf = limbs_vec(coeffs, ni), ch = limbs_vec(challenges, rounds), e;
  cubic_tail_probe(t, ni, m, cf, ch, &e);
  memcpy(tab, t.data(), 32 * t.size());
  memcpy(evs, e.data(), 32 * e.size());
  return (int)rounds;
}
int spz_eq_factor_probe(const uint64_t* rho, size_t nvars, size_t np, size_t ni, const uint64_t* coeffs, size_t nrounds, const uint64_t* claims,
                        const uint64_t* ev4, const uint64_t* challenges, uint64_t* evc_out, uint64_t* K_out) {
  auto vec = [](const uint64_t* p, size_t n) { FqVec v(n); memcpy(v.data(), p, 32 * n); return v; };
  FqVec evc, K;
  if (!eq_factor_probe(vec(rho, nvars), np, ni, vec(coeffs, ni), vec(claims, nrounds), vec(ev4, 4 * ni * nrounds), vec(challenges, nrounds), &evc, &K)) return 0;
  memcpy(evc_out, evc.data(), 32 * evc.size());
  memcpy(K_out, K.data(), 32 * K.size());
  return 1;
}
int spz_unipoly_probe(const uint64_t* evals, size_t n, const uint64_t r[4], uint64_t* coeffs, uint64_t* compressed, uint64_t eval_at_r[4]) {
  try {
    FqVec e(n), c, cc;
    memcpy(e.data(), evals, 32 * n);
    Fq rr, ev;
    memcpy(rr.l, r, 32);
    unipoly_probe(e, rr, &c, &cc, &ev);
    memcpy(coeffs, c.data(), 32 * c.size());
    memcpy(compressed, cc.data(), 32 * cc.size());
    memcpy(eval_at_r, ev.l, 32);
    return 0;
  } catch (const std::exception& ex) {
    g_err = ex.what();
    return -1;
  }
}
// test hook: the challenge inversion of the inner-product rounds (fq_inv.hpp), Montgomery limbs in and out
void spz_fq_invert_vartime(const uint64_t in[4], uint64_t out[4]) { Fq a; memcpy(a.l, in, 32); Fq r = fq_invert_vartime(a); memcpy(out, r.l, 32); }
void spz_seed_scalar(const char* domain, uint64_t seed, uint64_t out[4]) { Fq s = seed_scalar(domain, seed); memcpy(out, s.l, 32); }
void spz_instance_set_digest(void* inst, const uint8_t* d, size_t n) { ((Instance*)inst)->set_digest(d, n); }
// zlib header variant of the COMPUTED digest: 0 = 0x78 0x9C (miniz >= 2.2, miniz_oxide >= 0.4: FLEVEL from the level), 1 = 0x78 0x01 (older).
// An explicit parameter of the instance (not an environment switch); call before the digest is first used.
int spz_instance_set_digest_header(void* inst, int old_header) {  // 0 = ok, -1 = a digest with the other header has already been computed (or set)
  if (((Instance*)inst)->set_digest_header(old_header != 0)) return 0;
  g_err = "set_digest_header after the digest was first used";
  return -1;
}
// R1CSShape::get_digest (r1cs.rs:154-158): the zlib stream; and the bincode it compresses (for the round-trip tests)
size_t spz_instance_digest(void* inst, uint8_t* out, size_t cap) {
  const std::vector<uint8_t> d = ((Instance*)inst)->compute_digest();
  if (out && cap >= d.size()) memcpy(out, d.data(), d.size());
  return d.size();
}
size_t spz_instance_shape_bincode(void* inst, uint8_t* out, size_t cap) {
  std::vector<uint8_t> b = ((Instance*)inst)->shape_bincode();
  if (out && cap >= b.size()) memcpy(out, b.data(), b.size());
  return b.size();
}
// the deflater alone (CPU tests): zlib_level6_miniz of arbitrary bytes
size_t spz_zlib_level6(const uint8_t* data, size_t n, int old_header, uint8_t* out, size_t cap) {
  std::vector<uint8_t> z = zlib_level6_miniz(data, n, old_header != 0);
  if (out && cap >= z.size()) memcpy(out, z.data(), z.size());
  return z.size();
}
// the same deflater at another tdefl probe count (miniz levels 4..10 = 16/32/128/256/512/768/1500): pinned per level against the real miniz
size_t spz_zlib_probes(const uint8_t* data, size_t n, unsigned probes, int old_header, uint8_t* out, size_t cap) {
  std::vector<uint8_t> z = zlib_miniz_probes(data, n, probes, old_header != 0);
  if (out && cap >= z.size()) memcpy(out, z.data(), z.size());
  return z.size();
}
void spz_instance_free(void* i) { delete (Instance*)i; }
void* spz_snark_gens_new(void* ctx, size_t nc, size_t nv, size_t ni, size_t nnz) {
  return guard([&]() -> void* { return new SNARKGens(*(Ctx*)ctx, nc, nv, ni, nnz); });
}
void spz_snark_gens_free(void* g) { delete (SNARKGens*)g; }
void* spz_nizk_gens_new(void* ctx, size_t nc, size_t nv, size_t ni) {
  return guard([&]() -> void* { return new NIZKGens(*(Ctx*)ctx, nc, nv, ni); });
}
void spz_nizk_gens_free(void* g) { delete (NIZKGens*)g; }
// compressed points of the two generator streams (for parity checks against the oracle's MultiCommitGens)
size_t spz_snark_gens_stream(void* g, int which, uint8_t* out, size_t cap) {
  const std::vector<uint8_t>& v = which == 0 ? ((SNARKGens*)g)->stream_sat.compressed : ((SNARKGens*)g)->stream_eval.compressed;
  if (out && cap >= v.size()) memcpy(out, v.data(), v.size());
  return v.size();
}
// library options of the context (include/spartan_hip.h: sp_ctx_set_option); ctx == NULL: the process-wide defaults
int spz_ctx_set_option(void* ctx, const char* key, const char* value) { return sp_ctx_set_option(ctx ? ((Ctx*)ctx)->h : nullptr, key, value); }
// window width of the fixed-base tables of a generator stream (0: gens_r1cs_sat, 1: gens_r1cs_eval)
int spz_snark_gens_window_bits(void* g, int which) { return sp_gens_window_bits(which == 0 ? ((SNARKGens*)g)->stream_sat.g : ((SNARKGens*)g)->stream_eval.g); }
int spz_snark_gens_windows(void* g, int which) { return sp_gens_windows(which == 0 ? ((SNARKGens*)g)->stream_sat.g : ((SNARKGens*)g)->stream_eval.g); }
size_t spz_snark_gens_table_bytes(void* g, int which) { return sp_gens_table_bytes(which == 0 ? ((SNARKGens*)g)->stream_sat.g : ((SNARKGens*)g)->stream_eval.g); }
// bincode of SNARKGens / ComputationCommitment (wire formats)
size_t spz_snark_gens_bincode(void* g, uint8_t* out, size_t cap) {
  std::vector<uint8_t> b = ((SNARKGens*)g)->serialize();
  if (out && cap >= b.size()) memcpy(out, b.data(), b.size());
  return b.size();
}
size_t spz_commitment_bincode(void* e, uint8_t* out, size_t cap);
void* spz_snark_encode(void* ctx, void* inst, void* gens) {
  return guard([&]() -> void* {
    EncH* e = new EncH;
    try {
      SNARK::encode(*(Ctx*)ctx, *(Instance*)inst, *(SNARKGens*)gens, &e->comm, &e->decomm);
    } catch (...) {
      delete e;
      throw;
    }
    return e;
  });
}
void spz_encode_free(void* e) { delete (EncH*)e; }
size_t spz_commitment_bincode(void* e, uint8_t* out, size_t cap) {
  std::vector<uint8_t> b = ((EncH*)e)->comm.serialize();
  if (out && cap >= b.size()) memcpy(out, b.data(), b.size());
  return b.size();
}
// bincode(ComputationDecommitment): returns the size; fills out when cap suffices
size_t spz_decommitment_bincode(void* e, uint8_t* out, size_t cap) {
  try {
    std::vector<uint8_t> b = ((EncH*)e)->decomm.serialize();
    if (out && cap >= b.size()) memcpy(out, b.data(), b.size());
    return b.size();
  } catch (const std::exception& ex) {
    g_err = ex.what();
    return 0;
  }
}
size_t spz_encode_comm(void* ev, int which, uint8_t* out, size_t cap) {
  EncH* e = (EncH*)ev;
  const PolyCommitment& c = which == 0 ? e->comm.comm.comm_comb_ops : e->comm.comm.comm_comb_mem;
  if (out && cap >= 32 * c.C.size())
    for (size_t i = 0; i < c.C.size(); i++) memcpy(out + 32 * i, c.C[i].data(), 32);
  return c.C.size();
}
void* spz_snark_prove(void* ctx, void* inst, void* gens, void* enc, const uint64_t* vars, size_t nvars, const uint64_t* inputs, size_t ninputs,
                      const char* transcript_label, const uint64_t tape_seed[4], double* times10) {
  return guard([&]() -> void* {
    Transcript t(transcript_label);
    Fq seed;  // tape_seed == NULL: RandomTape::new, seeded from OS entropy (random.rs:13-15)
    if (tape_seed) memcpy(seed.l, tape_seed, 32);
    ProveTimes tm;
    EncH* e = (EncH*)enc;
    SNARK p = SNARK::prove(*(Ctx*)ctx, *(Instance*)inst, e->comm, e->decomm, (const sp::Fq*)vars, nvars, limbs_vec(inputs, ninputs), *(SNARKGens*)gens,
                           t, tape_seed ? &seed : nullptr, &tm);
    fill_times(tm, times10);
    ProofH* h = new ProofH;
    h->bytes = p.serialize();
    return h;
  });
}
// SNARK::prove / NIZK::prove on a CALLER-OWNED transcript (src/lib.rs:339-347, 501-509: `transcript: &mut Transcript`): the
// 203 bytes of the merlin transcript (STROBE state, pos, pos_begin, cur_flags) come in, the proof is produced on their
// continuation, and the state after the proof goes back so the caller's transcript can carry on — whatever it had absorbed
// before. `assignment` (a spz_vars_assignment_new handle) or `vars` (host scalars), one of them. This is the entry the Rust
// crate's SNARK::prove / NIZK::prove call under `--features gpu` (rust_shim/seams/lib.rs).
void* spz_snark_prove_t(void* ctx, void* inst, void* gens, void* enc, void* assignment, const uint64_t* vars, size_t nvars, const uint64_t* inputs,
                        size_t ninputs, uint8_t transcript_state[203], const uint64_t tape_seed[4], double* times10) {
  return guard([&]() -> void* {
    if (!transcript_state || (!assignment && !vars)) throw Error("spz_snark_prove_t: bad arguments");
    Transcript t(Transcript::FromState(), transcript_state);
    Fq seed;
    if (tape_seed) memcpy(seed.l, tape_seed, 32);
    ProveTimes tm;
    EncH* e = (EncH*)enc;
    SNARK p = assignment ? SNARK::prove(*(Ctx*)ctx, *(Instance*)inst, e->comm, e->decomm, *(VarsAssignment*)assignment, limbs_vec(inputs, ninputs),
                                        *(SNARKGens*)gens, t, tape_seed ? &seed : nullptr, &tm)
                         : SNARK::prove(*(Ctx*)ctx, *(Instance*)inst, e->comm, e->decomm, (const sp::Fq*)vars, nvars, limbs_vec(inputs, ninputs),
                                        *(SNARKGens*)gens, t, tape_seed ? &seed : nullptr, &tm);
    fill_times(tm, times10);
    ProofH* h = new ProofH;
    h->bytes = p.serialize();
    t.export_state(transcript_state);
    return h;
  });
}
void* spz_nizk_prove_t(void* ctx, void* inst, void* gens, void* assignment, const uint64_t* vars, size_t nvars, const uint64_t* inputs, size_t ninputs,
                       uint8_t transcript_state[203], const uint64_t tape_seed[4], double* times10) {
  return guard([&]() -> void* {
    if (!transcript_state || (!assignment && !vars)) throw Error("spz_nizk_prove_t: bad arguments");
    Transcript t(Transcript::FromState(), transcript_state);
    Fq seed;
    if (tape_seed) memcpy(seed.l, tape_seed, 32);
    ProveTimes tm;
    NIZK p = assignment ? NIZK::prove(*(Ctx*)ctx, *(Instance*)inst, *(VarsAssignment*)assignment, limbs_vec(inputs, ninputs), *(NIZKGens*)gens, t,
                                      tape_seed ? &seed : nullptr, &tm)
                        : NIZK::prove(*(Ctx*)ctx, *(Instance*)inst, (const sp::Fq*)vars, nvars, limbs_vec(inputs, ninputs), *(NIZKGens*)gens, t,
                                      tape_seed ? &seed : nullptr, &tm);
    fill_times(tm, times10);
    ProofH* h = new ProofH;
    h->bytes = p.serialize();
    t.export_state(transcript_state);
    return h;
  });
}
// VarsAssignment::new: the assignment uploaded once; spz_*_prove_resident prove from the device copy
void* spz_vars_assignment_new(void* ctx, const uint64_t* vars, size_t nvars) {
  return guard([&]() -> void* { return new VarsAssignment(*(Ctx*)ctx, (const sp::Fq*)vars, nvars); });
}
void spz_vars_assignment_free(void* a) { delete (VarsAssignment*)a; }
void* spz_snark_prove_resident(void* ctx, void* inst, void* gens, void* enc, void* assignment, const uint64_t* inputs, size_t ninputs,
                               const char* transcript_label, const uint64_t tape_seed[4], double* times10) {
  return guard([&]() -> void* {
    Transcript t(transcript_label);
    Fq seed;
    if (tape_seed) memcpy(seed.l, tape_seed, 32);
    ProveTimes tm;
    EncH* e = (EncH*)enc;
    SNARK p = SNARK::prove(*(Ctx*)ctx, *(Instance*)inst, e->comm, e->decomm, *(VarsAssignment*)assignment, limbs_vec(inputs, ninputs), *(SNARKGens*)gens,
                           t, tape_seed ? &seed : nullptr, &tm);
    fill_times(tm, times10);
    ProofH* h = new ProofH;
    h->bytes = p.serialize();
    return h;
  });
}
void* spz_nizk_prove_resident(void* ctx, void* inst, void* gens, void* assignment, const uint64_t* inputs, size_t ninputs, const char* transcript_label,
                              const uint64_t tape_seed[4], double* times10) {
  return guard([&]() -> void* {
    Transcript t(transcript_label);
    Fq seed;
    if (tape_seed) memcpy(seed.l, tape_seed, 32);
    ProveTimes tm;
    NIZK p = NIZK::prove(*(Ctx*)ctx, *(Instance*)inst, *(VarsAssignment*)assignment, limbs_vec(inputs, ninputs), *(NIZKGens*)gens, t,
                         tape_seed ? &seed : nullptr, &tm);
    fill_times(tm, times10);
    ProofH* h = new ProofH;
    h->bytes = p.serialize();
    return h;
  });
}
void* spz_nizk_prove(void* ctx, void* inst, void* gens, const uint64_t* vars, size_t nvars, const uint64_t* inputs, size_t ninputs,
                     const char* transcript_label, const uint64_t tape_seed[4], double* times10) {
  return guard([&]() -> void* {
    Transcript t(transcript_label);
    Fq seed;
    if (tape_seed) memcpy(seed.l, tape_seed, 32);
    ProveTimes tm;
    NIZK p = NIZK::prove(*(Ctx*)ctx, *(Instance*)inst, (const sp::Fq*)vars, nvars, limbs_vec(inputs, ninputs), *(NIZKGens*)gens, t, tape_seed ? &seed : nullptr, &tm);
    fill_times(tm, times10);
    ProofH* h = new ProofH;
    h->bytes = p.serialize();
    return h;
  });
}
size_t spz_proof_bytes(void* p, uint8_t* out, size_t cap) {
  ProofH* h = (ProofH*)p;
  if (out && cap >= h->bytes.size()) memcpy(out, h->bytes.data(), h->bytes.size());
  return h->bytes.size();
}
void spz_proof_free(void* p) { delete (ProofH*)p; }

// host-side Fiat–Shamir pieces exposed for CPU tests (no GPU needed)
const char* spz_keccak_variant() { return keccak_f1600_variant(); }
int spz_keccak_run_variant(const char* name, uint64_t state[25]) { return keccak_f1600_run_variant(name, state); }
void spz_shake256(const uint8_t* in, size_t n, uint8_t* out, size_t outlen) { Shake256 s; s.absorb(in, n); s.squeeze(out, outlen); }
size_t spz_merlin_script(const char* tlabel, size_t nops, const int* kinds, const char* const* labels, const uint8_t* const* datas, const size_t* lens,
                         uint8_t* out) {
  Transcript t(tlabel);
  size_t o = 0;
  for (size_t i = 0; i < nops; i++) {
    if (kinds[i] == 0) t.append_message(labels[i], datas[i], lens[i]);
    else if (kinds[i] == 1) { t.challenge_bytes(labels[i], out + o, lens[i]); o += lens[i]; }
    else { uint64_t x; memcpy(&x, datas[i], 8); t.append_u64(labels[i], x); }
  }
  return o;
}
// the 203-byte state of a merlin transcript after `Transcript::new(tlabel)` and a script of messages (kinds as above; no challenges)
void spz_merlin_state(const char* tlabel, size_t nops, const int* kinds, const char* const* labels, const uint8_t* const* datas, const size_t* lens,
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
// continue a transcript from a state: one challenge of n bytes (checks import/export round trips in the CPU tests)
void spz_merlin_challenge_from_state(uint8_t state[203], const char* label, uint8_t* out, size_t n) {
  Transcript t(Transcript::FromState(), state);
  t.challenge_bytes(label, out, n);
  t.export_state(state);
}
// The probe of the coarse Rust binding (rust_shim/src/gpu_tail.rs.in, transcript_state / set_transcript_state): that binding reads and writes
// merlin::Transcript's private state through the struct's memory, which `repr(Rust)` does not pin. Before its first use the Rust side runs
// this fixed script on its own merlin transcript — Transcript::new(b"spartan_amd binding probe"), append_message(b"probe-message", 0..63) —
// exports it through the raw copy and compares the 203 bytes with out_state; then it imports out_state into a fresh merlin transcript
// through the raw write and compares challenge_bytes(b"probe-challenge", 32) with out_challenge. A field order or padding that differs
// from {state[200], pos, pos_begin, cur_flags} fails one of the two, in release builds too.
void spz_transcript_probe(uint8_t out_state[203], uint8_t out_challenge[32]) {
  Transcript t("spartan_amd binding probe");
  uint8_t msg[64];
  for (int i = 0; i < 64; i++) msg[i] = (uint8_t)i;
  t.append_message("probe-message", msg, 64);
  t.export_state(out_state);
  t.challenge_bytes("probe-challenge", out_challenge, 32);
}
void spz_tape_draws(const uint64_t seed[4], const char* label, size_t n, uint64_t* out) {
  Fq s;
  memcpy(s.l, seed, 32);
  RandomTape tape("proof", s);
  for (size_t i = 0; i < n; i++) { Fq x = tape.random_scalar(label); memcpy(out + 4 * i, x.l, 32); }
}
}
