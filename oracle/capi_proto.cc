// ORACLE (test infrastructure only). C entry points over the protocol restatement, used by tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg via ctypes. Never linked into the product library.
#include <omp.h>

#include "spartan.h"

using namespace orc;

namespace {
struct InstH { R1CSShape inst; FqVec vars, inputs; };
struct SnarkGensH { SNARKGens g; };
struct NizkGensH { NIZKGens g; };
struct EncH { R1CSCommitment comm; R1CSDecommitment decomm; };
struct ProofH { std::vector<uint8_t> bytes; SNARKProof snark; NIZKProof nizk; bool is_snark; };
static Fq limbs(const uint64_t* p) { Fq x; memcpy(x.l, p, 32); return x; }
static FqVec limbs_vec(const uint64_t* p, size_t n) { FqVec v(n); for (size_t i = 0; i < n; i++) v[i] = limbs(p + 4 * i); return v; }
static void out_vec(const FqVec& v, uint64_t* o) { for (size_t i = 0; i < v.size(); i++) memcpy(o + 4 * i, v[i].l, 32); }
static void fill_times(const ProveTimes& t, double* o) {
  if (!o) return;
  o[0] = t.polycommit; o[1] = t.sc_phase_one; o[2] = t.sc_phase_two; o[3] = t.polyeval; o[4] = t.r1cs_sat; o[5] = t.eval_sparse_polys;
  o[6] = t.commit_nondet_witness; o[7] = t.build_layered_network; o[8] = t.evalproof_layered_network; o[9] = t.total;
}
}  // namespace

extern "C" {
void orc_set_threads(int n) { omp_set_num_threads(n); }

// ---- instances ----
void* orc_instance_synthetic(size_t num_cons, size_t num_vars, size_t num_inputs, uint64_t seed) {
  InstH* h = new InstH;
  produce_synthetic_r1cs(num_cons, num_vars, num_inputs, seed, &h->inst, &h->vars, &h->inputs);
  return h;
}
// entries: rows/cols as u64 arrays, vals as 4-limb Montgomery; matrices given back to back (A, B, C)
void* orc_instance_new(size_t num_cons, size_t num_vars, size_t num_inputs, const size_t nnz[3], const uint64_t* rows, const uint64_t* cols,
                       const uint64_t* vals, const uint64_t* vars, size_t n_assigned_vars, const uint64_t* inputs) {
  InstH* h = new InstH;
  h->inst.num_cons = num_cons; h->inst.num_vars = num_vars; h->inst.num_inputs = num_inputs;
  SparseMatPoly* m[3] = {&h->inst.A, &h->inst.B, &h->inst.C};
  size_t off = 0;
  for (int k = 0; k < 3; k++) {
    m[k]->num_vars_x = log_2(num_cons); m[k]->num_vars_y = log_2(2 * num_vars);
    for (size_t i = 0; i < nnz[k]; i++, off++) m[k]->M.push_back({(size_t)rows[off], (size_t)cols[off], limbs(vals + 4 * off)});
  }
  h->vars = limbs_vec(vars, n_assigned_vars);
  h->inputs = limbs_vec(inputs, num_inputs);
  return h;
}
void orc_instance_free(void* h) { delete (InstH*)h; }
size_t orc_instance_nnz(void* hv, int which) { InstH* h = (InstH*)hv; return (which == 0 ? h->inst.A : which == 1 ? h->inst.B : h->inst.C).M.size(); }
void orc_instance_export(void* hv, uint64_t* rows, uint64_t* cols, uint64_t* vals, uint64_t* vars, uint64_t* inputs) {
  InstH* h = (InstH*)hv; size_t off = 0;
  for (const SparseMatPoly* m : {&h->inst.A, &h->inst.B, &h->inst.C})
    for (auto& e : m->M) { rows[off] = e.row; cols[off] = e.col; memcpy(vals + 4 * off, e.val.l, 32); off++; }
  out_vec(h->vars, vars); out_vec(h->inputs, inputs);
}
int orc_instance_is_sat(void* hv) { InstH* h = (InstH*)hv; return h->inst.is_sat(h->vars, h->inputs) ? 1 : 0; }
size_t orc_instance_shape_bincode(void* hv, uint8_t* out, size_t cap) {
  std::vector<uint8_t> b = ser_r1cs_shape(((InstH*)hv)->inst);
  if (out && cap >= b.size()) memcpy(out, b.data(), b.size());
  return b.size();
}
void orc_seed_scalar(const char* domain, uint64_t seed, uint64_t out[4]) { Fq s = seed_scalar(domain, seed); memcpy(out, s.l, 32); }

// ---- gens / encode ----
void* orc_snark_gens_new(size_t num_cons, size_t num_vars, size_t num_inputs, size_t nnz) { return new SnarkGensH{SNARKGens::make(num_cons, num_vars, num_inputs, nnz)}; }
void orc_snark_gens_free(void* g) { delete (SnarkGensH*)g; }
void* orc_nizk_gens_new(size_t num_cons, size_t num_vars, size_t num_inputs) { return new NizkGensH{NIZKGens::make(num_cons, num_vars, num_inputs)}; }
void orc_nizk_gens_free(void* g) { delete (NizkGensH*)g; }
// compressed generators of MultiCommitGens::new(n, label): n points G then h
void orc_multi_commit_gens(size_t n, const char* label, uint8_t* out) {
  MultiCommitGens g = MultiCommitGens::make(n, label);
  for (size_t i = 0; i < n; i++) pt_compress(g.G[i], out + 32 * i);
  pt_compress(g.h, out + 32 * n);
}
void* orc_snark_encode(void* inst, void* gens) {
  EncH* e = new EncH;
  r1cs_commit(((InstH*)inst)->inst, ((SnarkGensH*)gens)->g.gens_r1cs_eval, &e->comm, &e->decomm);
  return e;
}
void orc_encode_free(void* e) { delete (EncH*)e; }
// the two commitment vectors of ComputationCommitment (for parity against the product's encode)
size_t orc_encode_comm(void* ev, int which, uint8_t* out, size_t cap) {
  EncH* e = (EncH*)ev; const PolyCommitment& c = which == 0 ? e->comm.comm.comm_comb_ops : e->comm.comm.comm_comb_mem;
  if (out && cap >= 32 * c.C.size()) for (size_t i = 0; i < c.C.size(); i++) memcpy(out + 32 * i, c.C[i].data(), 32);
  return c.C.size();
}

// ---- prove / verify ----
void* orc_snark_prove(void* inst, void* gens, void* enc, const char* transcript_label, const uint64_t tape_seed[4], double* times10) {
  InstH* I = (InstH*)inst; ProofH* p = new ProofH; p->is_snark = true;
  Transcript t(transcript_label); ProveTimes tm; memset(&tm, 0, sizeof tm);
  p->snark = snark_prove(I->inst, ((EncH*)enc)->comm, ((EncH*)enc)->decomm, I->vars, I->inputs, ((SnarkGensH*)gens)->g, t, limbs(tape_seed), &tm);
  p->bytes = ser_snark(p->snark); fill_times(tm, times10);
  return p;
}
// SNARK::prove / NIZK::prove continuing a caller-owned transcript (lib.rs:339-347, 501-509): state in, state out
void* orc_snark_prove_t(void* inst, void* gens, void* enc, uint8_t transcript_state[203], const uint64_t tape_seed[4]) {
  InstH* I = (InstH*)inst; ProofH* p = new ProofH; p->is_snark = true;
  Transcript t("x"); t.import_state(transcript_state); ProveTimes tm; memset(&tm, 0, sizeof tm);
  p->snark = snark_prove(I->inst, ((EncH*)enc)->comm, ((EncH*)enc)->decomm, I->vars, I->inputs, ((SnarkGensH*)gens)->g, t, limbs(tape_seed), &tm);
  p->bytes = ser_snark(p->snark);
  t.export_state(transcript_state);
  return p;
}
void* orc_nizk_prove_t(void* inst, void* gens, const uint8_t* digest, size_t digest_len, uint8_t transcript_state[203], const uint64_t tape_seed[4]) {
  InstH* I = (InstH*)inst; ProofH* p = new ProofH; p->is_snark = false;
  Transcript t("x"); t.import_state(transcript_state); ProveTimes tm; memset(&tm, 0, sizeof tm);
  std::vector<uint8_t> d(digest, digest + digest_len);
  p->nizk = nizk_prove(I->inst, d, I->vars, I->inputs, ((NizkGensH*)gens)->g, t, limbs(tape_seed), &tm);
  p->bytes = ser_nizk(p->nizk);
  t.export_state(transcript_state);
  return p;
}
int orc_snark_verify_t(void* proof, void* inst, void* gens, void* enc, uint8_t transcript_state[203]) {
  Transcript t("x"); t.import_state(transcript_state);
  int ok = snark_verify(((ProofH*)proof)->snark, ((EncH*)enc)->comm, ((InstH*)inst)->inputs, t, ((SnarkGensH*)gens)->g) ? 1 : 0;
  t.export_state(transcript_state);
  return ok;
}
int orc_snark_verify(void* proof, void* inst, void* gens, void* enc, const char* transcript_label) {
  Transcript t(transcript_label);
  return snark_verify(((ProofH*)proof)->snark, ((EncH*)enc)->comm, ((InstH*)inst)->inputs, t, ((SnarkGensH*)gens)->g) ? 1 : 0;
}
void* orc_nizk_prove(void* inst, void* gens, const uint8_t* digest, size_t digest_len, const char* transcript_label, const uint64_t tape_seed[4], double* times10) {
  InstH* I = (InstH*)inst; ProofH* p = new ProofH; p->is_snark = false;
  Transcript t(transcript_label); ProveTimes tm; memset(&tm, 0, sizeof tm);
  std::vector<uint8_t> d(digest, digest + digest_len);
  p->nizk = nizk_prove(I->inst, d, I->vars, I->inputs, ((NizkGensH*)gens)->g, t, limbs(tape_seed), &tm);
  p->bytes = ser_nizk(p->nizk); fill_times(tm, times10);
  return p;
}
int orc_nizk_verify(void* proof, void* inst, void* gens, const uint8_t* digest, size_t digest_len, const char* transcript_label) {
  Transcript t(transcript_label); std::vector<uint8_t> d(digest, digest + digest_len);
  return nizk_verify(((ProofH*)proof)->nizk, ((InstH*)inst)->inst, d, ((InstH*)inst)->inputs, t, ((NizkGensH*)gens)->g) ? 1 : 0;
}
size_t orc_proof_bytes(void* proof, uint8_t* out, size_t cap) {
  ProofH* p = (ProofH*)proof;
  if (out && cap >= p->bytes.size()) memcpy(out, p->bytes.data(), p->bytes.size());
  return p->bytes.size();
}
// lengths the reference prints (lib.rs:381,410 ; sparse_mlpoly.rs:1205-1210): sat proof, product layer proof, eval proof
void orc_proof_part_lens(void* proof, size_t out[3]) {
  ProofH* p = (ProofH*)proof;
  out[0] = ser_r1cs_proof(p->is_snark ? p->snark.r1cs_sat_proof : p->nizk.r1cs_sat_proof).size();
  out[1] = p->is_snark ? ser_product_layer_proof(p->snark.r1cs_eval_proof.proof_prod_layer).size() : 0;
  out[2] = p->is_snark ? ser_eval_proof(p->snark.r1cs_eval_proof).size() : 0;
}
// flip one byte-level field to check the verifier rejects (oracle self-check)
void orc_proof_tamper(void* proof, int what) {
  ProofH* p = (ProofH*)proof;
  R1CSProof& r = p->is_snark ? p->snark.r1cs_sat_proof : p->nizk.r1cs_sat_proof;
  if (what == 0) r.proof_eq_sc_phase2.z = r.proof_eq_sc_phase2.z + fq_one();
  if (what == 1) r.sc_proof_phase1.proofs[0].z[0] = r.sc_proof_phase1.proofs[0].z[0] + fq_one();
  if (what == 2 && p->is_snark) p->snark.inst_evals[0] = p->snark.inst_evals[0] + fq_one();
  if (what == 3 && p->is_snark) p->snark.r1cs_eval_proof.proof_hash_layer.eval_val[0] = p->snark.r1cs_eval_proof.proof_hash_layer.eval_val[0] + fq_one();
}
void orc_proof_free(void* p) { delete (ProofH*)p; }

// ---- kernel-level restatements (what tests compare each HIP kernel against) ----
// DensePolynomial::commit_inner (dense_mlpoly.rs:164-177): rows x cols scalars, generators given compressed
int orc_commit_rows(const uint8_t* G_comp, size_t n_gens, const uint8_t h_comp[32], const uint64_t* Z, size_t rows, size_t cols,
                    const uint64_t* blinds, uint8_t* out) {
  if (cols != n_gens) return -1;
  MultiCommitGens g; g.n = n_gens; g.G.resize(n_gens);
  for (size_t i = 0; i < n_gens; i++) if (!pt_decompress(G_comp + 32 * i, &g.G[i])) return -2;
  if (!pt_decompress(h_comp, &g.h)) return -2;
  FqVec z = limbs_vec(Z, rows * cols);
#pragma omp parallel for schedule(dynamic)
  for (size_t i = 0; i < rows; i++) {
    Fq b = blinds ? limbs(blinds + 4 * i) : fq_zero();
    pt_compress(commit_vec(&z[i * cols], cols, b, g), out + 32 * i);
  }
  return 0;
}
void orc_eq_evals(const uint64_t* r, size_t ell, uint64_t* out) { out_vec(eq_evals(limbs_vec(r, ell)), out); }
void orc_bound_top(uint64_t* Z, size_t len, const uint64_t r[4]) { DensePoly p(limbs_vec(Z, len)); p.bound_poly_var_top(limbs(r)); out_vec(p.Z, Z); }
void orc_bound_vecmat(const uint64_t* Z, size_t num_vars, const uint64_t* L, uint64_t* out) {
  DensePoly p(limbs_vec(Z, pow2(num_vars))); out_vec(p.bound(limbs_vec(L, pow2(num_vars / 2))), out);
}
// UniPoly::from_evals / compress / evaluate (unipoly.rs:23-88) for the reference's known answers (unipoly.rs:127-183)
void orc_unipoly_probe(const uint64_t* evals, size_t n, const uint64_t r[4], uint64_t* coeffs, uint64_t* compressed, uint64_t eval_at_r[4]) {
  UniPoly u = UniPoly::from_evals(limbs_vec(evals, n));
  out_vec(u.coeffs, coeffs); out_vec(u.compress(), compressed);
  Fq e = u.evaluate(limbs(r)); memcpy(eval_at_r, e.l, 32);
}
void orc_dot(const uint64_t* a, const uint64_t* b, size_t n, uint64_t out[4]) {
  Fq s = fq_zero(); for (size_t i = 0; i < n; i++) s += limbs(a + 4 * i) * limbs(b + 4 * i); memcpy(out, s.l, 32);
}
// sum-check round evaluations (sumcheck.rs:203-228, 460-469, 624-652). kind 0: A*B (out: e0,e2) ; 1: A*B*C (e0,e2,e3) ; 2: A*(B*C-D) (e0,e2,e3)
void orc_sumcheck_eval(int kind, const uint64_t* A, const uint64_t* B, const uint64_t* C, const uint64_t* D, size_t len, uint64_t* out) {
  size_t h = len / 2; Fq e0 = fq_zero(), e2 = fq_zero(), e3 = fq_zero();
  for (size_t i = 0; i < h; i++) {
    Fq a0 = limbs(A + 4 * i), a1 = limbs(A + 4 * (h + i)), b0 = limbs(B + 4 * i), b1 = limbs(B + 4 * (h + i));
    Fq a2 = a1 + a1 - a0, b2 = b1 + b1 - b0, a3 = a2 + a1 - a0, b3 = b2 + b1 - b0;
    if (kind == 0) { e0 += a0 * b0; e2 += a2 * b2; continue; }
    Fq c0 = limbs(C + 4 * i), c1 = limbs(C + 4 * (h + i)), c2 = c1 + c1 - c0, c3 = c2 + c1 - c0;
    if (kind == 1) { e0 += a0 * b0 * c0; e2 += a2 * b2 * c2; e3 += a3 * b3 * c3; continue; }
    Fq d0 = limbs(D + 4 * i), d1 = limbs(D + 4 * (h + i)), d2 = d1 + d1 - d0, d3 = d2 + d1 - d0;
    e0 += a0 * (b0 * c0 - d0); e2 += a2 * (b2 * c2 - d2); e3 += a3 * (b3 * c3 - d3);
  }
  memcpy(out, e0.l, 32); memcpy(out + 4, e2.l, 32); if (kind != 0) memcpy(out + 8, e3.l, 32);
}
}

// ---- verification of EXTERNAL proof bytes (the GPU path's output) by the oracle's restated verifier ----
namespace {
struct Rd {
  const uint8_t* p; size_t n, o; bool ok;
  Rd(const uint8_t* p_, size_t n_) : p(p_), n(n_), o(0), ok(true) {}
  uint64_t u64() { if (o + 8 > n) { ok = false; return 0; } uint64_t x; memcpy(&x, p + o, 8); o += 8; return x; }
  Fq fq() { Fq x = fq_zero(); if (o + 32 > n) { ok = false; return x; } memcpy(x.l, p + o, 32); o += 32;
            // a serialized Scalar must be a fully reduced Montgomery residue
            Fq t = fq_sub(x, FQ_MODULUS); (void)t; uint64_t b = 0; for (int i = 0; i < 4; i++) { u128 d = (u128)x.l[i] - FQ_MODULUS.l[i] - b; b = (uint64_t)(d >> 64) & 1; } if (!b) ok = false;
            return x; }
  CP cp() { CP c{}; if (o + 32 > n) { ok = false; return c; } memcpy(c.data(), p + o, 32); o += 32; return c; }
  FqVec fqv() { uint64_t k = u64(); FqVec v; if (k > (1u << 24)) { ok = false; return v; } for (uint64_t i = 0; i < k && ok; i++) v.push_back(fq()); return v; }
  std::vector<CP> cpv() { uint64_t k = u64(); std::vector<CP> v; if (k > (1u << 24)) { ok = false; return v; } for (uint64_t i = 0; i < k && ok; i++) v.push_back(cp()); return v; }
};
void r_dpp(Rd& r, DotProductProof& p) { p.delta = r.cp(); p.beta = r.cp(); p.z = r.fqv(); p.z_delta = r.fq(); p.z_beta = r.fq(); }
void r_zksc(Rd& r, ZKSumcheckProof& p) { p.comm_polys = r.cpv(); p.comm_evals = r.cpv(); uint64_t k = r.u64(); if (k > 64) { r.ok = false; return; } p.proofs.resize(k); for (auto& d : p.proofs) r_dpp(r, d); }
void r_eq(Rd& r, EqualityProof& p) { p.alpha = r.cp(); p.z = r.fq(); }
void r_pe(Rd& r, PolyEvalProof& p) { p.proof.bullet.L_vec = r.cpv(); p.proof.bullet.R_vec = r.cpv(); p.proof.delta = r.cp(); p.proof.beta = r.cp(); p.proof.z1 = r.fq(); p.proof.z2 = r.fq(); }
void r_r1cs(Rd& r, R1CSProof& p) {
  p.comm_vars.C = r.cpv(); r_zksc(r, p.sc_proof_phase1);
  for (int i = 0; i < 4; i++) p.claims_phase2[i] = r.cp();
  p.pok_Cz.alpha = r.cp(); p.pok_Cz.z1 = r.fq(); p.pok_Cz.z2 = r.fq();
  p.proof_prod.alpha = r.cp(); p.proof_prod.beta = r.cp(); p.proof_prod.delta = r.cp();
  for (int i = 0; i < 5; i++) p.proof_prod.z[i] = r.fq();
  r_eq(r, p.proof_eq_sc_phase1); r_zksc(r, p.sc_proof_phase2);
  p.comm_vars_at_ry = r.cp(); r_pe(r, p.proof_eval_vars_at_ry); r_eq(r, p.proof_eq_sc_phase2);
}
void r_batched(Rd& r, ProductCircuitEvalProofBatched& p) {
  uint64_t k = r.u64(); if (k > 64) { r.ok = false; return; }
  p.proof.resize(k);
  for (auto& l : p.proof) {
    uint64_t m = r.u64(); if (m > 64) { r.ok = false; return; }
    l.proof.compressed_polys.resize(m);
    for (auto& c : l.proof.compressed_polys) c = r.fqv();
    l.claims_prod_left = r.fqv(); l.claims_prod_right = r.fqv();
  }
  for (int i = 0; i < 3; i++) p.claims_dotp[i] = r.fqv();
}
void r_evalproof(Rd& r, SparseMatPolyEvalProof& p) {
  p.comm_derefs.C = r.cpv();
  ProductLayerProof& L = p.proof_prod_layer;
  L.row_init = r.fq(); L.row_read = r.fqv(); L.row_write = r.fqv(); L.row_audit = r.fq();
  L.col_init = r.fq(); L.col_read = r.fqv(); L.col_write = r.fqv(); L.col_audit = r.fq();
  L.eval_val[0] = r.fqv(); L.eval_val[1] = r.fqv();
  r_batched(r, L.proof_mem); r_batched(r, L.proof_ops);
  HashLayerProof& h = p.proof_hash_layer;
  h.row_addr = r.fqv(); h.row_read_ts = r.fqv(); h.row_audit_ts = r.fq();
  h.col_addr = r.fqv(); h.col_read_ts = r.fqv(); h.col_audit_ts = r.fq();
  h.eval_val = r.fqv(); h.eval_derefs[0] = r.fqv(); h.eval_derefs[1] = r.fqv();
  r_pe(r, h.proof_ops); r_pe(r, h.proof_mem); r_pe(r, h.proof_derefs);
}
}  // namespace

extern "C" {
// SNARK::verify (lib.rs:423-466) on bincode bytes produced elsewhere, against a computation commitment given as bytes
// (comm_comb_ops / comm_comb_mem: n x 32). Returns 1 accept, 0 reject, -1 malformed.
int orc_snark_verify_bytes(const uint8_t* proof, size_t proof_len, void* gens, size_t num_cons, size_t num_vars, size_t num_inputs,
                           size_t num_ops, size_t num_mem_cells, const uint8_t* comm_ops, size_t n_ops, const uint8_t* comm_mem, size_t n_mem,
                           const uint64_t* inputs, const char* transcript_label) {
  Rd r(proof, proof_len);
  SNARKProof P;
  r_r1cs(r, P.r1cs_sat_proof);
  for (int i = 0; i < 3; i++) P.inst_evals[i] = r.fq();
  r_evalproof(r, P.r1cs_eval_proof);
  if (!r.ok || r.o != proof_len) return -1;
  R1CSCommitment comm;
  comm.num_cons = num_cons; comm.num_vars = num_vars; comm.num_inputs = num_inputs;
  comm.comm.batch_size = 3; comm.comm.num_ops = num_ops; comm.comm.num_mem_cells = num_mem_cells;
  for (size_t i = 0; i < n_ops; i++) { CP c; memcpy(c.data(), comm_ops + 32 * i, 32); comm.comm.comm_comb_ops.C.push_back(c); }
  for (size_t i = 0; i < n_mem; i++) { CP c; memcpy(c.data(), comm_mem + 32 * i, 32); comm.comm.comm_comb_mem.C.push_back(c); }
  Transcript t(transcript_label);
  return snark_verify(P, comm, limbs_vec(inputs, num_inputs), t, ((SnarkGensH*)gens)->g) ? 1 : 0;
}
// NIZK::verify (lib.rs:549-587) on external bytes
int orc_nizk_verify_bytes(const uint8_t* proof, size_t proof_len, void* inst, void* gens, const uint8_t* digest, size_t digest_len,
                          const char* transcript_label) {
  Rd r(proof, proof_len);
  NIZKProof P;
  r_r1cs(r, P.r1cs_sat_proof);
  P.rx = r.fqv(); P.ry = r.fqv();
  if (!r.ok || r.o != proof_len) return -1;
  Transcript t(transcript_label);
  std::vector<uint8_t> d(digest, digest + digest_len);
  return nizk_verify(P, ((InstH*)inst)->inst, d, ((InstH*)inst)->inputs, t, ((NizkGensH*)gens)->g) ? 1 : 0;
}
}

extern "C" {
// Instance::new (lib.rs:121-228): padding of num_vars / num_cons, column shift for the constant/input columns, explicit
// zero rows when num_cons is 0 or 1, canonical-byte values. Returns NULL and sets *err (1 InvalidIndex, 2 InvalidScalar).
void* orc_instance_new_padded(size_t num_cons, size_t num_vars, size_t num_inputs, const size_t nnz[3], const uint64_t* rows,
                              const uint64_t* cols, const uint8_t* vals, const uint64_t* vars, size_t n_assigned_vars, const uint64_t* inputs,
                              int* err) {
  *err = 0;
  size_t nvp = num_vars > num_inputs + 1 ? num_vars : num_inputs + 1;   // :132-139
  nvp = next_pow2(nvp);
  size_t ncp = num_cons;                                                 // :143-156
  if (num_cons == 0 || num_cons == 1) ncp = 2;
  if (next_pow2(num_cons) != num_cons) ncp = next_pow2(num_cons);
  InstH* h = new InstH;
  h->inst.num_cons = ncp; h->inst.num_vars = nvp; h->inst.num_inputs = num_inputs;
  SparseMatPoly* m[3] = {&h->inst.A, &h->inst.B, &h->inst.C};
  size_t off = 0;
  for (int k = 0; k < 3; k++) {
    m[k]->num_vars_x = log_2(ncp); m[k]->num_vars_y = log_2(2 * nvp);
    for (size_t i = 0; i < nnz[k]; i++, off++) {
      if (rows[off] >= num_cons || cols[off] >= num_vars + 1 + num_inputs) { *err = 1; delete h; return nullptr; }   // :164-172
      Fq v;
      if (!fq_from_bytes(vals + 32 * off, &v)) { *err = 2; delete h; return nullptr; }                               // :174-186
      size_t col = cols[off] >= num_vars ? cols[off] + nvp - num_vars : cols[off];                                   // :178-182
      m[k]->M.push_back({(size_t)rows[off], col, v});
    }
    if (num_cons == 0 || num_cons == 1)                                                                                // :188-194
      for (size_t i = nnz[k]; i < ncp; i++) m[k]->M.push_back({i, num_vars, fq_zero()});
  }
  h->vars = limbs_vec(vars, n_assigned_vars);
  h->inputs = limbs_vec(inputs, num_inputs);
  return h;
}
}

// ---- wire formats of public parameters and the computation commitment (serde derives; bincode 1.3) ----
namespace {
struct Wb { std::vector<uint8_t> b; void u64(uint64_t x) { for (int i = 0; i < 8; i++) b.push_back((uint8_t)(x >> (8 * i))); }
            void pt(const Pt& p) { uint8_t c[32]; pt_compress(p, c); b.insert(b.end(), c, c + 32); } };
void wb_mcg(Wb& w, const MultiCommitGens& g) { w.u64(g.n); w.u64(g.G.size()); for (auto& p : g.G) w.pt(p); w.pt(g.h); }  // commitments.rs:7-12
void wb_pcg(Wb& w, const PolyCommitmentGens& g) { w.u64(g.gens.n); wb_mcg(w, g.gens.gens_n); wb_mcg(w, g.gens.gens_1); }   // dense_mlpoly.rs:24, nizk/mod.rs:407
}  // namespace
extern "C" {
size_t orc_snark_gens_bincode(void* gv, uint8_t* out, size_t cap) {  // lib.rs:278-282
  const SNARKGens& g = ((SnarkGensH*)gv)->g;
  Wb w;
  wb_mcg(w, g.gens_r1cs_sat.gens_sc.gens_1); wb_mcg(w, g.gens_r1cs_sat.gens_sc.gens_3); wb_mcg(w, g.gens_r1cs_sat.gens_sc.gens_4);
  wb_pcg(w, g.gens_r1cs_sat.gens_pc);
  wb_pcg(w, g.gens_r1cs_eval.gens_ops); wb_pcg(w, g.gens_r1cs_eval.gens_mem); wb_pcg(w, g.gens_r1cs_eval.gens_derefs);
  if (out && cap >= w.b.size()) memcpy(out, w.b.data(), w.b.size());
  return w.b.size();
}
// bincode of ComputationDecommitment (lib.rs:50-54 -> r1cs.rs:67-70 -> sparse_mlpoly.rs:274-282, 212-218; dense_mlpoly.rs:17-22):
// DensePolynomial = {num_vars, len, Vec<Scalar>}, AddrTimestamps = {Vec<Vec<usize>>, Vec<DensePolynomial> x2, DensePolynomial}
size_t orc_decommitment_bincode(void* ev, uint8_t* out, size_t cap) {
  const MultiSparseMatPolynomialAsDense& d = ((EncH*)ev)->decomm.dense;
  Wb w;
  auto poly = [&](const DensePoly& p) {
    w.u64(p.num_vars); w.u64(p.len); w.u64(p.Z.size());
    for (auto& x : p.Z) for (int i = 0; i < 4; i++) w.u64(x.l[i]);
  };
  auto polys = [&](const std::vector<DensePoly>& v) { w.u64(v.size()); for (auto& p : v) poly(p); };
  auto at = [&](const AddrTimestamps& a) {
    w.u64(a.ops_addr_usize.size());
    for (auto& v : a.ops_addr_usize) { w.u64(v.size()); for (size_t x : v) w.u64(x); }
    polys(a.ops_addr); polys(a.read_ts); poly(a.audit_ts);
  };
  w.u64(d.batch_size); polys(d.val); at(d.row); at(d.col); poly(d.comb_ops); poly(d.comb_mem);
  if (out && cap >= w.b.size()) memcpy(out, w.b.data(), w.b.size());
  return w.b.size();
}
size_t orc_commitment_bincode(void* ev, uint8_t* out, size_t cap) {  // lib.rs:44-48 -> r1cs.rs:50-56, sparse_mlpoly.rs:320-327
  const R1CSCommitment& c = ((EncH*)ev)->comm;
  Wb w;
  w.u64(c.num_cons); w.u64(c.num_vars); w.u64(c.num_inputs);
  w.u64(c.comm.batch_size); w.u64(c.comm.num_ops); w.u64(c.comm.num_mem_cells);
  w.u64(c.comm.comm_comb_ops.C.size()); for (auto& x : c.comm.comm_comb_ops.C) w.b.insert(w.b.end(), x.begin(), x.end());
  w.u64(c.comm.comm_comb_mem.C.size()); for (auto& x : c.comm.comm_comb_mem.C) w.b.insert(w.b.end(), x.begin(), x.end());
  if (out && cap >= w.b.size()) memcpy(out, w.b.data(), w.b.size());
  return w.b.size();
}
}
