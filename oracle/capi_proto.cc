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
