// ORACLE (test infrastructure only — never linked into the product path).
// CPU restatement of libspartan 0.9.0's prover (and verifier, as the oracle's own self-check) for the hot
// path named by BASELINE.json: commitments.rs, dense_mlpoly.rs, unipoly.rs, sumcheck.rs, nizk/{mod,bullet}.rs,
// r1csproof.rs, r1cs.rs, sparse_mlpoly.rs, product_tree.rs, lib.rs. Each function cites the file:line it follows.
// Parity status: the reference's tests pin no commitment/challenge/proof byte; the oracle
// is pinned on (i) the F_q known answers of scalar/ristretto255.rs tests, (ii) RFC 9496 / libsodium for the
// group, (iii) the Merlin test vector + hashlib for the transcript, (iv) README.md:362,371,374 proof lengths,
// (v) the known answers of unipoly.rs:127-183 and dense_mlpoly.rs:433-452, (vi) its restated verifier accepting its own
// proofs (and those of the HIP path up to 2^22 constraints). Against real libspartan bytes it is "parity unpinned"
// (no Rust toolchain in this environment).
#pragma once
#include <array>
#include <cstdint>
#include <string>
#include <vector>

#include "fq.h"
#include "ristretto.h"
#include "transcript.h"

namespace orc {

typedef std::array<uint8_t, 32> CP;  // CompressedGroup (group.rs:7)
typedef std::vector<Fq> FqVec;

static inline CP compress(const Pt& p) { CP c; pt_compress(p, c.data()); return c; }
static inline Pt decompress(const CP& c) {
  Pt p;
  if (!pt_decompress(c.data(), &p)) { fprintf(stderr, "oracle: decompress failed\n"); abort(); }
  return p;
}

// math.rs:7-29
static inline size_t pow2(size_t e) { return (size_t)1 << e; }
static inline size_t log_2(size_t n) {  // ceil for non powers of two (math.rs:21-29)
  size_t l = 0;
  while (((size_t)1 << l) < n) l++;
  return l;
}
static inline size_t next_pow2(size_t n) { return pow2(log_2(n == 0 ? 1 : n)); }

// ---------------- commitments.rs ----------------
struct MultiCommitGens {
  size_t n;
  std::vector<Pt> G;
  Pt h;
  static MultiCommitGens make(size_t n, const char* label);  // commitments.rs:15-33
  MultiCommitGens scale(const Fq& s) const;                  // :43-49
  void split_at(size_t mid, MultiCommitGens* a, MultiCommitGens* b) const;  // :51-69
};
Pt commit_scalar(const Fq& x, const Fq& blind, const MultiCommitGens& g1);            // :73-78
Pt commit_vec(const Fq* v, size_t n, const Fq& blind, const MultiCommitGens& gn);     // :80-92

// ---------------- nizk/mod.rs gens, dense_mlpoly.rs gens ----------------
struct DotProductProofGens {
  size_t n;
  MultiCommitGens gens_n, gens_1;
  static DotProductProofGens make(size_t n, const char* label);  // nizk/mod.rs:415-418
};
struct PolyCommitmentGens {
  DotProductProofGens gens;
  static PolyCommitmentGens make(size_t num_vars, const char* label);  // dense_mlpoly.rs:31-35
};

// ---------------- dense_mlpoly.rs ----------------
FqVec eq_evals(const FqVec& r);                                  // EqPolynomial::evals :68-84
Fq eq_evaluate(const FqVec& r, const FqVec& rx);                 // EqPolynomial::evaluate :60-66
void eq_factored_evals(const FqVec& r, FqVec* L, FqVec* R);      // :90-98

struct DensePoly {
  size_t num_vars, len;
  FqVec Z;
  DensePoly() : num_vars(0), len(0) {}
  explicit DensePoly(FqVec z) : num_vars(log_2(z.size())), len(z.size()), Z(std::move(z)) {}  // :119-126
  const Fq& operator[](size_t i) const { return Z[i]; }
  void bound_poly_var_top(const Fq& r);   // :215-223
  void bound_poly_var_bot(const Fq& r);   // :225-233
  FqVec bound(const FqVec& L) const;      // :206-213
  Fq evaluate(const FqVec& r) const;      // :236-242
  void split(size_t idx, DensePoly* a, DensePoly* b) const;  // :140-146
  void extend(const DensePoly& o);        // :248-257
  static DensePoly merge(const std::vector<const DensePoly*>& polys);  // :259-272
  static DensePoly from_usize(const std::vector<size_t>& z);           // :274-280
};
struct PolyCommitment { std::vector<CP> C; };
// DensePolynomial::commit :179-204 ; blinds drawn from the tape when tape != nullptr, else zero
PolyCommitment poly_commit(const DensePoly& p, const PolyCommitmentGens& gens, RandomTape* tape, FqVec* blinds_out);
void append_poly_commitment(Transcript& t, const char* label, const PolyCommitment& c);  // :292-300

// ---------------- unipoly.rs ----------------
struct UniPoly {
  FqVec coeffs;
  static UniPoly from_evals(const FqVec& evals);  // :23-55
  size_t degree() const { return coeffs.size() - 1; }
  Fq evaluate(const Fq& r) const;                 // :72-80
  Fq eval_at_zero() const { return coeffs[0]; }
  Fq eval_at_one() const;
  FqVec compress() const;                         // :82-88 (coeffs_except_linear_term)
  static UniPoly decompress(const FqVec& c, const Fq& hint);  // :96-110
  void append_to_transcript(Transcript& t, const char* label) const;  // :112-120
};

// ---------------- proof structs (field order = bincode order) ----------------
struct KnowledgeProof { CP alpha; Fq z1, z2; };                       // nizk/mod.rs:15-20
struct EqualityProof { CP alpha; Fq z; };                             // :77-81
struct ProductProof { CP alpha, beta, delta; Fq z[5]; };              // :146-152
struct DotProductProof { CP delta, beta; FqVec z; Fq z_delta, z_beta; };  // :292-299
struct BulletReductionProof { std::vector<CP> L_vec, R_vec; };        // bullet.rs:15-19
struct DotProductProofLog { BulletReductionProof bullet; CP delta, beta; Fq z1, z2; };  // nizk/mod.rs:421-428
struct PolyEvalProof { DotProductProofLog proof; };                   // dense_mlpoly.rs:303-306
struct ZKSumcheckProof { std::vector<CP> comm_polys, comm_evals; std::vector<DotProductProof> proofs; };  // sumcheck.rs:64-69
struct SumcheckProof { std::vector<FqVec> compressed_polys; };        // sumcheck.rs:17-20
struct R1CSProof {                                                    // r1csproof.rs:21-37
  PolyCommitment comm_vars;
  ZKSumcheckProof sc_proof_phase1;
  CP claims_phase2[4];
  KnowledgeProof pok_Cz;
  ProductProof proof_prod;
  EqualityProof proof_eq_sc_phase1;
  ZKSumcheckProof sc_proof_phase2;
  CP comm_vars_at_ry;
  PolyEvalProof proof_eval_vars_at_ry;
  EqualityProof proof_eq_sc_phase2;
};

// ---------------- sparse matrices / R1CS (sparse_mlpoly.rs:19-38, r1cs.rs:18-26) ----------------
struct SparseMatEntry { size_t row, col; Fq val; };
struct SparseMatPoly {
  size_t num_vars_x, num_vars_y;
  std::vector<SparseMatEntry> M;
  size_t num_nz_entries() const { return next_pow2(M.size()); }             // sparse_mlpoly.rs:349-351
  FqVec multiply_vec(size_t rows, size_t cols, const FqVec& z) const;        // :454-464
  FqVec compute_eval_table_sparse(const FqVec& rx, size_t rows, size_t cols) const;  // :466-481
  Fq evaluate_with_tables(const FqVec& tx, const FqVec& ty) const;           // :429-438
};
struct R1CSShape {
  size_t num_cons, num_vars, num_inputs;
  SparseMatPoly A, B, C;
  bool is_sat(const FqVec& vars, const FqVec& input) const;                  // r1cs.rs:240-266
};
// r1cs.rs:160-238 with the OsRng replaced by a SHAKE256 stream keyed by `seed`
void produce_synthetic_r1cs(size_t num_cons, size_t num_vars, size_t num_inputs, uint64_t seed, R1CSShape* inst,
                            FqVec* vars, FqVec* inputs);
Fq seed_scalar(const char* domain, uint64_t seed);  // from_bytes_wide(SHAKE256(domain || LE64(seed))[0..64])

// ---------------- gens (r1csproof.rs:39-74, r1cs.rs:28-48, sparse_mlpoly.rs:302-337, lib.rs:278-308) ----------------
struct R1CSSumcheckGens { MultiCommitGens gens_1, gens_3, gens_4; };
struct R1CSGens {
  R1CSSumcheckGens gens_sc;
  PolyCommitmentGens gens_pc;
  static R1CSGens make(const char* label, size_t num_cons, size_t num_vars);
};
struct SparseMatPolyCommitmentGens {
  PolyCommitmentGens gens_ops, gens_mem, gens_derefs;
  static SparseMatPolyCommitmentGens make(const char* label, size_t nvx, size_t nvy, size_t nnz, size_t batch);
};
struct SNARKGens {
  R1CSGens gens_r1cs_sat;
  SparseMatPolyCommitmentGens gens_r1cs_eval;
  static SNARKGens make(size_t num_cons, size_t num_vars, size_t num_inputs, size_t num_nz_entries);  // lib.rs:287-308
};
struct NIZKGens {
  R1CSGens gens_r1cs_sat;
  static NIZKGens make(size_t num_cons, size_t num_vars, size_t num_inputs);  // lib.rs:474-485
};

// ---------------- SPARK structures (sparse_mlpoly.rs) ----------------
struct AddrTimestamps {                                              // :213-266
  std::vector<std::vector<size_t>> ops_addr_usize;
  std::vector<DensePoly> ops_addr, read_ts;
  DensePoly audit_ts;
};
struct MultiSparseMatPolynomialAsDense {                             // :268-276
  size_t batch_size;
  std::vector<DensePoly> val;
  AddrTimestamps row, col;
  DensePoly comb_ops, comb_mem;
};
struct SparseMatPolyCommitment {                                     // :339-346
  size_t batch_size, num_ops, num_mem_cells;
  PolyCommitment comm_comb_ops, comm_comb_mem;
};
struct R1CSCommitment { size_t num_cons, num_vars, num_inputs; SparseMatPolyCommitment comm; };  // r1cs.rs:50-56
struct R1CSDecommitment { MultiSparseMatPolynomialAsDense dense; };                                // r1cs.rs:67-70
void r1cs_commit(const R1CSShape& inst, const SparseMatPolyCommitmentGens& gens, R1CSCommitment* comm,
                 R1CSDecommitment* decomm);                          // SNARK::encode lib.rs:325-336 -> r1cs.rs:305-318

struct LayerProofBatched { SumcheckProof proof; FqVec claims_prod_left, claims_prod_right; };   // product_tree.rs:133-139
struct ProductCircuitEvalProofBatched { std::vector<LayerProofBatched> proof; FqVec claims_dotp[3]; };  // :162-166
struct ProductLayerProof {                                           // sparse_mlpoly.rs:1021-1028
  Fq row_init; FqVec row_read, row_write; Fq row_audit;
  Fq col_init; FqVec col_read, col_write; Fq col_audit;
  FqVec eval_val[2];
  ProductCircuitEvalProofBatched proof_mem, proof_ops;
};
struct HashLayerProof {                                              // :680-689
  FqVec row_addr, row_read_ts; Fq row_audit_ts;
  FqVec col_addr, col_read_ts; Fq col_audit_ts;
  FqVec eval_val;
  FqVec eval_derefs[2];
  PolyEvalProof proof_ops, proof_mem, proof_derefs;
};
struct SparseMatPolyEvalProof {                                      // :1418-1422, :1307-1311
  PolyCommitment comm_derefs;
  ProductLayerProof proof_prod_layer;
  HashLayerProof proof_hash_layer;
};
struct SNARKProof {                                                  // lib.rs:312-317
  R1CSProof r1cs_sat_proof;
  Fq inst_evals[3];
  SparseMatPolyEvalProof r1cs_eval_proof;
};
struct NIZKProof { R1CSProof r1cs_sat_proof; FqVec rx, ry; };        // lib.rs:489-493

// ---------------- provers / verifiers ----------------
struct ProveTimes {  // span names follow src/timer.rs call sites
  double polycommit, sc_phase_one, sc_phase_two, polyeval, r1cs_sat, eval_sparse_polys, commit_nondet_witness,
      build_layered_network, evalproof_layered_network, total;
};
// r1csproof.rs:144-349
R1CSProof r1cs_prove(const R1CSShape& inst, const FqVec& vars, const FqVec& input, const R1CSGens& gens, Transcript& t,
                     RandomTape& tape, FqVec* rx, FqVec* ry, ProveTimes* times);
// r1csproof.rs:351-491
bool r1cs_verify(const R1CSProof& p, size_t num_vars, size_t num_cons, const FqVec& input, const Fq evals[3],
                 Transcript& t, const R1CSGens& gens, FqVec* rx, FqVec* ry);
// lib.rs:339-420 / 423-466 ; tape_seed replaces OsRng (random.rs:13-15)
SNARKProof snark_prove(const R1CSShape& inst, const R1CSCommitment& comm, const R1CSDecommitment& decomm, const FqVec& vars,
                       const FqVec& inputs, const SNARKGens& gens, Transcript& t, const Fq& tape_seed, ProveTimes* times);
bool snark_verify(const SNARKProof& p, const R1CSCommitment& comm, const FqVec& inputs, Transcript& t, const SNARKGens& gens);
// lib.rs:501-546 / 549-587 ; digest = opaque bytes (zlib(bincode(shape)) in the reference, r1cs.rs:154-158)
NIZKProof nizk_prove(const R1CSShape& inst, const std::vector<uint8_t>& digest, const FqVec& vars, const FqVec& inputs,
                     const NIZKGens& gens, Transcript& t, const Fq& tape_seed, ProveTimes* times);
bool nizk_verify(const NIZKProof& p, const R1CSShape& inst, const std::vector<uint8_t>& digest, const FqVec& inputs,
                 Transcript& t, const NIZKGens& gens);

// ---------------- bincode 1.3 default encoding (fixed-width LE ints, u64 lengths) ----------------
std::vector<uint8_t> ser_r1cs_proof(const R1CSProof& p);
std::vector<uint8_t> ser_snark(const SNARKProof& p);
std::vector<uint8_t> ser_nizk(const NIZKProof& p);
std::vector<uint8_t> ser_product_layer_proof(const ProductLayerProof& p);
std::vector<uint8_t> ser_eval_proof(const SparseMatPolyEvalProof& p);
std::vector<uint8_t> ser_r1cs_shape(const R1CSShape& s);  // input of get_digest (r1cs.rs:154-158)

}  // namespace orc
