// spartan_amd host driver: C++ mirror of libspartan's public API (src/lib.rs) for the prover path, on top of
// the C ABI in include/spartan_hip.h. It exists because no Rust toolchain is available here; in the drop-in
// deployment this layer IS libspartan (Rust) with the `gpu` feature. Names and argument
// meaning follow the reference: Instance, VarsAssignment/InputsAssignment, SNARKGens, NIZKGens,
// ComputationCommitment/Decommitment, SNARK::{encode,prove}, NIZK::prove. Verifiers are out of scope
// (SURVEY.md §2.1); tests verify the emitted bytes with the oracle's restated verifier.
//
// All table-sized field/group work of the prover runs on the GPU through sp_* calls; the host keeps what the
// reference keeps next to its Transcript — Fiat–Shamir, O(log n)-sized scalar bookkeeping, serialization — and the
// 2..5-term commitments of the Sigma protocols, whose dependent chains a host core finishes before a lone wavefront
// would (the library's host-side engine, sp_host_*).
#pragma once
#include <array>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <mutex>
#include <vector>

#include "../../include/spartan_hip.h"
#include "transcript.hpp"

namespace spz {

typedef std::array<uint8_t, 32> CP;  // CompressedGroup (src/group.rs:7)
typedef std::vector<Fq> FqVec;

struct Error : std::runtime_error {
  using std::runtime_error::runtime_error;
};

// value of a library option of the context (spartan_amd/csrc/options.hpp; sp_ctx_set_option): the driver's own switches live in the same table
inline long long ctx_opt(const sp_ctx* c, const char* key) {
  int64_t v = 0;
  if (sp_ctx_get_option(c, key, &v) != SP_OK) throw std::runtime_error(std::string("unknown library option ") + key);
  return (long long)v;
}

// ---- device handles (RAII) ----
struct Ctx {
  sp_ctx* h = nullptr;
  explicit Ctx(int device);
  ~Ctx();
  Ctx(const Ctx&) = delete;
};
// Row-sharded DensePolynomial::commit across the GPUs of one node (spartan_amd/host/shard.cc): every rank
// runs the same proof in lock-step; for a commitment of L rows rank r computes rows [r L/W, (r+1) L/W) and the 32-byte
// compressed commitments are exchanged with one all-gather of bytes (rows are independent MSMs: no elliptic-curve
// reduction exists in RCCL and none is needed). Three transports:
//   set_commit_shard_rccl     RCCL inside the library: rank 0 draws the id (rccl_unique_id), the caller hands it to every
//                             rank by whatever channel it has, each rank joins with ncclCommInitRank; commits then use
//                             ncclAllGather on device buffers. librccl.so is dlopen'ed on first use.
//   set_commit_shard          the caller moves the bytes: gather(user, buf, total, off, len) — on entry buf[off, off+len)
//                             holds this rank's bytes, on return buf[0, total) is complete (tests: gloo)
//   set_commit_shard_virtual  W shards on ONE GPU (W sub-contexts, in-process gather): the partitioning under test
// world <= 1 (or nshards <= 1) clears the setting. All ranks must then run identical prove() calls.
// CONTRACT OF THE TWO MULTI-PROCESS TRANSPORTS: configuring the sharding is itself a collective — set_commit_shard / set_commit_shard_rccl
// exchange 8 bytes per rank over the transport they were just given (check_switches_agree, shard.cc: ranks whose sharding options differ
// fail there with a message instead of deadlocking in the first proof). Every rank must therefore call it at the same point, and the
// gather callback must already work when it is handed over.
void unipoly_probe(const FqVec& evals, const Fq& r, FqVec* coeffs, FqVec* compressed, Fq* eval_at_r);  // test hook
void cubic_coeffs_probe(const Fq S[12], const Fq& r, Fq ev[3]);                                                // test hook (spark.inc)
void cubic_tail_probe(FqVec& tab, size_t ni, size_t m, const FqVec& coeffs, const FqVec& challenges, FqVec* evs);  // test hook (spark.inc)
bool eq_factor_probe(const FqVec& rho, size_t np, size_t ni, const FqVec& coeffs, const FqVec& claims, const FqVec& ev4, const FqVec& challenges, FqVec* evc_out,
                     FqVec* K_out);  // test hook (spark.inc)
typedef int (*CommitGatherFn)(void* user, uint8_t* buf, size_t total, size_t off, size_t len);
void set_commit_shard(Ctx& c, int rank, int world, CommitGatherFn gather, void* user);
void rccl_unique_id(uint8_t out[128]);
void set_commit_shard_rccl(Ctx& c, int rank, int world, const uint8_t unique_id[128]);
void set_commit_shard_virtual(Ctx& c, int nshards);
struct ShardStats { size_t gathers = 0, bytes = 0; };  // exchanges since the last reset (bench.py reports them per proof)
ShardStats commit_shard_stats(Ctx& c, bool reset);
// internal to the driver (prover.cc <-> shard.cc)
bool commit_shard_active(sp_ctx* c);
std::vector<sp_ctx*> residue_shard_ctxs(sp_ctx* c);  // virtual shards: [c, sub-contexts...]; empty when none
void commit_shard_note_gather(sp_ctx* c, size_t bytes);
// callback / RCCL transport configured: this rank's place among the lock-step ranks, and the table length (log2) from which a sum-check is
// residue-sharded over it (resolved once when the sharding was configured, identical on every rank: shard.cc, check_switches_agree)
bool commit_shard_transport(sp_ctx* c, int* rank, int* world, int* residue_min_log2 = nullptr);
bool commit_shard_residue_off(sp_ctx* c);  // option shard.residues = 0 as resolved when the sharding was configured
void commit_shard_gather(sp_ctx* c, uint8_t* all, size_t per);   // all-gather of `per` bytes per rank over that transport (rank order)
double rccl_allgather_probe(sp_ctx* c, size_t bytes, int iters);  // us per H2D + ncclAllGather + D2H + sync of `bytes` per rank
bool commit_shard_shared_seed(sp_ctx* c, Fq* seed);  // multi-rank transports only: a hash of every rank's OS-entropy contribution, the same on every rank
void commit_shard_forget(sp_ctx* c);
bool sharded_commit_rows(sp_ctx* c, const sp_gens* g, size_t g_off, size_t h_idx, const sp_table* Z, size_t Ls, size_t Rs, const uint64_t* blinds,
                         uint8_t* out);
struct DevTable {  // DensePolynomial with Z resident in HBM (src/dense_mlpoly.rs:14-18)
  sp_ctx* c = nullptr;
  sp_table* h = nullptr;
  DevTable() {}
  DevTable(sp_ctx* c_, sp_table* h_) : c(c_), h(h_) {}
  DevTable(DevTable&& o) noexcept : c(o.c), h(o.h) { o.h = nullptr; }
  DevTable& operator=(DevTable&& o) noexcept;
  DevTable(const DevTable&) = delete;
  ~DevTable();
  size_t len() const { return sp_table_len(h); }
};

// MultiCommitGens (src/commitments.rs:8-13) as a view into a device generator stream
struct MultiCommitGens {
  sp_gens* g = nullptr;
  std::vector<uint32_t> G;  // stream indices of G[0..n)
  uint32_t h = 0;           // stream index of h
  size_t n() const { return G.size(); }
};
struct DotProductProofGens {  // src/nizk/mod.rs:408-419
  size_t n;
  MultiCommitGens gens_n, gens_1;
};
struct PolyCommitmentGens {  // src/dense_mlpoly.rs:25-36
  DotProductProofGens gens;
};
struct GensStream {  // one SHAKE256 stream of points (label), uploaded once with its window tables
  sp_ctx* c = nullptr;
  sp_gens* g = nullptr;
  std::vector<uint8_t> compressed;
  GensStream() {}
  GensStream(sp_ctx* c, const char* label, size_t npoints, int windows = 0 /* table geometry planned by the caller, or 0: the library's per-set policy */);
  GensStream(GensStream&& o) noexcept : c(o.c), g(o.g), compressed(std::move(o.compressed)) { o.g = nullptr; }
  GensStream& operator=(GensStream&& o) noexcept;
  ~GensStream();
  MultiCommitGens multi_commit_gens(size_t n) const;        // MultiCommitGens::new(n, label)
  DotProductProofGens dot_product_gens(size_t n) const;     // DotProductProofGens::new(n, label)
  PolyCommitmentGens poly_commitment_gens(size_t num_vars) const;
};
struct R1CSSumcheckGens { MultiCommitGens gens_1, gens_3, gens_4; };  // src/r1csproof.rs:39-60
struct R1CSGens {                                                      // src/r1csproof.rs:62-74
  R1CSSumcheckGens gens_sc;
  PolyCommitmentGens gens_pc;
};
struct SparseMatPolyCommitmentGens { PolyCommitmentGens gens_ops, gens_mem, gens_derefs; };  // src/sparse_mlpoly.rs:302-337

struct NIZKGens {  // src/lib.rs:468-486
  GensStream stream_sat;
  R1CSGens gens_r1cs_sat;
  NIZKGens(Ctx& ctx, size_t num_cons, size_t num_vars, size_t num_inputs);
};
struct SNARKGens {  // src/lib.rs:276-309
  GensStream stream_sat, stream_eval;
  R1CSGens gens_r1cs_sat;
  SparseMatPolyCommitmentGens gens_r1cs_eval;
  SNARKGens(Ctx& ctx, size_t num_cons, size_t num_vars, size_t num_inputs, size_t num_nz_entries);
  // bincode of the serde-derived struct (lib.rs:278-282 -> r1csproof.rs:39-66, r1cs.rs:28-31, sparse_mlpoly.rs:284-289,
  // dense_mlpoly.rs:24-27, nizk/mod.rs:407-412, commitments.rs:7-12): every MultiCommitGens is {n, Vec<G>, h}, points as 32 bytes
  std::vector<uint8_t> serialize() const;
};

// ---- instance ----
struct SparseEntry { uint64_t row, col; Fq val; };
struct Instance {  // src/lib.rs:110-273 (R1CSInstance after padding) with the matrices resident on the device
  sp_ctx* c = nullptr;
  size_t num_cons = 0, num_vars = 0, num_inputs = 0;
  std::vector<SparseEntry> A, B, C;
  sp_sparse *dA = nullptr, *dB = nullptr, *dC = nullptr;
  // R1CSShapeDigest bytes: zlib(level 6)(bincode(shape)) (r1cs.rs:154-158), absorbed by NIZK::prove (lib.rs:514). Computed on first
  // use by compute_digest() (deflate.cc: a restatement of miniz's level-6 tdefl, what flate2's rust_backend runs — byte-identical to the
  // real C miniz 3.0.2 on the test corpus); a caller that holds the bytes of a real libspartan Instance may set them instead
  // (set_digest). The computation is guarded: concurrent NIZK::prove calls over one Instance compute it once; a caller that wants the
  // deflate of a large shape (~6 s at 2^20) out of its first prove calls compute_digest() right after construction, as the reference's
  // Instance::new does (lib.rs:228).
  mutable std::vector<uint8_t> digest;
  bool digest_old_header = false;  // zlib header 0x78 0x01 (miniz < 2.2, miniz_oxide 0.3) instead of 0x78 0x9C; set before the first compute_digest()
  mutable std::mutex digest_mu;
  std::vector<uint8_t> shape_bincode() const;  // bincode(R1CSShape): r1cs.rs:18-26, sparse_mlpoly.rs:19-38
  std::vector<uint8_t> compute_digest() const;
  bool set_digest_header(bool old_header);
  void set_digest(const uint8_t* d, size_t n);
  // Instance::new (lib.rs:121-228): padding of num_cons / num_vars and the column shift are applied here.
  Instance(Ctx& ctx, size_t num_cons, size_t num_vars, size_t num_inputs, const std::vector<SparseEntry>& A,
           const std::vector<SparseEntry>& B, const std::vector<SparseEntry>& C);
  ~Instance();
  Instance(const Instance&) = delete;
  // Instance::produce_synthetic_r1cs (lib.rs:262-273 -> r1cs.rs:160-238) with the OsRng replaced by a SHAKE256
  // stream keyed by `seed` ("spartan-synthetic-r1cs" || LE64(seed)); returns the satisfying assignment.
  static std::unique_ptr<Instance> produce_synthetic_r1cs(Ctx& ctx, size_t num_cons, size_t num_vars, size_t num_inputs, uint64_t seed,
                                                          FqVec* vars, FqVec* inputs);
};
// VarsAssignment (lib.rs:56-105, Assignment::new) with the scalars resident in HBM: the reference parses the caller's bytes
// into Scalars in the constructor, outside prove; here the constructor also uploads them, once, and every proof over the
// assignment starts from the device copy (one 32 MB PCIe transfer less per 2^20 proof). The padding to the instance's
// num_vars (lib.rs:360-368) stays in prove.
struct VarsAssignment {
  sp_ctx* c = nullptr;
  DevTable tab;
  size_t n = 0;
  VarsAssignment(Ctx& ctx, const Fq* vars, size_t n);
};
// TEST/BENCH ONLY: from_bytes_wide(SHAKE256(domain || LE64(seed))[0..64]), the documented seed -> scalar map behind the
// reproducible RandomTape seeds of tests/ and bench.py. 64 bits of entropy: never a production tape seed.
Fq seed_scalar(const char* domain, uint64_t seed);

// zlib stream of `data` as miniz's tdefl produces it at level 6 (deflate.cc). old_header: 0x78 0x01 (miniz, miniz_oxide 0.3) instead of 0x78 0x9C
std::vector<uint8_t> zlib_level6_miniz(const uint8_t* data, size_t n, bool old_header = false);
// the same compressor at another probe count (tdefl flags & 0xFFF: 16/32/128/256/512/768/1500 = miniz levels 4..10); tests pin each against miniz
std::vector<uint8_t> zlib_miniz_probes(const uint8_t* data, size_t n, unsigned probes, bool old_header = false);

// ---- proof structs: field order == bincode order (same as the reference's serde derives) ----
struct PolyCommitment { std::vector<CP> C; };                                   // dense_mlpoly.rs:38-41
struct KnowledgeProof { CP alpha; Fq z1, z2; };                                 // nizk/mod.rs:15-20
struct EqualityProof { CP alpha; Fq z; };                                       // :77-81
struct ProductProof { CP alpha, beta, delta; Fq z[5]; };                        // :146-152
struct DotProductProof { CP delta, beta; FqVec z; Fq z_delta, z_beta; };        // :292-299
struct BulletReductionProof { std::vector<CP> L_vec, R_vec; };                  // bullet.rs:15-19
struct DotProductProofLog { BulletReductionProof bullet; CP delta, beta; Fq z1, z2; };  // nizk/mod.rs:421-428
struct PolyEvalProof { DotProductProofLog proof; };                             // dense_mlpoly.rs:303-306
struct ZKSumcheckInstanceProof { std::vector<CP> comm_polys, comm_evals; std::vector<DotProductProof> proofs; };  // sumcheck.rs:64-69
struct SumcheckInstanceProof { std::vector<FqVec> compressed_polys; };          // sumcheck.rs:17-20
struct R1CSProof {                                                              // r1csproof.rs:21-37
  PolyCommitment comm_vars;
  ZKSumcheckInstanceProof sc_proof_phase1;
  CP claims_phase2[4];
  KnowledgeProof pok_claims_phase2;
  ProductProof proof_prod;
  EqualityProof proof_eq_sc_phase1;
  ZKSumcheckInstanceProof sc_proof_phase2;
  CP comm_vars_at_ry;
  PolyEvalProof proof_eval_vars_at_ry;
  EqualityProof proof_eq_sc_phase2;
};
struct LayerProofBatched { SumcheckInstanceProof proof; FqVec claims_prod_left, claims_prod_right; };  // product_tree.rs:133-139
struct ProductCircuitEvalProofBatched { std::vector<LayerProofBatched> proof; FqVec claims_dotp[3]; };  // :162-166
struct ProductLayerProof {                                                      // sparse_mlpoly.rs:1021-1028
  Fq row_init; FqVec row_read, row_write; Fq row_audit;
  Fq col_init; FqVec col_read, col_write; Fq col_audit;
  FqVec eval_val[2];
  ProductCircuitEvalProofBatched proof_mem, proof_ops;
};
struct HashLayerProof {                                                         // :680-689
  FqVec row_addr, row_read_ts; Fq row_audit_ts;
  FqVec col_addr, col_read_ts; Fq col_audit_ts;
  FqVec eval_val;
  FqVec eval_derefs[2];
  PolyEvalProof proof_ops, proof_mem, proof_derefs;
};
struct SparseMatPolyEvalProof {                                                 // :1418-1422, :1307-1311
  PolyCommitment comm_derefs;
  ProductLayerProof proof_prod_layer;
  HashLayerProof proof_hash_layer;
};

// ---- SPARK dense representation on the device (sparse_mlpoly.rs:213-276) ----
struct DevIndex {  // Vec<usize> resident on the device
  sp_ctx* c = nullptr;
  sp_index* h = nullptr;
  DevIndex() {}
  DevIndex(sp_ctx* c_, sp_index* h_) : c(c_), h(h_) {}
  DevIndex(DevIndex&& o) noexcept : c(o.c), h(o.h) { o.h = nullptr; }
  DevIndex& operator=(DevIndex&& o) noexcept;
  DevIndex(const DevIndex&) = delete;
  ~DevIndex();
};
struct AddrTimestamps {
  std::vector<DevIndex> ops_addr_usize;
  std::vector<DevTable> ops_addr, read_ts;
  DevTable audit_ts;
};
struct MultiSparseMatPolynomialAsDense {
  size_t batch_size = 0, num_ops = 0, num_mem_cells = 0;
  std::vector<DevTable> val;
  AddrTimestamps row, col;
  DevTable comb_ops, comb_mem;
};
struct SparseMatPolyCommitment {  // :339-346
  size_t batch_size, num_ops, num_mem_cells;
  PolyCommitment comm_comb_ops, comm_comb_mem;
};
struct ComputationCommitment {  // lib.rs:44-48 -> r1cs.rs:50-56
  size_t num_cons, num_vars, num_inputs;
  SparseMatPolyCommitment comm;
  std::vector<uint8_t> serialize() const;  // bincode (r1cs.rs:50-56, sparse_mlpoly.rs:320-327, dense_mlpoly.rs:42-45)
};
struct ComputationDecommitment {  // lib.rs:50-54 -> r1cs.rs:67-70
  MultiSparseMatPolynomialAsDense dense;
  // bincode of the serde-derived struct (sparse_mlpoly.rs:274-282, 212-218; dense_mlpoly.rs:17-22): every DensePolynomial is
  // {num_vars, len, Vec<Scalar>}; the tables are downloaded from the device (1.3 GB at 2^20: an interchange format, not a hot path)
  std::vector<uint8_t> serialize() const;
};

struct ProveTimes {  // span names follow src/timer.rs call sites
  double polycommit = 0, sc_phase_one = 0, sc_phase_two = 0, polyeval = 0, r1cs_sat = 0, eval_sparse_polys = 0, commit_nondet_witness = 0,
         build_layered_network = 0, evalproof_layered_network = 0, total = 0;
};

struct SNARK {  // lib.rs:311-467
  R1CSProof r1cs_sat_proof;
  Fq inst_evals[3];
  SparseMatPolyEvalProof r1cs_eval_proof;
  // SNARK::encode (lib.rs:325-336)
  static void encode(Ctx& ctx, const Instance& inst, const SNARKGens& gens, ComputationCommitment* comm, ComputationDecommitment* decomm);
  // SNARK::prove (lib.rs:339-420). tape_seed == nullptr: the RandomTape is seeded from OS entropy like RandomTape::new
  // (random.rs:13-15) — the production setting. A non-null seed is the `new_with_seed` test hook: it determines every blind,
  // so it must be secret, >= 256 bits of entropy and single-use, or the proof is no longer zero-knowledge.
  static SNARK prove(Ctx& ctx, const Instance& inst, const ComputationCommitment& comm, const ComputationDecommitment& decomm,
                     const FqVec& vars, const FqVec& inputs, const SNARKGens& gens, Transcript& transcript, const Fq* tape_seed,
                     ProveTimes* times = nullptr) {
    return prove(ctx, inst, comm, decomm, vars.data(), vars.size(), inputs, gens, transcript, tape_seed, times);
  }
  // same, reading the assignment in place (a Rust `&[Scalar]` / a caller-owned buffer): no host-side copy
  static SNARK prove(Ctx& ctx, const Instance& inst, const ComputationCommitment& comm, const ComputationDecommitment& decomm,
                     const Fq* vars, size_t num_vars_given, const FqVec& inputs, const SNARKGens& gens, Transcript& transcript,
                     const Fq* tape_seed, ProveTimes* times = nullptr, const sp_table* vars_resident = nullptr);
  // same, from an assignment already in HBM
  static SNARK prove(Ctx& ctx, const Instance& inst, const ComputationCommitment& comm, const ComputationDecommitment& decomm,
                     const VarsAssignment& vars, const FqVec& inputs, const SNARKGens& gens, Transcript& transcript, const Fq* tape_seed,
                     ProveTimes* times = nullptr) {
    return prove(ctx, inst, comm, decomm, nullptr, vars.n, inputs, gens, transcript, tape_seed, times, vars.tab.h);
  }
  std::vector<uint8_t> serialize() const;  // bincode 1.3 default encoding
};
struct NIZK {  // lib.rs:488-587
  R1CSProof r1cs_sat_proof;
  FqVec rx, ry;
  static NIZK prove(Ctx& ctx, const Instance& inst, const FqVec& vars, const FqVec& inputs, const NIZKGens& gens, Transcript& transcript,
                    const Fq* tape_seed, ProveTimes* times = nullptr) {
    return prove(ctx, inst, vars.data(), vars.size(), inputs, gens, transcript, tape_seed, times);
  }
  static NIZK prove(Ctx& ctx, const Instance& inst, const Fq* vars, size_t num_vars_given, const FqVec& inputs, const NIZKGens& gens,
                    Transcript& transcript, const Fq* tape_seed, ProveTimes* times = nullptr, const sp_table* vars_resident = nullptr);
  static NIZK prove(Ctx& ctx, const Instance& inst, const VarsAssignment& vars, const FqVec& inputs, const NIZKGens& gens, Transcript& transcript,
                    const Fq* tape_seed, ProveTimes* times = nullptr) {
    return prove(ctx, inst, nullptr, vars.n, inputs, gens, transcript, tape_seed, times, vars.tab.h);
  }
  std::vector<uint8_t> serialize() const;
};

std::vector<uint8_t> serialize_r1cs_proof(const R1CSProof& p);

// Where the few-term commitments of the Sigma protocols run: on the proving thread's core through the library's host-side
// engine (sp_host_commit_small / sp_host_zk_ahead_*, csrc/host_commit.hip — the default) or on the GPU (sp_msm_indexed;
// option commit.small_device = 1). Byte-identical proofs either way.
inline bool small_msm_on_host(const sp_ctx* c) { return ctx_opt(c, "commit.small_device") == 0; }

}  // namespace spz
