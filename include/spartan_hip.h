/*
 * spartan_hip.h — C ABI of the MI355X-native prover hot path for libspartan (microsoft/Spartan 0.9.0).
 *
 * The reference has NO plugin/FFI seam (all modules are private, src/lib.rs:16-31); this header is the seam
 * a `gpu` cargo feature would bind (INTEGRATION.md shows the Rust `extern "C"` block and the call sites).
 * Each entry point names the reference function body it replaces. Conventions:
 *   - scalars cross as raw `[u64;4]` little-endian Montgomery limbs, exactly as they sit in a Rust
 *     `&[Scalar]` (src/scalar/ristretto255.rs:199) — 4 uint64_t per element, always < q;
 *   - group elements cross ONLY as 32-byte CompressedRistretto (src/group.rs:7);
 *   - matrices are row-major; multilinear tables put r[0] on the most significant index bit
 *     (src/dense_mlpoly.rs:68-84);
 *   - every function returns 0 (SP_OK) or a negative sp_status; nothing throws across the boundary. The
 *     reference prover panics on these conditions (assert!/unwrap), so the Rust shim does `assert_eq!(rc, 0)`;
 *   - a context is bound to ONE GPU (one process per GPU) and may be used from one thread at a time. All
 *     calls are synchronous: results are on the host when the call returns;
 *   - the library fails with SP_EHIP when no gfx950 device is present. There is no CPU fallback.
 */
#ifndef SPARTAN_HIP_H
#define SPARTAN_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum sp_status {
  SP_OK = 0,
  SP_EINVAL = -1, /* size mismatch, non power of two, null pointer, index out of range */
  SP_ENOMEM = -2, /* hipMalloc failed */
  SP_EHIP = -3,   /* HIP runtime error / no device */
  SP_EPOINT = -4  /* a 32-byte string is not a valid ristretto255 encoding */
} sp_status;

typedef struct sp_ctx sp_ctx;     /* one GPU, one stream, scratch buffers */
typedef struct sp_gens sp_gens;   /* device-resident generator list with fixed-base window tables */
typedef struct sp_table sp_table; /* device-resident vector of F_q elements (a DensePolynomial's Z) */

const char* sp_strerror(int32_t status);
const char* sp_version(void);

/* ---- context --------------------------------------------------------------------------------------- */
int32_t sp_ctx_create(int device_id, sp_ctx** out);
void sp_ctx_destroy(sp_ctx* ctx);
/* HIP-event timing of every kernel family on the context's stream (bench.py's roofline numbers). */
int32_t sp_prof_enable(sp_ctx* ctx, int on);
int32_t sp_prof_reset(sp_ctx* ctx);
/* Fills up to cap entries; returns number of kernel families. name[i] is a static string. */
int32_t sp_prof_read(sp_ctx* ctx, const char** names, double* total_ms, uint64_t* launches, double* alg_bytes, int cap);

/* ---- generators: MultiCommitGens (src/commitments.rs:8-33) ------------------------------------------
 * A sp_gens is a list of n points P[0..n). A MultiCommitGens{G[0..m), h} made by
 * MultiCommitGens::new(m, label) is the list of its m+1 stream points with h = P[m]; gens that are prefixes
 * of one SHAKE stream (gens_3/gens_4/gens_pc of R1CSGens, src/r1csproof.rs:48-73) share one sp_gens.
 * Upload builds signed 8-bit fixed-base window tables (32 windows x 128 affine entries per point, 384 KiB
 * per point in HBM): generators are public parameters reused across proofs, so this is setup cost. */
int32_t sp_gens_upload(sp_ctx* ctx, const uint8_t* compressed /*32*n*/, size_t n, sp_gens** out);
/* MultiCommitGens::new body (commitments.rs:21-30): n blocks of 64 uniform bytes from the caller's
 * SHAKE256 stream -> from_uniform_bytes on the device. compressed_out (32*n) may be NULL. */
int32_t sp_gens_from_uniform(sp_ctx* ctx, const uint8_t* uniform /*64*n*/, size_t n, uint8_t* compressed_out, sp_gens** out);
size_t sp_gens_len(const sp_gens* g);
void sp_gens_free(sp_gens* g);

/* ---- Pedersen commitments (src/commitments.rs:73-92, src/dense_mlpoly.rs:164-177, src/group.rs:98-117) --
 * out[i] = compress( sum_j Z[i*cols+j] * P[g_off+j]  +  blinds[i] * P[h_idx] ),  i < rows.
 * Replaces DensePolynomial::commit_inner (rows = L) and [Scalar]::commit (rows = 1). blinds may be NULL (=0). */
int32_t sp_commit_rows(sp_ctx* ctx, const sp_gens* g, size_t g_off, size_t h_idx, const uint64_t* Z, size_t rows, size_t cols,
                       const uint64_t* blinds, uint8_t* out /*32*rows*/);
/* Same with Z taken from a device table: elements [z_off, z_off + rows*cols). */
int32_t sp_commit_rows_dev(sp_ctx* ctx, const sp_gens* g, size_t g_off, size_t h_idx, const sp_table* Z, size_t z_off, size_t rows,
                           size_t cols, const uint64_t* blinds, uint8_t* out);
/* Small/irregular commits (Scalar::commit, UniPoly::commit, the Sigma-protocol commitments of
 * src/nizk/mod.rs, the per-round L/R of src/nizk/bullet.rs:83-97 re-expressed over the ORIGINAL generators):
 * out[i] = compress( sum_j S[i*cols+j] * P[idx[j]] ). */
int32_t sp_msm_indexed(sp_ctx* ctx, const sp_gens* g, const uint32_t* idx, size_t cols, const uint64_t* S, size_t rows, uint8_t* out);

/* ---- device tables (DensePolynomial.Z, src/dense_mlpoly.rs:14-18) ----------------------------------- */
int32_t sp_table_alloc(sp_ctx* ctx, size_t len, sp_table** out);          /* zero-filled */
int32_t sp_table_upload(sp_ctx* ctx, const uint64_t* Z, size_t len, sp_table** out);
int32_t sp_table_write(sp_ctx* ctx, sp_table* t, size_t off, const uint64_t* Z, size_t len);
int32_t sp_table_download(sp_ctx* ctx, const sp_table* t, size_t off, size_t len, uint64_t* out);
int32_t sp_table_clone(sp_ctx* ctx, const sp_table* t, sp_table** out);
int32_t sp_table_copy(sp_ctx* ctx, sp_table* dst, size_t dst_off, const sp_table* src, size_t src_off, size_t len);
size_t sp_table_len(const sp_table* t); /* current (bound) length */
void sp_table_free(sp_table* t);

/* EqPolynomial::evals (dense_mlpoly.rs:68-84): chi_b(r) for all b in {0,1}^ell, r[0] <-> MSB. */
int32_t sp_eq_expand(sp_ctx* ctx, const uint64_t* r /*4*ell*/, size_t ell, sp_table** out);
/* Sum-check round evaluations over the current length of the tables (t = 0, 2[, 3]):
 *   kind 0: A*B          -> out[2]  (sumcheck.rs:460-469, comb of r1csproof.rs:122-123)
 *   kind 1: A*B*C        -> out[3]  (sumcheck.rs:203-228 / 290-357)
 *   kind 2: A*(B*C - D)  -> out[3]  (sumcheck.rs:624-652, comb of r1csproof.rs:87-91)        */
int32_t sp_sumcheck_eval(sp_ctx* ctx, int kind, sp_table* const* tabs, size_t ntabs, uint64_t* out_evals);
/* DensePolynomial::bound_poly_var_top (dense_mlpoly.rs:215-223) on every table: Z[i] += r*(Z[i+n]-Z[i]). */
int32_t sp_table_bind_top(sp_ctx* ctx, sp_table* const* tabs, size_t ntabs, const uint64_t r[4]);
/* Fused: bind all tables at r, then evaluate the next round on the bound tables in the same pass. */
int32_t sp_sumcheck_bind_eval(sp_ctx* ctx, int kind, sp_table* const* tabs, size_t ntabs, const uint64_t r[4], uint64_t* out_evals);
/* DensePolynomial::bound (dense_mlpoly.rs:206-213): out[i] = sum_j L[j]*Z[j*R+i], Z viewed as Lsz x (len/Lsz). */
int32_t sp_vecmat(sp_ctx* ctx, const uint64_t* L, size_t Lsz, const sp_table* Z, uint64_t* out);
/* compute_dotproduct / inner_product (nizk/mod.rs:435-438, bullet.rs:233-243) over n elements. */
int32_t sp_dot(sp_ctx* ctx, const sp_table* a, size_t a_off, const sp_table* b, size_t b_off, size_t n, uint64_t out[4]);
/* DensePolynomial::evaluate (dense_mlpoly.rs:236-242): <Z, chi(r)> with chi generated on the device. */
int32_t sp_evaluate(sp_ctx* ctx, const sp_table* Z, const uint64_t* r, size_t ell, uint64_t out[4]);
/* Read element 0 of each table (final claims after the last round: poly[0]). */
int32_t sp_table_heads(sp_ctx* ctx, sp_table* const* tabs, size_t ntabs, uint64_t* out /*4*ntabs*/);

#ifdef __cplusplus
}
#endif
#endif /* SPARTAN_HIP_H */
