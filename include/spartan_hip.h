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
/* Waits until everything queued on the context has run (calls that return results already do; calls that only enqueue —
 * sp_eq_expand, sp_gather, sp_hash_layer, sp_table_copy ... — do not). */
int32_t sp_ctx_sync(sp_ctx* ctx);
int sp_ctx_device(const sp_ctx* ctx); /* the device_id the context was created on */
/* Completed device round trips of the context so far (every wait for a result counts one): the difference around a proof is its
 * number of Fiat-Shamir steps that went to the GPU (bench.py reports it as fs_trips_per_proof). */
uint64_t sp_ctx_trips(const sp_ctx* ctx);
/* ---- options ---------------------------------------------------------------------------------------
 * Every tunable of the library is a named option with a compiled-in default (spartan_amd/csrc/options.hpp holds the table: key,
 * default, range, tier, one line of documentation; sp_option_describe enumerates it). The reference has three cargo features and no
 * environment variables (Cargo.toml:64-78); this library has an option table and ONE environment hook, SPARTAN_OPTIONS="key=value,...",
 * applied to the process-wide defaults (how A/B scripts reach the table without a recompile; an unknown or out-of-range entry aborts).
 *   sp_ctx_set_option(ctx, key, value): value is a decimal integer; SP_EINVAL for an unknown key, a value outside the option's range, or
 *     a tier-1 option (an A/B / test switch: same bytes, another placement or launch form) before "testing.unlock" = 1 was set.
 *     ctx == NULL sets the process-wide default that contexts created AFTERWARDS start from (and the options marked PROC in the table).
 *     Options that shape a generator set's tables (msm.wbits, msm.table_gb, msm.wide_gb, msm.lds_bits) are read when the set is built.
 *   sp_ctx_copy_options: a sub-context (a virtual shard) takes over its parent's settings. */
int32_t sp_ctx_set_option(sp_ctx* ctx, const char* key, const char* value);
int32_t sp_ctx_get_option(const sp_ctx* ctx, const char* key, int64_t* value);
int32_t sp_ctx_copy_options(sp_ctx* dst, const sp_ctx* src);
/* entry `index` of the table (0 .. until SP_EINVAL); any out pointer may be NULL */
int32_t sp_option_describe(int index, const char** key, int64_t* default_value, int64_t* min_value, int64_t* max_value, int* tier, const char** doc);
/* HIP-event timing of every kernel family on the context's stream (bench.py's roofline numbers). */
int32_t sp_prof_enable(sp_ctx* ctx, int on);
int32_t sp_prof_reset(sp_ctx* ctx);
/* Restrict event recording to one kernel family by name (NULL = all): each recorded launch costs two
 * hipEventRecord calls, so bench.py times with only the dominant family instrumented. */
int32_t sp_prof_select(sp_ctx* ctx, const char* family);
/* Fills up to cap entries; returns number of kernel families. name[i] is a static string. */
int32_t sp_prof_read(sp_ctx* ctx, const char** names, double* total_ms, uint64_t* launches, double* alg_bytes, int cap);

/* Algorithmic operation counts of the same families (F_q multiplications of the sum-check / streaming kernels, mixed point
 * additions of the MSM kernels assuming non-zero scalars): the numerators of the ALU roofline. */
int32_t sp_prof_read_ops(sp_ctx* ctx, double* alg_ops, int cap);
/* Per-launch-shape totals inside one family (msm_rows_fixed: shape = rows << 32 | cols, bit 63 set for launches on the
 * background stream; every other family: one pseudo-shape 0x4000000000000001 that sums its THROUGHPUT-SIZED launches, those with >= 64 MB
 * of algorithmic bytes, so that they can be reported apart from the launch-sized ones): returns the number of shapes seen, fills up to cap. */
int32_t sp_prof_read_shapes(sp_ctx* ctx, const char* family, uint64_t* shape, double* total_ms, uint64_t* launches, double* alg_bytes,
                            double* alg_ops, int cap);
/* The same launches as intervals [t0, t1) in ms on one clock (zero = the first sp_prof_enable of the context), in the order they were
 * recorded: launches of one family overlap (a background launch under a foreground one, a launch queued while its predecessor still holds
 * the CUs), so the family's busy time is the measure of the UNION of its intervals, not the sum of its durations. issued_adds: the mixed
 * additions a queue-form launch really performed (its wavefronts skip the upper windows of short scalars; 0 for the other forms). Returns
 * the number of intervals recorded since sp_prof_reset, fills up to cap. */
int32_t sp_prof_read_spans(sp_ctx* ctx, const char* family, uint64_t* shape, double* t0_ms, double* t1_ms, double* issued_adds, int cap);
/* default signed window width c (the width of a given set: sp_gens_window_bits): a committed scalar costs ceil(254 / c) mixed additions */
int sp_msm_window_bits(void);

/* ---- generators: MultiCommitGens (src/commitments.rs:8-33) ------------------------------------------
 * A sp_gens is a list of n points P[0..n). A MultiCommitGens{G[0..m), h} made by
 * MultiCommitGens::new(m, label) is the list of its m+1 stream points with h = P[m]; gens that are prefixes
 * of one SHAKE stream (gens_3/gens_4/gens_pc of R1CSGens, src/r1csproof.rs:48-73) share one sp_gens.
 * Upload builds signed c-bit fixed-base window tables (ceil(254/c) windows x 2^(c-1) affine entries, 96 B of values in a
 * 128-byte line, per point): generators are public parameters reused across proofs, so this is setup cost. c is chosen per set
 * by PROOF time, not launch time: 15 bits (17 additions per committed scalar, 35.7 MB per point) while the set's tables stay
 * under option msm.wide_gb (default 80), otherwise the widest of 14/13/12/10/8 that fits msm.table_gb (default 170 per set);
 * msm.wbits forces a width. 2^20: 15 bits for the 1025-point stream (36.6 GB), 14 for the 4098-point one (81.6 GB). With
 * msm.lds_bits = 10 the set also gets the packed 10-bit tables of the LDS-staged row MSM (1.25 MB per point: 1.3 + 5.2 GB at
 * 2^20; msm.form = 1 selects that form for commits of >= 512 rows). THE DEFAULT IS PER SET (msm.form = 0): the gathered wide-window forms
 * while the set's tables keep >= 12 bits — they win at 2^20, 2^22 and 2^24 — and the LDS-staged form, with its tables built alongside, when
 * device memory was so short that the wide tables came out at 10 or 8 bits (stderr says so). SP_ENOMEM (with a message on stderr) if not even 8-bit
 * tables fit in free device memory.
 * FIXED PUBLIC BASES ONLY: every multi-scalar multiplication of this library (sp_commit_rows*, sp_msm_indexed, the inner-product
 * argument) runs over the points of a sp_gens through its precomputed tables. There is no variable-base device MSM — the prover
 * never needs one (every base on the path is a public generator). */
/* A caller-supplied list may hold points with known relations between them (repeats, torsion shifts): the inner-product argument over such a
 * set always uses the COMPLETE addition formula in its trees; sets the library derives itself (sp_gens_from_uniform) use the faster dedicated one,
 * whose exceptional pairs cannot occur between sums over independent hash-to-curve points. */
int32_t sp_gens_upload(sp_ctx* ctx, const uint8_t* compressed /*32*n*/, size_t n, sp_gens** out);
/* MultiCommitGens::new body (commitments.rs:21-30): n blocks of 64 uniform bytes from the caller's
 * SHAKE256 stream -> from_uniform_bytes on the device. compressed_out (32*n) may be NULL. */
int32_t sp_gens_from_uniform(sp_ctx* ctx, const uint8_t* uniform /*64*n*/, size_t n, uint8_t* compressed_out, sp_gens** out);
size_t sp_gens_len(const sp_gens* g);
size_t sp_gens_table_bytes(const sp_gens* g); /* HBM held by the window tables of this set */
int sp_gens_window_bits(const sp_gens* g); /* the width c of the (narrow) windows the tables of this set were built with */
int sp_gens_windows(const sp_gens* g);     /* windows per scalar = mixed additions per committed scalar (17 .. 32; the top windows may be one bit wider than c) */
/* Window counts for two generator streams one prover holds side by side (SNARKGens: n_a / n_b points, committing w_a / w_b scalars per
 * proof), chosen together: fewest mixed additions per proof among the pairs whose tables fit the free device memory less the proof's
 * reserve. The caller sets option "msm.windows" to each result around the creation of that stream and back to 0 afterwards; 0 = no
 * recommendation (a geometry is forced, option msm.plan_pair is 0, or the streams are too small to matter). */
int32_t sp_gens_plan_pair(sp_ctx* ctx, size_t n_a, size_t n_b, double w_a, double w_b, int* windows_a, int* windows_b);
void sp_gens_free(sp_gens* g);

/* ---- Pedersen commitments (src/commitments.rs:73-92, src/dense_mlpoly.rs:164-177, src/group.rs:98-117) --
 * out[i] = compress( sum_j Z[i*cols+j] * P[g_off+j]  +  blinds[i] * P[h_idx] ),  i < rows.
 * Replaces DensePolynomial::commit_inner (rows = L) and [Scalar]::commit (rows = 1). blinds may be NULL (=0). */
int32_t sp_commit_rows(sp_ctx* ctx, const sp_gens* g, size_t g_off, size_t h_idx, const uint64_t* Z, size_t rows, size_t cols,
                       const uint64_t* blinds, uint8_t* out /*32*rows*/);
/* Same with Z taken from a device table: elements [z_off, z_off + rows*cols). */
int32_t sp_commit_rows_dev(sp_ctx* ctx, const sp_gens* g, size_t g_off, size_t h_idx, const sp_table* Z, size_t z_off, size_t rows,
                           size_t cols, const uint64_t* blinds, uint8_t* out);
/* Background variant (no blinds): the commit is queued on a lower-priority HIP stream behind everything issued so far and
 * runs concurrently with later calls on the context (persistent workgroups on bg.eighths/8 of the CUs, 5/8 by default, so the latency-bound kernels
 * of those calls keep idle CUs to run on); sp_job_wait blocks, copies the 32*rows bytes out and frees the job.
 * Z and g must stay alive and Z[z_off, z_off + rows*cols) unmodified until sp_job_wait returns.
 * Used to overlap the row half of the SPARK `derefs` commitment (sparse_mlpoly.rs:1473-1478), which only depends on rx,
 * with the latency-bound second sum-check of R1CSProof::prove. */
typedef struct sp_job sp_job;
int32_t sp_commit_rows_dev_begin(sp_ctx* ctx, const sp_gens* g, size_t g_off, const sp_table* Z, size_t z_off, size_t rows, size_t cols,
                                 sp_job** out);
/* Foreground variant (rows > 8, blinds allowed): queued on the context's main stream at full width; the caller may do HOST
 * work, and may sp_job_wait OTHER jobs of the context (a background one that finishes earlier), but makes no other call on
 * this context until it has waited for this job. */
int32_t sp_commit_rows_dev_start(sp_ctx* ctx, const sp_gens* g, size_t g_off, size_t h_idx, const sp_table* Z, size_t z_off, size_t rows,
                                 size_t cols, const uint64_t* blinds, sp_job** out);
/* The same commitment when the rows are still in host memory (the assignment SNARK::prove is handed, src/lib.rs:339-344): the rows
 * are copied into Z[z_off, z_off + rows*cols) in chunks and each chunk's additions are launched behind its copy, so the MSM overlaps
 * the PCIe transfer; one reduction and encode at the end. Collected with sp_job_wait; src must stay valid and unmodified until
 * sp_job_wait has returned (the runtime may still be reading it when this call returns). */
int32_t sp_commit_rows_upload_start(sp_ctx* ctx, const sp_gens* g, size_t g_off, size_t h_idx, sp_table* Z, size_t z_off, const uint64_t* src, size_t rows,
                                    size_t cols, const uint64_t* blinds, sp_job** out);
int32_t sp_job_wait(sp_job* job, uint8_t* out /*32*rows*/);
/* Small/irregular commits (Scalar::commit, UniPoly::commit, the Sigma-protocol commitments of
 * src/nizk/mod.rs, the per-round L/R of src/nizk/bullet.rs:83-97 re-expressed over the ORIGINAL generators):
 * out[i] = compress( sum_j S[i*cols+j] * P[idx[j]] ). */
int32_t sp_msm_indexed(sp_ctx* ctx, const sp_gens* g, const uint32_t* idx, size_t cols, const uint64_t* S, size_t rows, uint8_t* out);

/* ---- few-term commitments on the CALLING THREAD's core (Scalar::commit / UniPoly::commit of the Sigma protocols,
 * src/commitments.rs:73-93; src/nizk/mod.rs; the rounds of the zero-knowledge sum-checks, src/sumcheck.rs:471-583, 661-772).
 * A 2..5-term commitment is a chain of ~100 dependent point additions and one inverse square root that the transcript waits
 * for: ~15 us on the core that is waiting anyway, ~60 us + a round trip on a lone wavefront. These entry points compute
 * them on the host from signed 10-bit window tables of the generators involved (built lazily, 1.25 MiB per generator,
 * from the encodings the sp_gens was created from) with the same point arithmetic the kernels compile. No GPU work, no
 * context: any thread may call them. sp_msm_indexed is the device form of the same commitments (same bytes).
 *   out[r] = compress( sum_k S[r*cols+k] * P[idx[k]]  (+ *addend[r] when addend and addend[r] are non-NULL) ),  cols <= 16. */
typedef struct sp_host_point { uint64_t w[16]; } sp_host_point; /* an extended point in the library's own form; opaque to callers */
int32_t sp_host_commit_small(const sp_gens* g, const uint32_t* idx, size_t cols, const uint64_t* S, size_t rows, const sp_host_point* const* addend,
                             uint8_t* out /*32*rows*/);
/* One row left as a point (not encoded): a partial commitment to be added to a later one through `addend`. */
int32_t sp_host_commit_point(const sp_gens* g, const uint32_t* idx, size_t cols, const uint64_t* S, sp_host_point* out);
/* Column-sharded commitments (DensePolynomial::commit_inner, src/dense_mlpoly.rs:164-177, when it has fewer rows than
 * there are shards): sp_commit_rows_partial leaves sum_j Z[r*z_stride + j] * G[g_off + j], j < cols, of rows <= 8 as points (device);
 * sp_host_points_sum_encode adds the nsets points of every row (pts[set*rows + r]: one set per shard, one for the blind terms from
 * sp_host_commit_point) and encodes the sums (host). */
int32_t sp_commit_rows_partial(sp_ctx* ctx, const sp_gens* g, size_t g_off, const sp_table* Z, size_t z_off, size_t z_stride, size_t rows, size_t cols,
                               sp_host_point* out /*rows*/);
int32_t sp_host_points_sum_encode(const sp_host_point* pts, size_t nsets, size_t rows, uint8_t* out /*32*rows*/);
/* The same arithmetic without a device or an sp_gens (CPU tests): npts encoded points, rows x npts scalars. */
int32_t sp_host_commit_probe(const uint8_t* compressed /*32*npts*/, size_t npts, const uint64_t* S, size_t rows, uint8_t* out);
/* Device-free form, for tests, of the one variable-base multiplication of the prover: k1 P1 + k2 P2 over two compressed points, encoded — the
 * arithmetic with which sp_ipa_finish_commit ends an inner-product argument on the calling thread's core (delta = d g_hat + r_delta h of
 * nizk/mod.rs:496-501 from the last round's two row sums, bullet.rs:108). SP_EPOINT for an invalid encoding. */
int32_t sp_host_msm2_probe(const uint8_t p1[32], const uint64_t k1[4], const uint8_t p2[32], const uint64_t k2[4], uint8_t out[32]);
/* Look-ahead for the zero-knowledge sum-checks. Inside their round loop nothing but DotProductProof::prove draws from the
 * random tape (d_vec, r_delta, r_beta: nizk/mod.rs:330-334), so a caller can take the draws of all rounds up front, in the
 * reference's order, and have a helper thread compute everything that depends on the tape alone while the rounds run:
 * with idx_u = (gens_n.G[0..nn), gens_n.h, gens_1.G[0], gens_1.h) (W = nn + 3 entries), for round j
 *   delta_j = compress( <d_j, gens_n.G> + r_delta_j * gens_n.h )            the DotProductProof's delta, complete
 *   bp_hn_j = blinds_poly[j] * gens_n.h,  be_h_j = blinds_evals[j] * gens_1.h,  rb_h_j = r_beta_j * gens_1.h
 * the blind terms of comm_poly, comm_eval and beta, to be passed as `addend` to sp_host_commit_small.
 * begin copies its inputs and starts the thread; wait blocks until round j is done (rounds complete in order);
 * free joins the thread. d is [rounds][nn]; the other vectors [rounds]. */
typedef struct sp_zk_ahead sp_zk_ahead;
int32_t sp_host_zk_ahead_begin(const sp_gens* g, const uint32_t* idx_u, size_t W, size_t nn, size_t rounds, const uint64_t* blinds_poly,
                               const uint64_t* blinds_evals, const uint64_t* d, const uint64_t* r_delta, const uint64_t* r_beta, sp_zk_ahead** out);
int32_t sp_host_zk_ahead_wait(sp_zk_ahead* a, size_t j, uint8_t delta[32], sp_host_point* be_h, sp_host_point* rb_h, sp_host_point* bp_hn);
void sp_host_zk_ahead_free(sp_zk_ahead* a);

/* ---- device tables (DensePolynomial.Z, src/dense_mlpoly.rs:14-18) ----------------------------------- */
int32_t sp_table_alloc(sp_ctx* ctx, size_t len, sp_table** out);          /* zero-filled */
int32_t sp_table_alloc_uninit(sp_ctx* ctx, size_t len, sp_table** out);   /* contents undefined: for tables the caller overwrites entirely */
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
/* The last round of a sum-check: tables of length 2 are bound at r (length 1) and their remaining entry is returned,
 * out_heads[4k..4k+4) for table k — one launch and one wait for any number of tables (each listed once). */
int32_t sp_table_bind_top_heads(sp_ctx* ctx, sp_table* const* tabs, size_t ntabs, const uint64_t r[4], uint64_t* out_heads);
/* Fused: bind all tables at r, then evaluate the next round on the bound tables in the same pass. */
int32_t sp_sumcheck_bind_eval(sp_ctx* ctx, int kind, sp_table* const* tabs, size_t ntabs, const uint64_t r[4], uint64_t* out_evals);
/* sp_sumcheck_bind_eval in two halves: _start queues the bind and the evaluation and returns at once, _collect waits for the
 * sums. The host driver computes the round's few-term commitments on its own core in between (the device needs ~25 us for a
 * small round, the commitments about as long). No other call on ctx may come between the two; SP_EINVAL if one is pending
 * (_start) or none is (_collect). */
int32_t sp_sumcheck_bind_eval_start(sp_ctx* ctx, int kind, sp_table* const* tabs, size_t ntabs, const uint64_t r[4]);
int32_t sp_sumcheck_bind_eval_collect(sp_ctx* ctx, uint64_t* out_evals);
/* One round body of the zero-knowledge sum-checks (sumcheck.rs:471-583 / 661-772): sp_sumcheck_bind_eval at r and, at the
 * same time on a second stream, the `rows` <= 8 small commitments of the round whose scalars are known as soon as r is
 * (comm_eval and the DotProductProof's delta): out_points[k] = compress( sum_j S[k*cols+j] * P[idx[j]] ), cols <= 11.
 * One completion wait instead of two, and the two kernels overlap. */
int32_t sp_sumcheck_bind_eval_commit(sp_ctx* ctx, int kind, sp_table* const* tabs, size_t ntabs, const uint64_t r[4], uint64_t* out_evals,
                                     const sp_gens* g, const uint32_t* idx, size_t cols, const uint64_t* S, size_t rows, uint8_t* out_points);
/* DensePolynomial::bound (dense_mlpoly.rs:206-213): out[i] = sum_j L[j]*Z[j*R+i], Z viewed as Lsz x (len/Lsz). */
int32_t sp_vecmat(sp_ctx* ctx, const uint64_t* L, size_t Lsz, const sp_table* Z, uint64_t* out);
/* The same with the result left on the device as a new table (PolyEvalProof::prove only commits to it and feeds it to the
 * inner-product argument: it never has to visit the host). L is copied before the call returns; the multiplication is queued,
 * not waited for — the table is ready for every later call on ctx (stream order). */
int32_t sp_vecmat_dev(sp_ctx* ctx, const uint64_t* L, size_t Lsz, const sp_table* Z, sp_table** out);
/* The same with L a device table (e.g. EqPolynomial::evals of the left half from sp_eq_expand): nothing of L crosses PCIe and the host
 * does not compute it. Queued, not waited for; L may be freed right after the call. */
int32_t sp_vecmat_tab(sp_ctx* ctx, const sp_table* L, const sp_table* Z, sp_table** out);
/* compute_dotproduct / inner_product (nizk/mod.rs:435-438, bullet.rs:233-243) over n elements. */
int32_t sp_dot(sp_ctx* ctx, const sp_table* a, size_t a_off, const sp_table* b, size_t b_off, size_t n, uint64_t out[4]);
/* DensePolynomial::evaluate (dense_mlpoly.rs:236-242): <Z, chi(r)> with chi generated on the device. */
int32_t sp_evaluate(sp_ctx* ctx, const sp_table* Z, const uint64_t* r, size_t ell, uint64_t out[4]);
/* Read element 0 of each table (final claims after the last round: poly[0]). */
int32_t sp_table_heads(sp_ctx* ctx, sp_table* const* tabs, size_t ntabs, uint64_t* out /*4*ntabs*/);
/* Read `count` consecutive elements starting at offs[k] from each table in one round trip (ProductCircuit::evaluate,
 * product_tree.rs:58-63, reads the two roots of every circuit): out[(k*count + e)*4 ..]. */
int32_t sp_table_gather(sp_ctx* ctx, sp_table* const* tabs, const size_t* offs, size_t ntabs, size_t count, uint64_t* out);

/* ---- sharding helpers: a sum-check table split by INDEX RESIDUE keeps every top-variable pair (i, i + len/2) on one
 * shard while len/2 is a multiple of W, so a shard runs the ordinary round kernels on its sub-table and only the 2..3 partial sums of
 * a round (<= 96 bytes) are exchanged; DensePolynomial::bound shards by row blocks and its partial vectors are added.
 *   sp_table_residue_split: out[k] = src[k W + g], k < len(src) / W, as a new table of `ctx` (src may belong to another context of the
 *     same GPU: the virtual-shard test transport; the caller orders the two contexts' streams with sp_ctx_sync).
 *   sp_table_set_len: the current (bound) length of a table, e.g. W after the W surviving entries of the shards were written back.
 *   sp_table_add_into: dst[k] += src[k] (F_q), equal current lengths. */
int32_t sp_table_residue_split(sp_ctx* ctx, const sp_table* src, size_t W, size_t g, sp_table** out);
/* Hand-over of residue-sharded tables (the batched cubic sum-checks of SPARK, src/sumcheck.rs:254-424, sharded like the ZK ones): once the
 * tables are short enough that a round costs less than the exchange, every shard packs the first `count` entries of its sub-tables —
 *   sp_tables_pack: out[(t * count + k) * 4 ..] = tabs[t][k], any size (one DMA) —
 * the buffers are gathered in shard order, and the owner scatters them back into its full tables —
 *   sp_tables_unpack_residues: tabs[t][k * W + g] = in[((g * ntabs + t) * sub + k) * 4 ..]; every table's current length becomes W * sub
 *   (<= its capacity; the tables must be distinct). */
int32_t sp_tables_pack(sp_ctx* ctx, sp_table* const* tabs, size_t ntabs, size_t count, uint64_t* out);
int32_t sp_tables_unpack_residues(sp_ctx* ctx, sp_table* const* tabs, size_t ntabs, size_t W, size_t sub, const uint64_t* in);
int32_t sp_table_set_len(sp_table* t, size_t len);
int32_t sp_table_add_into(sp_ctx* ctx, sp_table* dst, const sp_table* src);

/* ---- sparse matrices: SparseMatPolynomial (src/sparse_mlpoly.rs:19-38, 429-481) ----------------------
 * Entries (row, col, val). Upload keeps a row-sorted (CSR) and a column-sorted (CSC) copy on the device so
 * both products are gather-only (F_q has no atomic add). */
typedef struct sp_sparse sp_sparse;
int32_t sp_sparse_upload(sp_ctx* ctx, const uint64_t* rows, const uint64_t* cols, const uint64_t* vals /*4*nnz*/, size_t nnz,
                         size_t num_rows, size_t num_cols, sp_sparse** out);
void sp_sparse_free(sp_sparse* m);
/* multiply_vec (sparse_mlpoly.rs:454-464): out[row] = sum val * z[col]; out has num_rows elements. */
int32_t sp_sparse_mulvec(sp_ctx* ctx, const sp_sparse* m, const sp_table* z, sp_table** out);
/* compute_eval_table_sparse (sparse_mlpoly.rs:466-481) for nm matrices, combined as r1csproof.rs:275-283 does:
 * out[col] = sum_k w[k] * sum_{(row,col,val) in M_k} rx[row] * val ; out has num_cols elements. */
int32_t sp_sparse_eval_table(sp_ctx* ctx, const sp_sparse* const* ms, const uint64_t* w /*4*nm*/, size_t nm, const sp_table* rx,
                             sp_table** out);
/* evaluate_with_tables (sparse_mlpoly.rs:429-438): sum tx[row] * ty[col] * val. */
int32_t sp_sparse_evaluate(sp_ctx* ctx, const sp_sparse* m, const sp_table* tx, const sp_table* ty, uint64_t out[4]);
/* The same for up to four matrices at once, queued on a low-priority stream and collected with sp_job_wait (32 bytes per matrix, in
 * order): SNARK::prove (src/lib.rs:393-400) starts R1CSInstance::evaluate as soon as ry is known and collects the three values after
 * the witness opening. tx and ty must outlive the job. */
int32_t sp_sparse_evaluate_begin(sp_ctx* ctx, const sp_sparse* const* ms, size_t count, const sp_table* tx, const sp_table* ty, sp_job** out);

/* ---- inner-product argument: BulletReductionProof::prove (src/nizk/bullet.rs:32-132) -----------------
 * The folded generators G^(k) are never materialised: G^(k)[i] = sum_p s_k[p] * G[p*n_k + i] with
 * s_{k+1}[2p] = s_k[p]*u^-1, s_{k+1}[2p+1] = s_k[p]*u, so every round's L and R are fixed-base MSMs over the
 * ORIGINAL generators P[g_off .. g_off+n) with scalar vectors s (x) a — same group elements, same bytes.
 * Q = q_scale * P[q_idx] (DotProductProofLog scales gens_1 by r, src/nizk/mod.rs:479-480), H = P[h_idx]. */
typedef struct sp_ipa sp_ipa;
int32_t sp_ipa_begin(sp_ctx* ctx, const sp_gens* g, size_t g_off, size_t n, size_t q_idx, size_t h_idx, const uint64_t q_scale[4],
                     const uint64_t* a /*4*n*/, const uint64_t* b /*4*n*/, sp_ipa** out);
/* DotProductProofLog::prove (nizk/mod.rs:440-480) with x already on the device: a = the first n entries of a_dev, and
 * commit_a = compress(<a, G> + blind_a * H) (its Cx) is computed from the same device copy. The scale of Q (= the
 * challenge r drawn AFTER Cx is absorbed) is set afterwards with sp_ipa_set_scale, before the first round. */
int32_t sp_ipa_begin_dev(sp_ctx* ctx, const sp_gens* g, size_t g_off, size_t n, size_t q_idx, size_t h_idx, const sp_table* a_dev, const uint64_t* b,
                         const uint64_t blind_a[4], uint8_t commit_a[32], sp_ipa** out);
int32_t sp_ipa_set_scale(sp_ipa* ipa, const uint64_t q_scale[4]);
/* bullet.rs:72-100: c_L, c_R, L = <a_L,G_R> + c_L Q + blind_L H, R = <a_R,G_L> + c_R Q + blind_R H, compressed. */
/* Puts the kernel of the next sp_ipa_round_lr in flight now: it depends on neither that round's blinds nor the scale of Q, so
 * DotProductProofLog::prove (src/nizk/mod.rs:469-480) launches the first round before it absorbs Cx, Cy and `a` and draws r. No other
 * call on the context until sp_ipa_round_lr. A no-op when the round does not take the one-launch path. */
int32_t sp_ipa_round_prelaunch(sp_ipa* ipa);
int32_t sp_ipa_round_lr(sp_ipa* ipa, const uint64_t blind_L[4], const uint64_t blind_R[4], uint8_t L_out[32], uint8_t R_out[32]);
/* bullet.rs:105-109: fold a, b (and the generator coefficients s) with the round challenge. */
int32_t sp_ipa_round_fold(sp_ipa* ipa, const uint64_t u[4], const uint64_t u_inv[4]);
/* bullet.rs:121-131: a_hat = a[0], b_hat = b[0], g_hat = G^(last)[0] (compressed; may be NULL). */
int32_t sp_ipa_finish(sp_ipa* ipa, uint64_t a_hat[4], uint64_t b_hat[4], uint8_t* g_hat);
/* commit(d, r) under {G: g_hat, h: H}: d * g_hat + r * H, compressed (nizk/mod.rs:496-501 `delta`). */
int32_t sp_ipa_commit_ghat(sp_ipa* ipa, const uint64_t d[4], const uint64_t r[4], uint8_t out[32]);
/* sp_ipa_finish (without g_hat) and sp_ipa_commit_ghat in one round trip: the commitment under g_hat does not depend on
 * a_hat, b_hat (nizk/mod.rs:498-503 forms y_hat = a_hat * b_hat afterwards). */
int32_t sp_ipa_finish_commit(sp_ipa* ipa, const uint64_t d[4], const uint64_t r[4], uint64_t a_hat[4], uint64_t b_hat[4], uint8_t delta_out[32]);
void sp_ipa_free(sp_ipa* ipa);

/* ---- SPARK: sparse-polynomial evaluation proof building blocks (src/sparse_mlpoly.rs, src/product_tree.rs) ---- */
typedef struct sp_index sp_index; /* device-resident Vec<usize> (AddrTimestamps::ops_addr_usize, sparse_mlpoly.rs:213-219) */
int32_t sp_index_upload(sp_ctx* ctx, const uint64_t* idx, size_t n, sp_index** out);
void sp_index_free(sp_index* ix);
/* SNARK::encode without a host pass over the matrices (lib.rs:325-336 -> sparse_mlpoly.rs:367-427): sp_sparse keeps its entries in the order
 * they were given (SparseMatPolynomial.M);
 *   sp_sparse_entry_index: their row (which = 0) or column (which = 1) addresses as a device index list of n >= nnz elements, zero-padded
 *     (MultiSparseMatPolynomialAsDense pads every matrix to the batch's num_nz_entries) — free with sp_index_free;
 *   sp_sparse_entry_values: their values written into dst[dst_off, dst_off + n), zero-padded. Device to device, queued on the context's stream. */
int32_t sp_sparse_entry_index(sp_ctx* ctx, const sp_sparse* m, int which, size_t n, sp_index** out);
int32_t sp_sparse_entry_values(sp_ctx* ctx, const sp_sparse* m, sp_table* dst, size_t dst_off, size_t n);
/* AddrTimestamps::new (sparse_mlpoly.rs:221-254) for `nlists` address lists of equal length walked one after the other
 * over ONE array of `cells` counters: read_ts of list k goes to ts_dst[ts_off[k] ..] and the final counters (audit_ts)
 * to audit_dst[audit_off .. audit_off + cells), both as F_q tables. Addresses must be < cells. The sequential scan of
 * the reference is computed as a stable sort by address plus rank-in-run. */
int32_t sp_addr_timestamps(sp_ctx* ctx, sp_index* const* addr, size_t nlists, size_t cells, sp_table* ts_dst, const size_t* ts_off,
                           sp_table* audit_dst, size_t audit_off);
/* DensePolynomial::from_usize (dense_mlpoly.rs:274-280) written into dst[dst_off ..]. */
int32_t sp_table_from_index(sp_ctx* ctx, const sp_index* ix, sp_table* dst, size_t dst_off);
/* Non-owning view of elements [off, off+len) of a table (DensePolynomial::split, dense_mlpoly.rs:140-146). The
 * parent must outlive the view; free the view with sp_table_free. */
int32_t sp_table_view(sp_ctx* ctx, const sp_table* parent, size_t off, size_t len, sp_table** out);
/* AddrTimestamps::deref_mem (sparse_mlpoly.rs:256-265): dst[dst_off + i] = mem[addr[i]]. */
int32_t sp_gather(sp_ctx* ctx, const sp_table* mem, const sp_index* addr, sp_table* dst, size_t dst_off);
/* Layers::build_hash_layer (sparse_mlpoly.rs:529-604):
 *   dst[dst_off+i] = (ts[i] + ts_inc) * r_hash^2 + val[i] * r_hash + addr[i] - r_multiset,  i < n
 * addr == NULL means the identity (addr[i] = i); ts == NULL means 0. */
int32_t sp_hash_layer(sp_ctx* ctx, const sp_table* addr, const sp_table* val, const sp_table* ts, int ts_inc, size_t n,
                      const uint64_t r_hash[4], const uint64_t r_multiset[4], sp_table* dst, size_t dst_off);
/* The same leaves AND the first multiplication layer of ProductCircuit::new (product_tree.rs:36-56: layer 1 [i] = leaf[i] * leaf[i + n/2], at
 * offset n of a 2n-element circuit store) in one pass, n >= 4: the first layer does not re-read the leaves. dst_write != NULL (ts_inc must be
 * 0): also the circuit of the WRITE set of the same matrix — its leaves are those of the read set with ts + 1 (sparse_mlpoly.rs:572-598), i.e.
 * plus r_hash^2 — from the same pass over addr, val, ts. The remaining layers: sp_product_tree_many_from(.., layers_done = 1). */
int32_t sp_hash_layer_first(sp_ctx* ctx, const sp_table* addr, const sp_table* val, const sp_table* ts, int ts_inc, size_t n,
                            const uint64_t r_hash[4], const uint64_t r_multiset[4], sp_table* dst, sp_table* dst_write);
/* ProductCircuit::new (product_tree.rs:36-56). `store` has 2n elements with the n leaves in [0,n); layer k
 * (n/2^k elements, left half then right half) is written at offset 2n - 2n/2^k for k = 1..log2(n)-1. */
int32_t sp_product_tree(sp_ctx* ctx, sp_table* store, size_t n);
/* The same for `count` circuits of equal size n, one launch per layer for all of them. */
int32_t sp_product_tree_many(sp_ctx* ctx, sp_table* const* stores, size_t count, size_t n);
/* The same, starting above the layers already present: layers_done = 0 (all of them) or 1 (layer 1 was written by sp_hash_layer_first). */
int32_t sp_product_tree_many_from(sp_ctx* ctx, sp_table* const* stores, size_t count, size_t n, size_t layers_done);
/* prove_cubic_batched evaluations (sumcheck.rs:287-357): for each instance k, the sums of A_k*B_k*C_k at
 * t = 0, 2, 3 over the current length -> out[4*(3k + {0,1,2})]. Tables may repeat across instances. */
int32_t sp_sumcheck_eval_batched(sp_ctx* ctx, sp_table* const* A, sp_table* const* B, sp_table* const* C, size_t ninst, uint64_t* out);
/* Fused round: bind every A_k, B_k and C_k at r (halving them) and evaluate the next round on the bound values in the
 * same pass. A and B tables must be distinct; a C table may be shared by several instances (poly_C_par) and is bound
 * once, out of place. Requires current length >= 4. */
int32_t sp_sumcheck_bind_eval_batched(sp_ctx* ctx, sp_table* const* A, sp_table* const* B, sp_table* const* C, size_t ninst,
                                      const uint64_t r[4], uint64_t* out);
/* The same two calls with the eq table as a FACTOR, for the throughput-sized rounds (current length >= 65536) of the batches
 * ProductCircuitEvalProofBatched::prove builds (product_tree.rs:259-383): instances [0, neq) are product-circuit instances whose third
 * table is the shared poly_C_par = EqPolynomial::new(rand).evals() (product_tree.rs:279), C[0] == ... == C[neq-1]; instances [neq, ninst)
 * are generic (the dot-product circuits). After binds at r_1..r_{j-1} the bound eq table is a scalar the caller knows times eq(t, rand_j)
 * times the LEADING entries of the original table, so the device never binds it and returns, per product-circuit instance,
 *     q(t) = sum_x A(t,x) B(t,x) C_original[x]   at t = 0 and t = 2   (q is quadratic in t: q(1) follows from the round's claim, q(3) by
 *     extrapolation; the caller multiplies by kappa_j(t) = [prod_{k<j} eq(r_k, rand_k) / (1 - rand_k)] * eq(t, rand_j) / (1 - rand_j)),
 * 8 multiplications per index and instance instead of 12 (4 instead of 6 without a bind). Generic instances return e(t) at t = 0, 1, 2, 3.
 * out[16 * ninst]: 4 scalars per instance, {q(0), q(2), 0, 0} or {e(0), e(1), e(2), e(3)}. The shared eq table is read, never written,
 * and keeps its length; A, B (and the generic C tables) are bound in place as by sp_sumcheck_bind_eval_batched.
 * sp_table_scale_prefix: table[i] *= k for i < n and n becomes its length — the hand-over to the generic rounds (k = the scalar above:
 * the table is then the bound eq table the generic calls expect). Queued, not waited for. */
int32_t sp_sumcheck_eval_batched_eq(sp_ctx* ctx, sp_table* const* A, sp_table* const* B, sp_table* const* C, size_t ninst, size_t neq, uint64_t* out);
int32_t sp_sumcheck_bind_eval_batched_eq(sp_ctx* ctx, sp_table* const* A, sp_table* const* B, sp_table* const* C, size_t ninst, size_t neq,
                                         const uint64_t r[4], uint64_t* out);
int32_t sp_table_scale_prefix(sp_ctx* ctx, sp_table* table, size_t n, const uint64_t k[4]);
/* TWO rounds per call, for the latency-bound rounds of prove_cubic_batched (a round trip to the device costs more than the
 * arithmetic of a short round). The evaluations of the round after a bind are a cubic in that bind's challenge r:
 *   E(t; r) = (1-r)^3 M0(t) + (1-r)^2 r M1(t) + (1-r) r^2 M2(t) + r^3 M3(t),   M1 = (T1 - T2)/2 - M3,  M2 = (T1 + T2)/2 - M0,
 * so a caller that holds (M0, M3, T1, T2) for t = 0, 2, 3 can derive the next challenge, evaluate the round after it and
 * derive that challenge too before it talks to the device again.
 *   sp_sumcheck_eval_coeffs_batched: the tables as they are (current length n >= 2): out_evals[12*ninst] as
 *     sp_sumcheck_eval_batched, and, when n >= 4, out_coeffs[48*ninst] = per instance, for t = 0, 2, 3: M0, M3, T1, T2 of the
 *     round that follows a bind of these tables.
 *   sp_sumcheck_bind2_eval_batched: binds every table at r0 and, when r1 != NULL, then at r1 (length L becomes L/2 or L/4; a
 *     shared C table is bound once, out of place), and returns for the tables so bound: out_evals (new length >= 2),
 *     out_coeffs (new length >= 4), or, when the new length is 1, out_heads = A_0, B_0, A_1, B_1, ..., then each distinct C
 *     table in order of first appearance (the final claims, as sp_table_bind_top_heads). Unused outputs may be NULL.
 *   weights != NULL (4*ninst limbs: the `coeffs` of sumcheck.rs:359-369): the evaluations and coefficients are multiplied by
 *     their instance's weight on the device and summed over the instances — out_evals[12] and out_coeffs[48] in all, which is
 *     all the protocol uses.
 * While the bound tables still have >= 4 entries (and were not handed over, see below) the next call of this family on the
 * same tables is known to be a two-bind call: with option sumcheck.launch_ahead = 1 (not the default) its kernel is enqueued
 * before this call returns and waits for r0, r1 in host memory (bounded wait). That call then only hands the challenges
 * over; any other use of the context makes the waiting kernel give up first. Nothing changes in what the calls return. */
int32_t sp_sumcheck_eval_coeffs_batched(sp_ctx* ctx, sp_table* const* A, sp_table* const* B, sp_table* const* C, size_t ninst, const uint64_t* weights,
                                        uint64_t* out_evals, uint64_t* out_coeffs);
int32_t sp_sumcheck_bind2_eval_batched(sp_ctx* ctx, sp_table* const* A, sp_table* const* B, sp_table* const* C, size_t ninst, const uint64_t r0[4],
                                       const uint64_t* r1, const uint64_t* weights, uint64_t* out_evals, uint64_t* out_coeffs, uint64_t* out_heads);
/* The same call (r0 == NULL: no bind, as sp_sumcheck_eval_coeffs_batched; r1 == NULL: one bind) that ALSO hands over the tables
 * once they are short: when the tables the outputs describe have 2, 4 or 8 entries (and ninst <= 21), out_tables receives them,
 * [ninst][3: A, B, C][n] Montgomery limbs, and the caller can finish the last <= 3 rounds of the sum-check (sumcheck.rs:287-419:
 * evaluations, binds, final claims) on its own core — ~20 us of field arithmetic against two more round trips. Otherwise
 * out_tables[0] is set to all ones and nothing else is written there. The device tables are left at that length. */
int32_t sp_sumcheck_bind2_eval_tables_batched(sp_ctx* ctx, sp_table* const* A, sp_table* const* B, sp_table* const* C, size_t ninst, const uint64_t* r0,
                                              const uint64_t* r1, const uint64_t* weights, uint64_t* out_evals, uint64_t* out_coeffs,
                                              uint64_t* out_heads, uint64_t* out_tables /* 4*ninst*3*8 */);
/* out[k] = <chi, T_k> for k < nt (the ~23 DensePolynomial::evaluate calls of HashLayerProof::prove share chi). */
int32_t sp_dot_many(sp_ctx* ctx, const sp_table* chi, sp_table* const* tabs, size_t nt, uint64_t* out /*4*nt*/);
/* DotProductCircuit::evaluate (product_tree.rs:84-88): sum l[i]*r[i]*w[i] over n elements from the given offsets. */
int32_t sp_dot3(sp_ctx* ctx, const sp_table* l, const sp_table* r, const sp_table* w, size_t off, size_t n, uint64_t out[4]);
/* nt of them in one launch and one wait (the six claim_eval_dotp_left/right of ProductLayerProof::prove, sparse_mlpoly.rs:1084-1101):
 * out[4k..4k+4) = sum_{i<n} l_k[i] r_k[i] w_k[i]. */
int32_t sp_dot3_many(sp_ctx* ctx, const sp_table* const* l, const sp_table* const* r, const sp_table* const* w, size_t nt, size_t n, uint64_t* out);

#ifdef __cplusplus
}
#endif
#endif /* SPARTAN_HIP_H */
