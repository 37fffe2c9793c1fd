// spartan_amd: the library's options. Every tunable is a NAMED OPTION of a context with a compiled-in default — set with
// sp_ctx_set_option(ctx, "key", "value") (include/spartan_hip.h), never by an environment variable read somewhere inside a kernel
// launcher. (The reference has three cargo features and no environment variables: Cargo.toml:64-78.)
//
//   tier 0  deployment tuning: memory budgets, the background share of the CUs, sharding thresholds, the MSM form.
//   tier 1  A/B and test switches: alternative placements and launch forms that produce the SAME BYTES (every one is exercised by
//           tests/test_gpu_proofs.py::test_every_ab_switch_gives_the_same_proof). Refused (SP_EINVAL) until the caller has set
//           "testing.unlock" = 1 on the same context (or process-wide): a production caller cannot trip into them.
//
// sp_ctx_set_option(NULL, ...) changes the process-wide defaults that contexts created afterwards start from (and the few options that
// are process-wide by nature, marked PROC). The ONE environment hook is SPARTAN_OPTIONS="key=value,key=value", applied to the
// process-wide defaults when the library first needs them: it is how A/B scripts reach the same table without a recompile.
#pragma once
#include <cstdint>

// X(id, key, default, min, max, tier, doc)
#define SP_OPTION_TABLE(X)                                                                                                                  \
  X(TESTING_UNLOCK, "testing.unlock", 0, 0, 1, 0, "1: accept tier-1 (A/B / test) options on this context")                                   \
  X(MSM_FORM, "msm.form", 0, 0, 3, 0, "row MSM of many rows: 0 = per launch and generator set (the queue form over the wide-window tables for commits of >= 256 rows; LDS-staged when the set's wide tables came out <= 10 bits; strip / balanced forms for the rest), 1 = LDS-staged small windows (needs msm.lds_bits at set creation), 2 = queue form also for sets that would take the LDS-staged form, 3 = strip / balanced forms always (the default before round 6)") \
  X(MSM_LDS_BITS, "msm.lds_bits", 0, 0, 10, 0, "window width of the LDS-staged form's tables built with a generator set (0 = not built; 10 = 48 KB sub-tables, double-buffered)") \
  X(MSM_WBITS, "msm.wbits", 0, 0, 15, 0, "force UNIFORM windows of this width (4..15: ceil(254 / width) additions per scalar); 0 = chosen by the policy below") \
  X(MSM_WINDOWS, "msm.windows", 0, 0, 32, 0, "force this many windows = additions per committed scalar (17..32; mixed widths, as narrow as 254 bits allow); 0 = the fewest that fit the budgets below") \
  X(MSM_PLAN_PAIR, "msm.plan_pair", 1, 0, 1, 0, "SNARKGens: the window counts of the two generator streams are chosen together (sp_gens_plan_pair: fewest additions per proof within free memory); 0 = each set by the per-set policy below") \
  X(MSM_TABLE_GB, "msm.table_gb", 180, 1, 100000, 0, "HBM budget of one generator set's wide tables, GB")                                    \
  X(MSM_WIDE_GB, "msm.wide_gb", 80, 1, 100000, 0, "17 windows only while the set's tables stay under this many GB")                          \
  X(BG_EIGHTHS, "bg.eighths", 5, 0, 8, 0, "share of the CUs (in eighths) the background half of the derefs commitment runs on; 0 = plain low-priority launches") \
  X(UPLOAD_CHUNKS, "upload.chunks", 4, 1, 16, 0, "row chunks the witness upload + commit is issued in (each chunk's additions behind its PCIe copy)") \
  X(SHARD_COLS, "shard.cols", 1, 0, 1, 0, "column-sharded commitments for commits with fewer rows than shards")                              \
  X(SHARD_RESIDUES, "shard.residues", 1, 0, 1, 0, "index-residue sharding of sum-check tables, bound and evaluate")                          \
  X(SHARD_RESIDUE_MIN_LOG2, "shard.residue_min_log2", 22, 0, 64, 0, "over multi-process transports: shard a sum-check when its tables have >= 2^this entries (0 = always, 64 = never)") \
  X(HOST_KECCAK, "host.keccak", 0, 0, 3, 0, "PROC. Keccak-f[1600] form of the host transcript: 0 = fastest by calibration, 1 = plain, 2 = BMI2, 3 = AVX-512") \
  X(HOST_PROOF_GATE, "host.proof_gate", 0, 0, 1, 0, "PROC. several proofs in flight on one device: admit one at a time to the throughput-bound part") \
  X(HOST_PIN_THREAD, "host.pin_thread", 1, 0, 1, 0, "PROC. sp_ctx_create narrows the calling thread's CPU affinity to the cores of the GPU's own NUMA node (sysfs local_cpulist of its PCI function): a proof is ~330 PCIe round trips between that thread and the device; 0 = leave the affinity alone") \
  X(IPA_UNIFIED_TREE, "ipa.unified_tree", 0, 0, 1, 1, "inner-product rounds always with the unified (complete) addition tree")               \
  X(IPA_DEDICATED_UPLOADED, "ipa.dedicated_uploaded", 0, 0, 1, 1, "dedicated (incomplete, two-multiplication) addition tree also for caller-supplied generator lists (default: only for sets the library derived by hash-to-curve)") \
  X(IPA_FUSED, "ipa.fused", 1, 0, 1, 1, "one launch per inner-product round (0: prepare + lookups + reduce, the form vectors longer than 16384 fall back to - its test hook)")                                \
  X(IPA_RERUN_EXCEPTIONAL, "ipa.rerun_exceptional", 1, 0, 1, 1, "re-run a round with the unified tree when the dedicated tree met an exceptional sum") \
  X(IPA_FINISH_DEVICE, "ipa.finish_device", 0, 0, 1, 1, "the end of an inner-product argument on the device instead of the proving core")     \
  X(ENCODE_DEVICE, "encode.device", 0, 0, 1, 1, "every RFC 9496 encode on the device (few-row commitments are encoded by the proving core by default)") \
  X(COMMIT_SMALL_DEVICE, "commit.small_device", 0, 0, 1, 1, "the 2..5-term Sigma-protocol commitments on the device instead of the proving core") \
  X(MSM_Q_WAVES, "msm.q_waves", 12, 4, 12, 1, "queue form: wavefronts per workgroup (4 / 8 / 12 = 1 / 2 / 3 per SIMD; one workgroup per CU)")                      \
  X(MSM_Q_BG_WAVES, "msm.q_bg_waves", 8, 4, 12, 1, "queue form: wavefronts per workgroup of a launch that shares the chip (the background launch, a foreground launch next to one)")                                 \
  X(MSM_Q_UNITS, "msm.q_units", 32, 4, 4096, 1, "queue form: (column, window) units per queue item")                                             \
  X(UPLOAD_OVERLAP, "upload.overlap", 1, 0, 1, 1, "witness commit issued in row chunks behind the upload")                                   \
  X(UPLOAD_THREAD, "upload.thread", 1, 0, 1, 1, "witness upload + commit issued by a helper thread while the proving thread hashes the transcript prefix") \
  X(SUMCHECK_INLINE_ARGS, "sumcheck.inline_args", 1, 0, 1, 1, "table pointers of the batched sum-check kernels in the kernel arguments (0: the staged form that more than 24 instances or 13 variables fall back to - its test hook)")      \
  X(SUMCHECK_DOUBLE_ROUND_MAX_LEN, "sumcheck.double_round_max_len", 4096, 0, 1073741824, 1, "two rounds per trip while the tables have at most this many entries") \
  X(SUMCHECK_HOST_TAIL, "sumcheck.host_tail", 1, 0, 1, 1, "last <= 3 rounds of a batched sum-check on the proving core")                     \
  X(SUMCHECK_LAUNCH_AHEAD, "sumcheck.launch_ahead", 0, 0, 2, 1, "1: two-rounds-per-trip kernels are enqueued one trip ahead and wait for their challenges on a bell in host memory (measured: -0.3 ms per 2^20 proof in five sessions, +-0 in four others, inside the run-to-run spread: not the default); 0: launched when the challenges are known; 2: test hook - the bell is never rung, every such launch gives up and is repeated the ordinary way") \
  X(SPARK_PROD_LAYER2, "spark.prod_layer2", 1, 0, 1, 1, "two product-circuit layers per launch in the launch-sized middle of the tree")       \
  X(SPARK_PROD_LAYER2_MAX_LOG2, "spark.prod_layer2_max_log2", 18, 0, 40, 1, "... for layers of at most 2^this entries")                      \
  X(SPARK_EQ_FACTOR, "spark.eq_factor", 1, 0, 1, 1, "the eq table as a factor in the throughput-sized batched rounds")                       \
  X(SPARK_HASH_FUSE, "spark.hash_fuse", 1, 0, 1, 1, "hash layer + first product layer in one pass")                                          \
  X(POLYEVAL_EVAL_FROM_OPENING, "polyeval.eval_from_opening", 1, 0, 1, 1, "R1CSProof::prove: the witness evaluation at ry as <LZ, R> from the opening's own vector-matrix product (0: a separate DensePolynomial::evaluate pass)") \
  X(OVERLAP_DEREFS, "overlap.derefs", 1, 0, 1, 1, "row half of the derefs commitment on the background stream under the second sum-check")    \
  X(OVERLAP_EVAL_AHEAD, "overlap.eval_ahead", 1, 0, 1, 1, "R1CSInstance::evaluate queued on a low-priority stream as soon as ry is known")    \
  X(SHARD_RESIDUE_TRANSPORT, "shard.residue_transport", 0, 0, 1, 1, "residue-shard every sum-check over multi-process transports (= shard.residue_min_log2 0)") \
  X(SHARD_CUBIC_MIN_LEN, "shard.cubic_min_len", 0, 0, 1073741824, 1, "batched cubic sum-checks stay sharded while a bind leaves at least this many entries (a power of two >= 32; 0 = twice sumcheck.double_round_max_len)") \
  X(HOST_CALLSTATS, "host.callstats", 0, 0, 1, 1, "PROC. per-entry-point wall time on stderr at exit")                                        \
  X(DEBUG_KTIME, "debug.ktime", 0, 0, 1, 1, "SP_KTIME builds: in-kernel time stamps")

enum SpOpt {
#define X(id, key, def, lo, hi, tier, doc) OPT_##id,
  SP_OPTION_TABLE(X)
#undef X
  OPT_COUNT
};
struct SpOptDesc { const char* key; long long def, lo, hi; int tier; const char* doc; };
extern const SpOptDesc kOptDesc[OPT_COUNT];
struct SpOptions { long long v[OPT_COUNT]; };
// process-wide defaults (compiled-in defaults overlaid by SPARTAN_OPTIONS once, then by sp_ctx_set_option(NULL, ...))
SpOptions sp_default_options();
