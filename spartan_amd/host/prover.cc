// spartan_amd host driver: the prover of libspartan (src/lib.rs, r1csproof.rs, sumcheck.rs, nizk/, dense_mlpoly.rs,
// sparse_mlpoly.rs, product_tree.rs) re-expressed over device-resident tables and fixed-base MSMs.
// Every group operation and every O(N) field pass is an sp_* call (HIP); see libspartan.hpp.
#include "libspartan.hpp"
#include "fq_inv.hpp"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <map>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>

namespace spz {
using namespace sp;

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
// optional per-entry-point wall-clock accounting (option host.callstats = 1): where the latency of a proof accumulates.
// Diagnostic only: the table is process-wide and unsynchronised, so use it with one proving thread.
struct CallStats {
  struct E { double t = 0; size_t n = 0; };
  std::map<std::string, E> m;
  bool on = ctx_opt(nullptr, "host.callstats") != 0;  // process-wide option (SPARTAN_OPTIONS=testing.unlock=1,host.callstats=1)
  ~CallStats() { dump(); }
  void dump() {
    if (!on || m.empty()) return;
    std::vector<std::pair<std::string, E>> v(m.begin(), m.end());
    std::sort(v.begin(), v.end(), [](auto& a, auto& b) { return a.second.t > b.second.t; });
    fprintf(stderr, "[callstats] %-34s %8s %12s %10s\n", "entry point", "calls", "total ms", "avg us");
    for (auto& kv : v) fprintf(stderr, "[callstats] %-34s %8zu %12.3f %10.1f\n", kv.first.c_str(), kv.second.n, kv.second.t * 1e3, kv.second.t / kv.second.n * 1e6);
    m.clear();
  }
};
static CallStats g_calls;
#ifdef SPZ_HOSTPROF
// diagnostic build only: host-exclusive time per driver function = wall - device-API time - nested spans
static double g_api_t = 0;
struct HostSpan {
  const char* name; double t0, api0, child = 0; HostSpan* parent;
  static HostSpan*& top() { static HostSpan* t = nullptr; return t; }
  static std::map<std::string, std::pair<double, size_t>>& acc() { static std::map<std::string, std::pair<double, size_t>> m; return m; }
  static std::map<std::string, double>& incl() { static std::map<std::string, double> m; return m; }
  explicit HostSpan(const char* n) : name(n), t0(now_s()), api0(g_api_t), parent(top()) { top() = this; }
  ~HostSpan() {
    double wall = now_s() - t0, api = g_api_t - api0;
    auto& e = acc()[name]; e.first += wall - api - child; e.second++;
    incl()[name] += wall;
    if (parent) parent->child += wall - api;
    top() = parent;
  }
  static void dump() {
    for (auto& kv : acc())
      fprintf(stderr, "[hostprof] %-28s %6zu calls %9.3f ms host-exclusive %9.3f ms wall-inclusive\n", kv.first.c_str(), kv.second.second,
              kv.second.first * 1e3, incl()[kv.first] * 1e3);
    acc().clear();
    incl().clear();
  }
};
#define HSPAN(n) HostSpan hspan_(n)
#define HAPI(dt) g_api_t += (dt)
#else
#define HSPAN(n)
#define HAPI(dt)
#endif
#ifdef SPZ_HOSTPROF
#define SPX_T0(v) v = now_s();
#define SPX_T1(v) HAPI(now_s() - v);
#else
#define SPX_T0(v)
#define SPX_T1(v)
#endif
#define SPX(call)                                                                                        \
  do {                                                                                                   \
    double t0_ = (g_calls.on || sizeof(#call) == 0) ? now_s() : 0;                                       \
    SPX_T0(t0_)                                                                                          \
    int32_t rc_ = (call);                                                                                \
    SPX_T1(t0_)                                                                                          \
    if (g_calls.on) {                                                                                    \
      std::string k_(#call);                                                                             \
      auto& e_ = g_calls.m[k_.substr(0, k_.find('('))];                                                  \
      e_.t += now_s() - t0_;                                                                             \
      e_.n++;                                                                                            \
    }                                                                                                    \
    if (rc_ != SP_OK) throw Error(std::string(#call) + " failed: " + sp_strerror(rc_) + " (" + std::to_string(rc_) + ")"); \
  } while (0)
#define REQUIRE(cond)                                                    \
  do {                                                                   \
    if (!(cond)) throw Error(std::string("requirement failed: ") + #cond); \
  } while (0)

static inline const uint64_t* U(const Fq& x) { return x.l; }
static inline const uint64_t* U(const FqVec& v) { return v.empty() ? nullptr : v[0].l; }
static inline uint64_t* U(FqVec& v) { return v.empty() ? nullptr : v[0].l; }
static size_t pow2(size_t e) { return (size_t)1 << e; }
static size_t log_2(size_t n) {  // math.rs:21-29: ceil for non powers of two
  size_t l = 0;
  while (((size_t)1 << l) < n) l++;
  return l;
}
static size_t next_pow2(size_t n) { return pow2(log_2(n == 0 ? 1 : n)); }

// ------------------------------------------------------------------ handles
Ctx::Ctx(int device) { SPX(sp_ctx_create(device, &h)); }
Ctx::~Ctx() {
  commit_shard_forget(h);
  sp_ctx_destroy(h);
}
DevTable::~DevTable() { sp_table_free(h); }
DevTable& DevTable::operator=(DevTable&& o) noexcept {
  if (this != &o) { sp_table_free(h); c = o.c; h = o.h; o.h = nullptr; }
  return *this;
}
DevIndex::~DevIndex() { sp_index_free(h); }
DevIndex& DevIndex::operator=(DevIndex&& o) noexcept {
  if (this != &o) { sp_index_free(h); c = o.c; h = o.h; o.h = nullptr; }
  return *this;
}
GensStream::~GensStream() { sp_gens_free(g); }
GensStream& GensStream::operator=(GensStream&& o) noexcept {
  if (this != &o) {
    sp_gens_free(g); c = o.c; g = o.g; compressed = std::move(o.compressed); o.g = nullptr;
  }
  return *this;
}

static DevTable tab_alloc(sp_ctx* c, size_t len) { sp_table* t; SPX(sp_table_alloc(c, len, &t)); return DevTable(c, t); }
static DevTable tab_alloc_uninit(sp_ctx* c, size_t len) { sp_table* t; SPX(sp_table_alloc_uninit(c, len, &t)); return DevTable(c, t); }
static DevTable tab_upload(sp_ctx* c, const FqVec& v) { sp_table* t; SPX(sp_table_upload(c, U(v), v.size(), &t)); return DevTable(c, t); }
static DevTable tab_view(sp_ctx* c, const DevTable& p, size_t off, size_t len) { sp_table* t; SPX(sp_table_view(c, p.h, off, len, &t)); return DevTable(c, t); }
static DevTable tab_clone(sp_ctx* c, const DevTable& p) { sp_table* t; SPX(sp_table_clone(c, p.h, &t)); return DevTable(c, t); }
static DevTable tab_eq(sp_ctx* c, const FqVec& r) { sp_table* t; SPX(sp_eq_expand(c, U(r), r.size(), &t)); return DevTable(c, t); }

// ------------------------------------------------------------------ generators (commitments.rs:15-33)
static const uint8_t kBasepointCompressed[32] = {0xe2, 0xf2, 0xae, 0x0a, 0x6a, 0xbc, 0x4e, 0x71, 0xa8, 0x84, 0xa9, 0x61, 0xc5, 0x00, 0x51, 0x5f,
                                                 0x58, 0xe3, 0x0b, 0x6a, 0xa5, 0x82, 0xdd, 0x8d, 0xb6, 0xa6, 0x59, 0x45, 0xe0, 0x8d, 0x2d, 0x76};
GensStream::GensStream(sp_ctx* c_, const char* label, size_t npoints, int windows) : c(c_) {
  // windows != 0: the table geometry SNARKGens planned for this stream together with its sibling (sp_gens_plan_pair): handed to the
  // library as option msm.windows for the time of this creation
  struct WindowsFor { sp_ctx* c; bool on; WindowsFor(sp_ctx* c_, int w) : c(c_), on(w != 0) { if (on) SPX(sp_ctx_set_option(c, "msm.windows", std::to_string(w).c_str())); }
                      ~WindowsFor() { if (on) (void)sp_ctx_set_option(c, "msm.windows", "0"); } } windows_for(c_, windows);
  Shake256 shake;  // commitments.rs:16-19
  shake.absorb(label, strlen(label));
  shake.absorb(kBasepointCompressed, 32);
  std::vector<uint8_t> uniform(64 * npoints);
  shake.squeeze(uniform.data(), uniform.size());
  compressed.resize(32 * npoints);
  SPX(sp_gens_from_uniform(c, uniform.data(), npoints, compressed.data(), &g));  // from_uniform_bytes on the device (:21-30)
}
MultiCommitGens GensStream::multi_commit_gens(size_t n) const {
  REQUIRE(n + 1 <= sp_gens_len(g));
  MultiCommitGens m;
  m.g = g;
  m.G.resize(n);
  for (size_t i = 0; i < n; i++) m.G[i] = (uint32_t)i;
  m.h = (uint32_t)n;
  return m;
}
DotProductProofGens GensStream::dot_product_gens(size_t n) const {  // nizk/mod.rs:415-418: new(n+1).split_at(n)
  MultiCommitGens all = multi_commit_gens(n + 1);
  DotProductProofGens d;
  d.n = n;
  d.gens_n.g = d.gens_1.g = g;
  d.gens_n.G.assign(all.G.begin(), all.G.begin() + n);
  d.gens_1.G.assign(all.G.begin() + n, all.G.end());
  d.gens_n.h = d.gens_1.h = all.h;
  return d;
}
PolyCommitmentGens GensStream::poly_commitment_gens(size_t num_vars) const {  // dense_mlpoly.rs:31-35
  size_t right = num_vars - num_vars / 2;
  return PolyCommitmentGens{dot_product_gens(pow2(right))};
}
static size_t pad_vars(size_t num_vars, size_t num_inputs) {  // lib.rs:288-294
  size_t v = num_vars > num_inputs + 1 ? num_vars : num_inputs + 1;
  return next_pow2(v);
}
static R1CSGens make_r1cs_gens(const GensStream& s, size_t num_vars_padded) {  // r1csproof.rs:48-73
  R1CSGens g;
  g.gens_pc = s.poly_commitment_gens(log_2(num_vars_padded));
  g.gens_sc.gens_1 = g.gens_pc.gens.gens_1;
  g.gens_sc.gens_3 = s.multi_commit_gens(3);
  g.gens_sc.gens_4 = s.multi_commit_gens(4);
  return g;
}
static size_t sat_stream_points(size_t nvp) {
  size_t lv = log_2(nvp);
  size_t n = pow2(lv - lv / 2) + 2;
  return n < 5 ? 5 : n;
}
NIZKGens::NIZKGens(Ctx& ctx, size_t, size_t num_vars, size_t num_inputs)
    : stream_sat(ctx.h, "gens_r1cs_sat", sat_stream_points(pad_vars(num_vars, num_inputs))),
      gens_r1cs_sat(make_r1cs_gens(stream_sat, pad_vars(num_vars, num_inputs))) {}
SNARKGens::SNARKGens(Ctx& ctx, size_t num_cons, size_t num_vars, size_t num_inputs, size_t nnz) {
  size_t nvp = pad_vars(num_vars, num_inputs);
  // r1cs.rs:33-48 -> sparse_mlpoly.rs:313-337, batch_size = 3
  size_t nvx = log_2(num_cons), nvy = log_2(2 * nvp);
  size_t num_vars_ops = log_2(next_pow2(nnz)) + log_2(next_pow2(3 * 5));
  size_t num_vars_mem = (nvx > nvy ? nvx : nvy) + 1;
  size_t num_vars_derefs = log_2(next_pow2(nnz)) + log_2(next_pow2(3 * 2));
  size_t mx = 0;
  for (size_t v : {num_vars_ops, num_vars_mem, num_vars_derefs}) mx = std::max(mx, pow2(v - v / 2));
  // the window tables of the two streams are sized TOGETHER (round 6): a proof commits num_vars scalars under the first stream and
  // 6 * nnz (`derefs`) under the second; the library picks the pair of window counts with the fewest additions that fits the free HBM
  int w_sat = 0, w_eval = 0;
  SPX(sp_gens_plan_pair(ctx.h, sat_stream_points(nvp), mx + 2, (double)nvp, 6.0 * (double)next_pow2(nnz), &w_sat, &w_eval));
  stream_sat = GensStream(ctx.h, "gens_r1cs_sat", sat_stream_points(nvp), w_sat);
  gens_r1cs_sat = make_r1cs_gens(stream_sat, nvp);
  stream_eval = GensStream(ctx.h, "gens_r1cs_eval", mx + 2, w_eval);
  gens_r1cs_eval.gens_ops = stream_eval.poly_commitment_gens(num_vars_ops);
  gens_r1cs_eval.gens_mem = stream_eval.poly_commitment_gens(num_vars_mem);
  gens_r1cs_eval.gens_derefs = stream_eval.poly_commitment_gens(num_vars_derefs);
}

// ------------------------------------------------------------------ instance (lib.rs:121-228)
Instance::Instance(Ctx& ctx, size_t nc, size_t nv, size_t ni, const std::vector<SparseEntry>& A_, const std::vector<SparseEntry>& B_,
                   const std::vector<SparseEntry>& C_)
    : c(ctx.h) {
  // padding rules: lib.rs:129-156
  size_t nvp = next_pow2(std::max(nv, ni + 1));
  size_t ncp = nc;
  if (nc == 0 || nc == 1) ncp = 2;
  if (next_pow2(nc) != nc) ncp = next_pow2(nc);
  size_t shift = nvp - nv;
  auto conv = [&](const std::vector<SparseEntry>& in, std::vector<SparseEntry>& out) {
    for (auto& e : in) {
      if (e.row >= nc || e.col >= nv + 1 + ni) throw Error("InvalidIndex");  // lib.rs:164-172
      SparseEntry o = e;
      if (e.col >= nv) o.col = e.col + shift;  // lib.rs:178-182: references to 1 / inputs move with the padding
      out.push_back(o);
    }
    // lib.rs:188-194: explicit zero constraints only when the original count was 0 or 1
    if (nc == 0 || nc == 1)
      for (size_t i = in.size(); i < ncp; i++) out.push_back(SparseEntry{i, nv, fq_zero()});
  };
  conv(A_, A); conv(B_, B); conv(C_, C);
  num_cons = ncp; num_vars = nvp; num_inputs = ni;
  auto up = [&](const std::vector<SparseEntry>& m, sp_sparse** d) {
    std::vector<uint64_t> r(m.size()), cc(m.size());
    FqVec v(m.size());
    for (size_t i = 0; i < m.size(); i++) { r[i] = m[i].row; cc[i] = m[i].col; v[i] = m[i].val; }
    SPX(sp_sparse_upload(c, r.data(), cc.data(), U(v), m.size(), num_cons, 2 * num_vars, d));
  };
  try {
    up(A, &dA); up(B, &dB); up(C, &dC);
  } catch (...) {  // a constructor that throws gets no destructor call: release what was uploaded
    sp_sparse_free(dA); sp_sparse_free(dB); sp_sparse_free(dC);
    throw;
  }
}
Instance::~Instance() { sp_sparse_free(dA); sp_sparse_free(dB); sp_sparse_free(dC); }
std::vector<uint8_t> Instance::shape_bincode() const {
  std::vector<uint8_t> b;
  auto u64 = [&](uint64_t x) { for (int i = 0; i < 8; i++) b.push_back((uint8_t)(x >> (8 * i))); };
  size_t total = 24;
  for (auto m : {&A, &B, &C}) total += 24 + 48 * m->size();
  b.reserve(total);
  u64(num_cons); u64(num_vars); u64(num_inputs);
  for (auto m : {&A, &B, &C}) {
    u64(log_2(num_cons)); u64(log_2(2 * num_vars)); u64(m->size());  // SparseMatPolynomial { num_vars_x, num_vars_y, M } (r1cs.rs:104-112)
    for (auto& e : *m) { u64(e.row); u64(e.col); for (int i = 0; i < 4; i++) u64(e.val.l[i]); }
  }
  return b;
}
std::vector<uint8_t> Instance::compute_digest() const {  // by value, copied under the lock: a concurrent set_digest cannot pull the bytes from under a proof
  std::lock_guard<std::mutex> lk(digest_mu);
  if (digest.empty()) {
    static std::once_flag said;
    std::call_once(said, [] {
      fprintf(stderr, "spartan_host: R1CSShapeDigest computed by the in-tree deflater (deflate.cc: byte-identical to miniz 3.0.2 level 6 on the test corpus; "
                      "equality with miniz_oxide is the port's claim) — pass the digest of a libspartan Instance with set_digest to bypass it\n");
    });
    std::vector<uint8_t> sb = shape_bincode();
    digest = zlib_level6_miniz(sb.data(), sb.size(), digest_old_header);
  }
  return digest;
}
void Instance::set_digest(const uint8_t* d, size_t n) {
  std::lock_guard<std::mutex> lk(digest_mu);
  digest.assign(d, d + n);
}
bool Instance::set_digest_header(bool old_header) {  // false once a digest exists: the header variant is a property of the bytes already handed out
  std::lock_guard<std::mutex> lk(digest_mu);
  if (!digest.empty() && old_header != digest_old_header) return false;
  digest_old_header = old_header;
  return true;
}

Fq seed_scalar(const char* domain, uint64_t seed) {
  Shake256 sh;
  sh.absorb(domain, strlen(domain));
  uint8_t sb[8];
  for (int i = 0; i < 8; i++) sb[i] = (uint8_t)(seed >> (8 * i));
  sh.absorb(sb, 8);
  uint8_t buf[64];
  sh.squeeze(buf, 64);
  uint64_t w[8];
  memcpy(w, buf, 64);
  return fq_from_u512(w);
}
std::unique_ptr<Instance> Instance::produce_synthetic_r1cs(Ctx& ctx, size_t num_cons, size_t num_vars, size_t num_inputs, uint64_t seed,
                                                           FqVec* vars, FqVec* inputs) {
  REQUIRE(pow2(log_2(num_cons)) == num_cons && pow2(log_2(num_vars)) == num_vars && num_inputs < num_vars);  // r1cs.rs:165-167
  size_t size_z = num_vars + num_inputs + 1;
  Shake256 sh;
  const char* dom = "spartan-synthetic-r1cs";
  sh.absorb(dom, strlen(dom));
  uint8_t sb[8];
  for (int i = 0; i < 8; i++) sb[i] = (uint8_t)(seed >> (8 * i));
  sh.absorb(sb, 8);
  FqVec Z(size_z);  // Scalar::random = from_u512 of 64 random bytes (ristretto255.rs:374-380)
  for (size_t i = 0; i < size_z; i++) {
    uint8_t buf[64];
    sh.squeeze(buf, 64);
    uint64_t w[8];
    memcpy(w, buf, 64);
    Z[i] = fq_from_u512(w);
  }
  Z[num_vars] = fq_one();  // r1cs.rs:187
  // C_val = Z_A * Z_B / Z_C (r1cs.rs:205-213): one batch inversion instead of num_cons inversions (same values)
  FqVec cinv(num_cons), pref(num_cons);
  Fq acc = fq_one();
  for (size_t i = 0; i < num_cons; i++) {
    Fq cz = Z[(i + 3) % size_z];
    cinv[i] = fq_is_zero(cz) ? fq_one() : cz;
    pref[i] = acc;
    acc = acc * cinv[i];
  }
  acc = fq_invert(acc);
  for (size_t i = num_cons; i-- > 0;) {
    Fq tmp = acc * cinv[i];
    cinv[i] = acc * pref[i];
    acc = tmp;
  }
  std::vector<SparseEntry> A, B, C;
  for (size_t i = 0; i < num_cons; i++) {
    size_t A_idx = i % size_z, B_idx = (i + 2) % size_z, C_idx = (i + 3) % size_z;
    A.push_back({i, A_idx, fq_one()});
    B.push_back({i, B_idx, fq_one()});
    Fq AB = Z[A_idx] * Z[B_idx];
    if (fq_is_zero(Z[C_idx])) C.push_back({i, num_vars, AB});
    else C.push_back({i, C_idx, AB * cinv[i]});
  }
  vars->assign(Z.begin(), Z.begin() + num_vars);
  inputs->assign(Z.begin() + num_vars + 1, Z.end());
  return std::unique_ptr<Instance>(new Instance(ctx, num_cons, num_vars, num_inputs, A, B, C));
}

// ------------------------------------------------------------------ commitments (commitments.rs:73-92)
static CP to_cp(const uint8_t* p) { CP c; memcpy(c.data(), p, 32); return c; }
// rows of scalars over an explicit generator index list
static std::vector<CP> msm_rows(sp_ctx* c, const sp_gens* g, const std::vector<uint32_t>& idx, const FqVec& scalars, size_t rows) {
  REQUIRE(scalars.size() == rows * idx.size());
  std::vector<uint8_t> out(32 * rows);
  // few-term commitments: on this core unless told otherwise (the library's host-side engine, csrc/host_commit.hip)
  if (idx.size() <= 8 && small_msm_on_host(c)) SPX(sp_host_commit_small(g, idx.data(), idx.size(), U(scalars), rows, nullptr, out.data()));
  else SPX(sp_msm_indexed(c, g, idx.data(), idx.size(), U(scalars), rows, out.data()));
  std::vector<CP> r(rows);
  for (size_t i = 0; i < rows; i++) r[i] = to_cp(&out[32 * i]);
  return r;
}
static CP commit_scalar(sp_ctx* c, const Fq& x, const Fq& blind, const MultiCommitGens& g1) {  // Scalar::commit :73-78
  REQUIRE(g1.n() == 1);
  return msm_rows(c, g1.g, {g1.G[0], g1.h}, {x, blind}, 1)[0];
}
// DensePolynomial::commit (dense_mlpoly.rs:179-204): L row commitments of the R-wide rows of a device table
static PolyCommitment poly_commit(sp_ctx* c, const DevTable& Z, size_t num_vars, const PolyCommitmentGens& gens, const FqVec* blinds) { HSPAN("poly_commit");
  size_t Ls = pow2(num_vars / 2), Rs = pow2(num_vars - num_vars / 2);
  const MultiCommitGens& g = gens.gens.gens_n;
  REQUIRE(g.n() == Rs && Z.len() == Ls * Rs);
  REQUIRE(!blinds || blinds->size() == Ls);
  std::vector<uint8_t> out(32 * Ls);
  // rows are independent MSMs over shared generators: sharded over the ranks / virtual shards of the context when configured
  if (!sharded_commit_rows(c, g.g, g.G[0], g.h, Z.h, Ls, Rs, blinds ? U(*blinds) : nullptr, out.data()))
    SPX(sp_commit_rows_dev(c, g.g, g.G[0], g.h, Z.h, 0, Ls, Rs, blinds ? U(*blinds) : nullptr, out.data()));
  PolyCommitment pc;
  pc.C.resize(Ls);
  for (size_t i = 0; i < Ls; i++) pc.C[i] = to_cp(&out[32 * i]);
  return pc;
}
static void append_poly_commitment(Transcript& t, const char* label, const PolyCommitment& c) { HSPAN("append_poly_commitment");  // dense_mlpoly.rs:292-300
  t.append_message(label, "poly_commitment_begin");
  for (auto& pt : c.C) t.append_point("poly_commitment_share", pt.data());
  t.append_message(label, "poly_commitment_end");
}

// ------------------------------------------------------------------ small host-side algebra (O(log n) / O(sqrt n) sized)
static FqVec eq_evals_host(const FqVec& r) { HSPAN("eq_evals_host");  // EqPolynomial::evals (dense_mlpoly.rs:68-84); used only for the sqrt(N)-sized L/R vectors
  size_t ell = r.size();
  FqVec evals(pow2(ell), fq_one());
  size_t size = 1;
  for (size_t j = 0; j < ell; j++) {
    size *= 2;
    for (size_t i = size - 1;; i -= 2) {
      Fq scalar = evals[i / 2];
      evals[i] = scalar * r[j];
      evals[i - 1] = scalar - evals[i];
      if (i == 1) break;
    }
  }
  return evals;
}
struct UniPoly {  // unipoly.rs
  FqVec coeffs;
  static UniPoly from_evals(const FqVec& e) {  // :23-55
    static const Fq two_inv = fq_invert(fq_from_u64(2)), six_inv = fq_invert(fq_from_u64(6));
    UniPoly u;
    if (e.size() == 3) {
      Fq c = e[0];
      Fq a = two_inv * (e[2] - e[1] - e[1] + c);
      Fq b = e[1] - c - a;
      u.coeffs = {c, b, a};
    } else {
      REQUIRE(e.size() == 4);
      Fq d = e[0];
      Fq a = six_inv * (e[3] - e[2] - e[2] - e[2] + e[1] + e[1] + e[1] - e[0]);
      Fq b = two_inv * (e[0] + e[0] - e[1] - e[1] - e[1] - e[1] - e[1] + e[2] + e[2] + e[2] + e[2] - e[3]);
      Fq c = e[1] - d - a - b;
      u.coeffs = {d, c, b, a};
    }
    return u;
  }
  size_t degree() const { return coeffs.size() - 1; }
  Fq evaluate(const Fq& r) const {  // :72-80
    Fq eval = coeffs[0], power = r;
    for (size_t i = 1; i < coeffs.size(); i++) { eval += power * coeffs[i]; power *= r; }
    return eval;
  }
  FqVec compress() const {  // :82-88
    FqVec c;
    c.push_back(coeffs[0]);
    c.insert(c.end(), coeffs.begin() + 2, coeffs.end());
    return c;
  }
  void append_to_transcript(Transcript& t, const char* label) const {  // :112-120
    t.append_message(label, "UniPoly_begin");
    for (auto& c : coeffs) t.append_scalar("coeff", c);
    t.append_message(label, "UniPoly_end");
  }
};
// test hook (tests/test_host_transcript.py): the reference's own UniPoly known answers (unipoly.rs:127-183)
void unipoly_probe(const FqVec& evals, const Fq& r, FqVec* coeffs, FqVec* compressed, Fq* eval_at_r) {
  UniPoly u = UniPoly::from_evals(evals);
  *coeffs = u.coeffs;
  *compressed = u.compress();
  *eval_at_r = u.evaluate(r);
}
static Fq dot_host(const FqVec& a, const FqVec& b) {
  Fq s = fq_zero();
  for (size_t i = 0; i < a.size(); i++) s += a[i] * b[i];
  return s;
}

// ------------------------------------------------------------------ nizk/mod.rs Sigma protocols
static KnowledgeProof knowledge_prove(sp_ctx* c, const MultiCommitGens& g, Transcript& t, RandomTape& tape, const Fq& x, const Fq& r, CP* C_out) { HSPAN("knowledge_prove");
  t.append_protocol_name("knowledge proof");  // :27-52
  Fq t1 = tape.random_scalar("t1"), t2 = tape.random_scalar("t2");
  std::vector<CP> cm = msm_rows(c, g.g, {g.G[0], g.h}, {x, r, t1, t2}, 2);
  t.append_point("C", cm[0].data());
  t.append_point("alpha", cm[1].data());
  Fq ch = t.challenge_scalar("c");
  *C_out = cm[0];
  return KnowledgeProof{cm[1], x * ch + t1, r * ch + t2};
}
static EqualityProof equality_prove(sp_ctx* c, const MultiCommitGens& g, Transcript& t, RandomTape& tape, const Fq& v1, const Fq& s1, const Fq& v2,
                                    const Fq& s2) { HSPAN("equality_prove");
  t.append_protocol_name("equality proof");  // :88-116
  Fq r = tape.random_scalar("r");
  std::vector<CP> cm = msm_rows(c, g.g, {g.G[0], g.h}, {v1, s1, v2, s2, fq_zero(), r}, 3);  // C1, C2, alpha = r*h
  t.append_point("C1", cm[0].data());
  t.append_point("C2", cm[1].data());
  t.append_point("alpha", cm[2].data());
  Fq ch = t.challenge_scalar("c");
  return EqualityProof{cm[2], ch * (s1 - s2) + r};
}
static ProductProof product_prove(sp_ctx* c, const MultiCommitGens& g, Transcript& t, RandomTape& tape, const Fq& x, const Fq& rX, const Fq& y,
                                  const Fq& rY, const Fq& z, const Fq& rZ, CP* Xo, CP* Yo, CP* Zo) { HSPAN("product_prove");
  t.append_protocol_name("product proof");  // :159-227
  Fq b1 = tape.random_scalar("b1"), b2 = tape.random_scalar("b2"), b3 = tape.random_scalar("b3");
  Fq b4 = tape.random_scalar("b4"), b5 = tape.random_scalar("b5");
  // delta = b3*X + b5*h with X = x*G + rX*h  ==  (b3*x)*G + (b3*rX + b5)*h : same group element, fixed-base form
  std::vector<CP> cm = msm_rows(c, g.g, {g.G[0], g.h}, {x, rX, y, rY, z, rZ, b1, b2, b3, b4, b3 * x, b3 * rX + b5}, 6);
  t.append_point("X", cm[0].data());
  t.append_point("Y", cm[1].data());
  t.append_point("Z", cm[2].data());
  t.append_point("alpha", cm[3].data());
  t.append_point("beta", cm[4].data());
  t.append_point("delta", cm[5].data());
  Fq ch = t.challenge_scalar("c");
  ProductProof p;
  p.alpha = cm[3]; p.beta = cm[4]; p.delta = cm[5];
  p.z[0] = b1 + ch * x; p.z[1] = b2 + ch * rX; p.z[2] = b3 + ch * y; p.z[3] = b4 + ch * rY; p.z[4] = b5 + ch * (rZ - rX * y);
  *Xo = cm[0]; *Yo = cm[1]; *Zo = cm[2];
  return p;
}

// Where the few-term commitments run (libspartan.hpp)

// The tape-only halves of a ZK sum-check's commitments, computed ahead of the rounds by the library's helper thread
// (sp_host_zk_ahead_*, csrc/host_commit.hip) while the proving thread is in its first evaluation: the DotProductProof's
// delta = commit(d_j, r_delta_j) is complete (nothing but the tape enters it), and of comm_eval, beta and comm_poly the blind
// terms blinds_evals[j]*h, r_beta_j*h, blinds_poly[j]*h_n are. The tape is read in the reference's order: inside the round loop
// nothing but DotProductProof::prove draws from it (d_vec, r_delta, r_beta: nizk/mod.rs:330-334), so drawing the rounds'
// values one after the other up front is the same stream.
struct ZkAhead {
  std::vector<FqVec> d;
  FqVec r_delta, r_beta;
  sp_zk_ahead* h = nullptr;
  struct Round { CP delta; sp_host_point be_h, rb_h, bp_hn; };
  std::vector<Round> got;      // rounds fetched so far (they complete in order)
  ~ZkAhead() { sp_host_zk_ahead_free(h); }
  const Round& wait(size_t j) {
    while (got.size() <= j) {
      Round r;
      SPX(sp_host_zk_ahead_wait(h, got.size(), r.delta.data(), &r.be_h, &r.rb_h, &r.bp_hn));
      got.push_back(r);
    }
    return got[j];
  }
};

// ------------------------------------------------------------------ sum-check tables sharded by index residue
// Shard g of W holds the entries {i : i = g (mod W)} of every table as a contiguous sub-table T_g[k] = T[k W + g]. The
// top-variable pair (i, i + len/2) of a round (sumcheck.rs:460-469, 624-652) stays on one shard while len/2 is a multiple of W, so
// every shard runs the ordinary round kernels on its sub-tables and a round exchanges only its 2..3 partial sums (<= 96 bytes per
// shard), added in F_q here (the modular addition is no RCCL operator: all-gather, then add). When the sub-tables are down to
// one entry the W survivors are gathered into the owner's tables (entry g of a table = shard g's entry) and the last log2(W)
// rounds run unsharded. Same field values as the unsharded rounds — sums of the same terms — hence the same proof bytes.
struct ResidueShards {
  sp_ctx* owner = nullptr;
  std::vector<sp_ctx*> ctxs;               // virtual shards: every residue class lives on a context of this process
  std::vector<std::vector<DevTable>> sub;  // [shard][table]; with a multi-process transport: [0] = this rank's residue class
  std::vector<std::vector<sp_table*>> h;
  bool active = false, remote = false;     // remote: one residue class per lock-step rank, partial sums over the commit transport
  bool in_flight = false;                  // a bind + evaluation has been started on every shard context and not collected yet
  size_t ntab = 0, W = 0, me = 0;
  // An exception between bind_eval_start and the collect (a failed commitment on the proving core, one shard's collect failing) must
  // not leave pend_eval armed on the persistent sub-contexts — every later sp_sumcheck_bind_eval_start on them would return SP_EINVAL.
  void drain() {
    if (!in_flight) return;
    in_flight = false;
    uint64_t sink[12];
    for (sp_ctx* x : ctxs) (void)sp_sumcheck_bind_eval_collect(x, sink);
  }
  ~ResidueShards() { drain(); }
  void split(sp_ctx* c, const std::vector<sp_table*>& tabs) {
    size_t len = tabs.empty() ? 0 : sp_table_len(tabs[0]);
    if (commit_shard_residue_off(c)) return;  // (the switches were resolved when the sharding was configured, the same on every rank)
    ctxs = residue_shard_ctxs(c);
    int rank = 0, world = 0, min_log2 = 64;
    if (ctxs.size() >= 2) {
      W = ctxs.size();
    } else if (commit_shard_transport(c, &rank, &world, &min_log2) && min_log2 < 64 && len >= ((size_t)1 << min_log2)) {
      // SURVEY 8e over real ranks: every rank has the full tables (it computed them like every other rank) and keeps only its residue
      // class from here on; a round's partial sums travel over the transport the sharded commitments use (96 bytes per rank). Chosen by
      // TABLE LENGTH: the exchange (26 us over RCCL) must be shorter than the round it shortens — tables of >= 2^22 entries (DESIGN.md,
      // section 6); option shard.residue_min_log2 moves the threshold, shard.residue_transport = 1 is the old "always"
      W = (size_t)world; me = (size_t)rank; remote = true;
      ctxs.assign(1, c);
    } else {
      return;
    }
    if (W < 2 || (W & (W - 1)) || len < 4 * W) { remote = false; return; }
    owner = c; ntab = tabs.size();
    SPX(sp_ctx_sync(c));  // the tables as produced by everything queued on the owning context
    const size_t nloc = remote ? 1 : W;
    sub.resize(nloc); h.resize(nloc);
    for (size_t g = 0; g < nloc; g++)
      for (sp_table* t : tabs) {
        sp_table* o = nullptr;
        SPX(sp_table_residue_split(ctxs[g], t, W, remote ? me : g, &o));
        sub[g].emplace_back(ctxs[g], o);
        h[g].push_back(o);
      }
    active = true;
  }
  size_t sub_len() const { return sp_table_len(h[0][0]); }
  // parts: the partial sums of the residue classes held here; the others come over the transport. ev = their sum (F_q: exact, any order).
  void add_partials(uint64_t* ev, std::vector<std::array<uint64_t, 12>>& parts, int n) {
    if (remote) {
      std::vector<std::array<uint64_t, 12>> all(W);
      all[me] = parts[0];
      commit_shard_gather(owner, (uint8_t*)all.data(), sizeof(all[0]));
      parts.swap(all);
    }
    for (int k = 0; k < n; k++) {
      Fq acc = fq_zero();
      for (auto& p : parts) { Fq x; memcpy(x.l, &p[4 * k], 32); acc += x; }
      memcpy(ev + 4 * k, acc.l, 32);
    }
    if (!remote) commit_shard_note_gather(owner, 32 * (size_t)n * parts.size());
  }
  void eval(int kind, uint64_t* ev) {
    std::vector<std::array<uint64_t, 12>> parts(ctxs.size());
    for (size_t g = 0; g < ctxs.size(); g++) { parts[g].fill(0); SPX(sp_sumcheck_eval(ctxs[g], kind, h[g].data(), ntab, parts[g].data())); }
    add_partials(ev, parts, kind == 0 ? 2 : 3);
  }
  void bind_eval_start(int kind, const Fq& r) {  // every shard's bind + next evaluation in flight together (own streams)
    in_flight = true;  // (set first: a start that fails half-way leaves the earlier shards armed, and drain() disarms them)
    for (size_t g = 0; g < ctxs.size(); g++) SPX(sp_sumcheck_bind_eval_start(ctxs[g], kind, h[g].data(), ntab, U(r)));
  }
  void bind_eval_collect(int kind, uint64_t* ev) {
    std::vector<std::array<uint64_t, 12>> parts(ctxs.size());
    int32_t first_bad = SP_OK;  // every shard is collected before anything is thrown
    for (size_t g = 0; g < ctxs.size(); g++) {
      parts[g].fill(0);
      int32_t rc = sp_sumcheck_bind_eval_collect(ctxs[g], parts[g].data());
      if (rc != SP_OK && first_bad == SP_OK) first_bad = rc;
    }
    in_flight = false;
    SPX(first_bad);
    add_partials(ev, parts, kind == 0 ? 2 : 3);
  }
  // sub-tables of two entries: bind them to one and hand the W survivors of every table back to the owner's tables
  void bind_last_and_gather(const Fq& r, std::vector<sp_table*>& tabs) {
    std::vector<FqVec> heads(W, FqVec(ntab));
    if (remote) {
      SPX(sp_table_bind_top_heads(owner, h[0].data(), ntab, U(r), U(heads[me])));
      FqVec all(W * ntab);
      memcpy(&all[me * ntab], heads[me].data(), 32 * ntab);
      commit_shard_gather(owner, (uint8_t*)all.data(), 32 * ntab);
      for (size_t g = 0; g < W; g++) memcpy(heads[g].data(), &all[g * ntab], 32 * ntab);
    } else {
      for (size_t g = 0; g < W; g++) SPX(sp_table_bind_top_heads(ctxs[g], h[g].data(), ntab, U(r), U(heads[g])));
    }
    for (size_t t = 0; t < ntab; t++) {
      FqVec v(W);
      for (size_t g = 0; g < W; g++) v[g] = heads[g][t];
      SPX(sp_table_write(owner, tabs[t], 0, U(v), W));
      SPX(sp_table_set_len(tabs[t], W));
    }
    if (!remote) commit_shard_note_gather(owner, 32 * ntab * W);
    sub.clear(); h.clear();
    active = false;
  }
};

// ------------------------------------------------------------------ sumcheck.rs: the two ZK provers share everything but `kind`
// kind 2: prove_cubic_with_additive_term (:588-776), tables (A,B,C,D), comb A*(B*C-D), gens_n = gens_4
// kind 0: prove_quad (:428-586), tables (A,B), comb A*B, gens_n = gens_3
static ZKSumcheckInstanceProof zk_sumcheck_prove(sp_ctx* c, int kind, const Fq& claim, const Fq& blind_claim, size_t num_rounds,
                                                 std::vector<sp_table*> tabs, const MultiCommitGens& g1, const MultiCommitGens& gn, Transcript& t,
                                                 RandomTape& tape, FqVec* r_out, FqVec* final_claims, Fq* blind_post) { HSPAN("zk_sumcheck_prove");
  FqVec blinds_poly = tape.random_vector("blinds_poly", num_rounds);
  FqVec blinds_evals = tape.random_vector("blinds_evals", num_rounds);
  Fq claim_per_round = claim;
  ZKSumcheckInstanceProof out;
  FqVec r;
  // One generator index list serves every commitment of a round: (gn.G..., gn.h, g1.G, g1.h).
  std::vector<uint32_t> idx_u = gn.G;
  idx_u.push_back(gn.h);
  idx_u.push_back(g1.G[0]);
  idx_u.push_back(g1.h);
  size_t nn = gn.n(), W = idx_u.size();
  // The round's 2..5-term commitments: on this core (small_msm.cc) while the device binds and evaluates, or — with
  // option commit.small_device = 1 — two launches per round on the device
  const bool on_host = small_msm_on_host(c) && gn.g == g1.g;
  ZkAhead ahead;
  if (on_host) {
    ahead.d.resize(num_rounds); ahead.r_delta.resize(num_rounds); ahead.r_beta.resize(num_rounds);
    for (size_t j = 0; j < num_rounds; j++) {
      ahead.d[j] = tape.random_vector("d_vec", nn);  // DotProductProof randomness (nizk/mod.rs:330-332), rounds in order
      ahead.r_delta[j] = tape.random_scalar("r_delta");
      ahead.r_beta[j] = tape.random_scalar("r_beta");
    }
    FqVec dflat;
    for (auto& v : ahead.d) dflat.insert(dflat.end(), v.begin(), v.end());
    SPX(sp_host_zk_ahead_begin(gn.g, idx_u.data(), W, nn, num_rounds, U(blinds_poly), U(blinds_evals), U(dflat), U(ahead.r_delta), U(ahead.r_beta), &ahead.h));
    ahead.got.reserve(num_rounds);  // references into it are handed out
  }
  // rows of scalars over idx_u -> encoded commitments; addend[r] (host mode): a point computed ahead
  auto commit_rows = [&](const FqVec& rows, size_t nrows, const sp_host_point* const* addend) {
    std::vector<CP> cm(nrows);
    if (on_host) {
      std::vector<uint8_t> o(32 * nrows);
      SPX(sp_host_commit_small(gn.g, idx_u.data(), W, U(rows), nrows, addend, o.data()));
      for (size_t k = 0; k < nrows; k++) cm[k] = to_cp(&o[32 * k]);
      return cm;
    }
    return msm_rows(c, gn.g, idx_u, rows, nrows);
  };
  auto make_poly = [&](const uint64_t* ev, const Fq& cl) {
    Fq e0, e2, e3;
    memcpy(e0.l, ev, 32); memcpy(e2.l, ev + 4, 32); memcpy(e3.l, ev + 8, 32);
    return kind == 0 ? UniPoly::from_evals({e0, cl - e0, e2}) : UniPoly::from_evals({e0, cl - e0, e2, e3});
  };
  // comm_poly = commit(poly.coeffs, blinds_poly[j]) under gens_n; in host mode the blind term comes from `ahead`
  auto poly_row = [&](FqVec& rows, size_t row, const UniPoly& poly, size_t j) {
    for (size_t k = 0; k < nn; k++) rows[row * W + k] = poly.coeffs[k];
    rows[row * W + nn] = on_host ? fq_zero() : blinds_poly[j];
  };
  uint64_t ev[12];
  ResidueShards rs;  // virtual shards configured: the rounds run on W residue classes of the tables
  if (on_host) rs.split(c, tabs);
  if (rs.active) rs.eval(kind, ev);
  else SPX(sp_sumcheck_eval(c, kind, tabs.data(), tabs.size(), ev));
  UniPoly poly = make_poly(ev, claim_per_round);
  REQUIRE(poly.coeffs.size() == nn);
  CP comm_claim_per_round, comm_poly;
  {  // comm_claim_per_round (sumcheck.rs:448 / 611) and the first comm_poly (:473 / 661)
    FqVec rows0(2 * W, fq_zero());
    rows0[nn + 1] = claim_per_round; rows0[nn + 2] = blind_claim;
    poly_row(rows0, 1, poly, 0);
    const sp_host_point* add0[2] = {nullptr, on_host ? &ahead.wait(0).bp_hn : nullptr};
    std::vector<CP> cm = commit_rows(rows0, 2, add0);
    comm_claim_per_round = cm[0];
    comm_poly = cm[1];
  }
  for (size_t j = 0; j < num_rounds; j++) {
    t.append_point("comm_poly", comm_poly.data());
    out.comm_polys.push_back(comm_poly);
    Fq r_j = t.challenge_scalar("challenge_nextround");
    // bind every table at r_j (sumcheck.rs:485-486 / 673-676); fused with the next round's evaluation
    bool more = j + 1 < num_rounds;
    // ---- round tail (sumcheck.rs:491-583 / 681-772)
    Fq eval = poly.evaluate(r_j);
    FqVec d;
    Fq r_delta, r_beta;
    if (on_host) { d = ahead.d[j]; r_delta = ahead.r_delta[j]; r_beta = ahead.r_beta[j]; }
    else {
      d = tape.random_vector("d_vec", nn);  // DotProductProof randomness (nizk/mod.rs:330-332)
      r_delta = tape.random_scalar("r_delta"); r_beta = tape.random_scalar("r_beta");
    }
    // comm_eval = eval*G1 + blinds_evals[j]*h ; delta = <d, Gn> + r_delta*hn: their scalars are known as soon as r_j is, like the bind
    CP comm_eval, delta;
    bool pending = false;  // the bind and the next evaluation are in flight on the device while this core commits
    bool resharded = false;  // the shards have just handed their last entries back: the next evaluation is a call of its own
    if (on_host) {
      if (rs.active && rs.sub_len() >= 4) {
        rs.bind_eval_start(kind, r_j);
        pending = true;
      } else if (rs.active) {
        rs.bind_last_and_gather(r_j, tabs);
        resharded = true;
      } else if (sp_table_len(tabs[0]) >= 4) {
        SPX(sp_sumcheck_bind_eval_start(c, kind, tabs.data(), tabs.size(), U(r_j)));
        pending = true;
      } else {
        SPX(sp_table_bind_top(c, tabs.data(), tabs.size(), U(r_j)));
      }
      try {
        const ZkAhead::Round& aj = ahead.wait(j);
        FqVec row(W, fq_zero());
        row[nn + 1] = eval;
        const sp_host_point* add[1] = {&aj.be_h};
        comm_eval = commit_rows(row, 1, add)[0];
        delta = aj.delta;
      } catch (...) { if (pending && !rs.active) (void)sp_sumcheck_bind_eval_collect(c, ev); rs.drain(); throw; }
    } else {
      FqVec rows1(2 * W, fq_zero());
      rows1[nn + 1] = eval; rows1[nn + 2] = blinds_evals[j];
      for (size_t k = 0; k < nn; k++) rows1[W + k] = d[k];
      rows1[W + nn] = r_delta;
      std::vector<CP> cm1;
      if (sp_table_len(tabs[0]) >= 4) {  // one call: the round's r-dependent commitments on a second stream, one wait
        uint8_t pts[64];
        SPX(sp_sumcheck_bind_eval_commit(c, kind, tabs.data(), tabs.size(), U(r_j), ev, gn.g, idx_u.data(), W, U(rows1), 2, pts));
        cm1 = {to_cp(pts), to_cp(pts + 32)};
      } else {
        SPX(sp_table_bind_top(c, tabs.data(), tabs.size(), U(r_j)));
        cm1 = msm_rows(c, gn.g, idx_u, rows1, 2);
      }
      comm_eval = cm1[0]; delta = cm1[1];
    }
    t.append_point("comm_claim_per_round", comm_claim_per_round.data());
    t.append_point("comm_eval", comm_eval.data());
    FqVec w = t.challenge_vector("combine_two_claims_to_one", 2);
    Fq target = w[0] * claim_per_round + w[1] * eval;
    const Fq& blind_sc = (j == 0) ? blind_claim : blinds_evals[j - 1];
    Fq blind = w[0] * blind_sc + w[1] * blinds_evals[j];
    FqVec a(nn);
    Fq pw = fq_one();
    for (size_t i = 0; i < nn; i++) {
      Fq a_sc = (i == 0) ? fq_one() + fq_one() : fq_one();
      a[i] = w[0] * a_sc + w[1] * pw;
      pw = pw * r_j;
    }
    // DotProductProof::prove (nizk/mod.rs:311-370): x = poly.coeffs, blind_x = blinds_poly[j], y = target, blind_y = blind
    t.append_protocol_name("dot product proof");
    t.append_point("Cx", comm_poly.data());  // Cx = commit(x, blind_x): same inputs and generators as comm_poly
    Fq dp = dot_host(a, d);
    // Cy = target*G1 + blind*h ; beta = dp*G1 + r_beta*h ; and the next round's comm_poly (its inputs are the next
    // evaluations and claim = eval)
    UniPoly next_poly;
    std::vector<CP> cm2;
    bool dpp_early = false;
    Fq ch_early = fq_zero();
    if (on_host) {
      try {
        FqVec rows2(2 * W, fq_zero());
        rows2[nn + 1] = target; rows2[nn + 2] = blind;
        rows2[W + nn + 1] = dp;
        const sp_host_point* add2[2] = {nullptr, &ahead.wait(j).rb_h};
        cm2 = commit_rows(rows2, 2, add2);
      } catch (...) { if (pending && !rs.active) (void)sp_sumcheck_bind_eval_collect(c, ev); rs.drain(); throw; }
      // the rest of the DotProductProof's transcript work needs nothing from the device: absorbed (and `c` drawn) while the bind and the
      // next evaluation are still in flight, before they are collected
      Fq ch;
      try {
        t.append_point("Cy", cm2[0].data());
        t.append_scalars("a", a);
        t.append_point("delta", delta.data());
        t.append_point("beta", cm2[1].data());
        ch = t.challenge_scalar("c");
      } catch (...) { if (pending && !rs.active) (void)sp_sumcheck_bind_eval_collect(c, ev); rs.drain(); throw; }
      dpp_early = true; ch_early = ch;
      if (pending && rs.active) rs.bind_eval_collect(kind, ev);
      else if (pending) SPX(sp_sumcheck_bind_eval_collect(c, ev));
      if (resharded && more) { SPX(sp_sumcheck_eval(c, kind, tabs.data(), tabs.size(), ev)); pending = true; }
      if (more) {
        REQUIRE(pending);  // a further round means the tables had >= 4 entries: the next evaluations came with the bind
        next_poly = make_poly(ev, eval);
        FqVec row3(W, fq_zero());
        poly_row(row3, 0, next_poly, j + 1);
        const sp_host_point* add3[1] = {&ahead.wait(j + 1).bp_hn};
        cm2.push_back(commit_rows(row3, 1, add3)[0]);
      }
    } else {
      size_t nrows2 = more ? 3 : 2;
      FqVec rows2(nrows2 * W, fq_zero());
      rows2[nn + 1] = target; rows2[nn + 2] = blind;
      rows2[W + nn + 1] = dp; rows2[W + nn + 2] = r_beta;
      if (more) {
        next_poly = make_poly(ev, eval);
        poly_row(rows2, 2, next_poly, j + 1);
      }
      cm2 = msm_rows(c, gn.g, idx_u, rows2, nrows2);
    }
    Fq ch;
    if (dpp_early) ch = ch_early;
    else {
      t.append_point("Cy", cm2[0].data());
      t.append_scalars("a", a);
      t.append_point("delta", delta.data());
      t.append_point("beta", cm2[1].data());
      ch = t.challenge_scalar("c");
    }
    DotProductProof dpp;
    dpp.delta = delta; dpp.beta = cm2[1];
    dpp.z.resize(nn);
    for (size_t i = 0; i < nn; i++) dpp.z[i] = ch * poly.coeffs[i] + d[i];
    dpp.z_delta = ch * blinds_poly[j] + r_delta;
    dpp.z_beta = ch * blind + r_beta;
    claim_per_round = eval;
    comm_claim_per_round = comm_eval;
    out.proofs.push_back(dpp);
    out.comm_evals.push_back(comm_eval);
    r.push_back(r_j);
    if (more) { poly = next_poly; comm_poly = cm2[2]; }
  }
  final_claims->resize(tabs.size());
  SPX(sp_table_heads(c, tabs.data(), tabs.size(), U(*final_claims)));
  *r_out = r;
  *blind_post = blinds_evals[num_rounds - 1];
  return out;
}

// ------------------------------------------------------------------ DotProductProofLog (nizk/mod.rs:440-525) + bullet.rs:32-132
// x lives on the device (it is the bound polynomial LZ of PolyEvalProof::prove): it is committed to and handed to the
// inner-product argument from there
static DotProductProofLog dotproductlog_prove(sp_ctx* c, const DotProductProofGens& gens, Transcript& t, RandomTape& tape, const DevTable& x,
                                              const Fq& blind_x, const FqVec& a, const Fq& y, const Fq& blind_y, CP* Cy_out) { HSPAN("dotproductlog_prove");
  t.append_protocol_name("dot product proof (log)");
  size_t n = x.len();
  REQUIRE(a.size() == n && gens.n == n);
  Fq d = tape.random_scalar("d");
  Fq r_delta = tape.random_scalar("r_delta");
  Fq r_beta = tape.random_scalar("r_delta");  // sic: the reference draws r_beta under the label "r_delta" (nizk/mod.rs:459)
  size_t lg_n = log_2(n);
  FqVec v1 = tape.random_vector("blinds_vec_1", lg_n), v2 = tape.random_vector("blinds_vec_2", lg_n);
  const MultiCommitGens &gn = gens.gens_n, &g1 = gens.gens_1;
  for (size_t i = 0; i < gn.G.size(); i++) REQUIRE(gn.G[i] == gn.G[0] + i);  // one contiguous run of the generator stream
  // Cx = commit(x, blind_x) and the argument's device state from one copy of x; BulletReductionProof::prove runs with
  // Q = r*G1 (gens_1.scale(r), nizk/mod.rs:479-480) and H = h, r being drawn below
  sp_ipa* ipa = nullptr;
  CP Cx;
  SPX(sp_ipa_begin_dev(c, gn.g, gn.G[0], n, g1.G[0], gn.h, x.h, U(a), U(blind_x), Cx.data(), &ipa));
  // the first round's kernel needs neither r nor the blinds: in flight while this core absorbs Cx, Cy and `a` (no device call in between:
  // Cy comes from the host-side engine)
  if (small_msm_on_host(c) && n >= 2 && sp_ipa_round_prelaunch(ipa) != SP_OK) { sp_ipa_free(ipa); throw Error("sp_ipa_round_prelaunch failed"); }
  DotProductProofLog p;
  Fq blind_hat, r;
  CP Cy;
  try {
    t.append_point("Cx", Cx.data());
    Cy = commit_scalar(c, y, blind_y, g1);
    t.append_point("Cy", Cy.data());
    t.append_scalars("a", a);
    r = t.challenge_scalar("r");
    SPX(sp_ipa_set_scale(ipa, U(r)));
    blind_hat = blind_x + r * blind_y;  // blind_Gamma
    for (size_t k = 0; k < lg_n; k++) {
      CP L, R;
      SPX(sp_ipa_round_lr(ipa, U(v1[k]), U(v2[k]), L.data(), R.data()));
      Fq u, u_inv;
      { HSPAN("ipa_transcript");
      t.append_point("L", L.data());
      t.append_point("R", R.data());
      u = t.challenge_scalar("u"); }
      { HSPAN("ipa_invert");
      // u is a public challenge: division steps (fq_inv.hpp, ~1 us) instead of the a^(q-2) chain (~6 us), same value (checked against
      // the chain in tests/test_host_arith.py)
      u_inv = fq_invert_vartime(u); }
      SPX(sp_ipa_round_fold(ipa, U(u), U(u_inv)));
      blind_hat = blind_hat + v1[k] * u * u + v2[k] * u_inv * u_inv;
      p.bullet.L_vec.push_back(L);
      p.bullet.R_vec.push_back(R);
    }
    Fq a_hat, b_hat;
    CP delta;
    SPX(sp_ipa_finish_commit(ipa, U(d), U(r_delta), a_hat.l, b_hat.l, delta.data()));  // a_hat, b_hat and commit(d, r_delta) under {g_hat, h}: one trip
    Fq y_hat = a_hat * b_hat;
    t.append_point("delta", delta.data());
    CP beta = msm_rows(c, g1.g, {g1.G[0], g1.h}, {d * r, r_beta}, 1)[0];  // commit(d, r_beta) under gens_1.scale(r)
    t.append_point("beta", beta.data());
    Fq ch = t.challenge_scalar("c");
    p.delta = delta; p.beta = beta;
    p.z1 = d + ch * y_hat;
    p.z2 = b_hat * (ch * blind_hat + r_beta) + r_delta;
  } catch (...) {
    sp_ipa_free(ipa);
    throw;
  }
  sp_ipa_free(ipa);
  if (Cy_out) *Cy_out = Cy;
  return p;
}

// PolyEvalProof::prove (dense_mlpoly.rs:312-365)
// Zr_from_LZ != nullptr: the evaluation itself is not known yet and is computed HERE, as <LZ, R> — the same field element as
// DensePolynomial::evaluate(r) = sum_i Z_i chi_i(r) = L^T Z R, from the vector-matrix product the opening needs anyway — and returned through
// it (round 6: R1CSProof::prove's `poly_vars.evaluate(&ry[1..])`, r1csproof.rs:299, was a pass of its own over the witness, 0.13 ms at 2^20
// and 0.6 ms at 2^22 on the critical path; the dot product of two sqrt(N) vectors is one launch-sized trip). `Zr` is ignored then.
static PolyEvalProof polyeval_prove(sp_ctx* c, const DevTable& poly, const FqVec* blinds_opt, const FqVec& r, const Fq& Zr, const Fq* blind_Zr_opt,
                                    const PolyCommitmentGens& gens, Transcript& t, RandomTape& tape, CP* C_Zr, Fq* Zr_from_LZ = nullptr) { HSPAN("polyeval_prove");
  t.append_protocol_name("polynomial evaluation proof");
  size_t Ls = pow2(r.size() / 2), Rs = pow2(r.size() - r.size() / 2);
  REQUIRE(poly.len() == Ls * Rs);
  std::vector<sp_ctx*> shards = residue_shard_ctxs(c);
  const bool shard_bound = shards.size() >= 2 && Ls % shards.size() == 0 && !commit_shard_residue_off(c);
  FqVec Lv;  // the chi vector of the left half on the host: only the blinded opening and the sharded product need it here
  if (blinds_opt || shard_bound) Lv = eq_evals_host(FqVec(r.begin(), r.begin() + r.size() / 2));
  sp_table* lz = nullptr;
  if (shard_bound) {
    // DensePolynomial::bound sharded by row blocks (the same contiguous rows a sharded commitment gives each shard):
    // shard g multiplies rows [g Ls/W, (g+1) Ls/W) by its slice of L; the W partial vectors (R scalars each) are added in F_q
    const size_t W = shards.size(), per = Ls / W;
    SPX(sp_ctx_sync(c));
    std::vector<DevTable> views, parts;
    for (size_t g = 0; g < W; g++) {
      views.push_back(tab_view(shards[g], poly, g * per * Rs, per * Rs));
      sp_table* pz = nullptr;
      SPX(sp_vecmat_dev(shards[g], U(Lv) + 4 * g * per, per, views.back().h, &pz));
      parts.emplace_back(shards[g], pz);
    }
    for (size_t g = 1; g < W; g++) SPX(sp_ctx_sync(shards[g]));
    for (size_t g = 1; g < W; g++) SPX(sp_table_add_into(c, parts[0].h, parts[g].h));
    SPX(sp_ctx_sync(c));  // the partial vectors go back to their contexts' pools below
    commit_shard_note_gather(c, 32 * Rs * W);
    lz = parts[0].h;
    parts[0].h = nullptr;
  } else if (!blinds_opt) {
    // no blinds: L is needed for this product only — generated and consumed on the device (the host copy above is not used)
    DevTable Lt = tab_eq(c, FqVec(r.begin(), r.begin() + r.size() / 2));
    SPX(sp_vecmat_tab(c, Lt.h, poly.h, &lz));
  } else {
    SPX(sp_vecmat_dev(c, U(Lv), Ls, poly.h, &lz));  // DensePolynomial::bound :349, kept on the device; queued, not waited for
  }
  DevTable LZ(c, lz);
  Fq Zr_val = Zr;
  DevTable Rt;
  if (Zr_from_LZ) Rt = tab_eq(c, FqVec(r.begin() + r.size() / 2, r.end()));  // queued behind the product; the host computes its own copy meanwhile
  FqVec Rv = eq_evals_host(FqVec(r.begin() + r.size() / 2, r.end()));  // while the device multiplies
  if (Zr_from_LZ) {
    SPX(sp_dot(c, LZ.h, 0, Rt.h, 0, Rs, Zr_val.l));
    *Zr_from_LZ = Zr_val;
  }
  Fq LZ_blind = fq_zero();
  if (blinds_opt) {
    REQUIRE(blinds_opt->size() == Ls);
    LZ_blind = dot_host(*blinds_opt, Lv);
  }
  Fq blind_Zr = blind_Zr_opt ? *blind_Zr_opt : fq_zero();
  PolyEvalProof p;
  p.proof = dotproductlog_prove(c, gens.gens, t, tape, LZ, LZ_blind, Rv, Zr_val, blind_Zr, C_Zr);
  return p;
}

// ------------------------------------------------------------------ R1CSProof::prove (r1csproof.rs:144-349)
// on_rx / on_ry: called as soon as the first / second sum-check has fixed rx / ry (SNARK::prove starts work that only depends
// on them there)
static R1CSProof r1cs_prove(sp_ctx* c, const Instance& inst, const Fq* vars, size_t nvars_given, const FqVec& input, const R1CSGens& gens,
                            Transcript& t, RandomTape& tape, FqVec* rx_out, FqVec* ry_out, ProveTimes* tm,
                            const std::function<void(const FqVec&)>* on_rx = nullptr,
                            const std::function<void()>* transcript_prefix = nullptr,
                            const std::function<void(const FqVec&)>* on_ry = nullptr, const sp_table* vars_resident = nullptr) { HSPAN("r1cs_prove");
  double t0 = now_s();
  // lib.rs:360-368 / 519-526: the assignment is zero-padded to the instance's (padded) num_vars — done in the device table
  REQUIRE(nvars_given <= inst.num_vars && input.size() < inst.num_vars && input.size() == inst.num_inputs);
  R1CSProof P;
  size_t num_vars = inst.num_vars, lv = log_2(num_vars);
  // polycommit (:160-171). The witness commitment does not depend on the transcript: it is started first, and everything
  // the transcript has to absorb BEFORE it (the caller's prefix — for a SNARK the 6144 shares of the computation
  // commitment — and the inputs) is hashed while the GPU computes; the order of the transcript operations is the reference's.
  DevTable poly_vars = tab_alloc(c, num_vars);  // zero-filled: implicit padding
  if (vars_resident) {  // a VarsAssignment: device-to-device
    REQUIRE(sp_table_len(vars_resident) == nvars_given);
    if (nvars_given) SPX(sp_table_copy(c, poly_vars.h, 0, vars_resident, 0, nvars_given));
  }
  FqVec blinds_vars = tape.random_vector("poly_blinds", pow2(lv / 2));
  {
    size_t Ls = pow2(lv / 2), Rs = pow2(lv - lv / 2);
    const MultiCommitGens& g = gens.gens_pc.gens.gens_n;
    sp_job* job = nullptr;
    bool async = Ls > 8 && !commit_shard_active(c) && g.n() == Rs;
    // a host assignment of full length: the upload and the commitment in one call (the additions of a chunk of rows run while the
    // next chunk crosses PCIe); otherwise upload, then commit
    const bool upload_commit = async && !vars_resident && nvars_given == num_vars;
    if (!vars_resident && nvars_given && !upload_commit) SPX(sp_table_write(c, poly_vars.h, 0, vars[0].l, nvars_given));
    // A copy out of pageable memory keeps the calling thread busy until the bytes have left the caller's buffer (0.75 ms for 32 MB), and
    // the transcript prefix (the computation commitment: 0.65 ms of Keccak at 2^20) needs no device: the upload + commit is issued by a
    // helper thread while this one hashes; the context is not touched here before the join. Option upload.thread = 0: one thread (A/B).
    const bool upload_thread = ctx_opt(c, "upload.thread") != 0;
    std::thread uploader;
    int32_t up_rc = SP_OK;
    if (upload_commit && upload_thread && transcript_prefix)
      uploader = std::thread([&] { up_rc = sp_commit_rows_upload_start(c, g.g, g.G[0], g.h, poly_vars.h, 0, vars[0].l, Ls, Rs, U(blinds_vars), &job); });
    else if (upload_commit) SPX(sp_commit_rows_upload_start(c, g.g, g.G[0], g.h, poly_vars.h, 0, vars[0].l, Ls, Rs, U(blinds_vars), &job));
    else if (async) SPX(sp_commit_rows_dev_start(c, g.g, g.G[0], g.h, poly_vars.h, 0, Ls, Rs, U(blinds_vars), &job));
    try {
      if (transcript_prefix) (*transcript_prefix)();
      t.append_protocol_name("R1CS proof");
      t.append_scalars("input", input);
    } catch (...) {
      if (uploader.joinable()) uploader.join();
      if (job) { std::vector<uint8_t> sink(32 * Ls); (void)sp_job_wait(job, sink.data()); }
      throw;
    }
    if (uploader.joinable()) { uploader.join(); SPX(up_rc); }
    if (async) {
      std::vector<uint8_t> out(32 * Ls);
      SPX(sp_job_wait(job, out.data()));
      P.comm_vars.C.resize(Ls);
      for (size_t i = 0; i < Ls; i++) P.comm_vars.C[i] = to_cp(&out[32 * i]);
    } else {
      P.comm_vars = poly_commit(c, poly_vars, lv, gens.gens_pc, &blinds_vars);
    }
  }
  // z = vars | 1 | input | 0...  (:177-185) and Az, Bz, Cz (:187-196) do not depend on the transcript: they are queued now and
  // the device builds them while this core absorbs the commitment
  DevTable z = tab_alloc(c, 2 * num_vars);
  SPX(sp_table_copy(c, z.h, 0, poly_vars.h, 0, num_vars));
  {
    FqVec tail;
    tail.push_back(fq_one());
    tail.insert(tail.end(), input.begin(), input.end());
    SPX(sp_table_write(c, z.h, num_vars, U(tail), tail.size()));
  }
  sp_table *tAz, *tBz, *tCz;
  SPX(sp_sparse_mulvec(c, inst.dA, z.h, &tAz));
  DevTable poly_Az(c, tAz);
  SPX(sp_sparse_mulvec(c, inst.dB, z.h, &tBz));
  DevTable poly_Bz(c, tBz);
  SPX(sp_sparse_mulvec(c, inst.dC, z.h, &tCz));
  DevTable poly_Cz(c, tCz);
  append_poly_commitment(t, "poly_commitment", P.comm_vars);
  if (tm) tm->polycommit = now_s() - t0;

  double t1 = now_s();
  size_t num_rounds_x = log_2(inst.num_cons), num_rounds_y = log_2(2 * num_vars);
  FqVec tau = t.challenge_vector("challenge_tau", num_rounds_x);
  DevTable poly_tau = tab_eq(c, tau);
  FqVec rx, claims1;
  Fq blind_claim_postsc1;
  P.sc_proof_phase1 = zk_sumcheck_prove(c, 2, fq_zero(), fq_zero(), num_rounds_x, {poly_tau.h, poly_Az.h, poly_Bz.h, poly_Cz.h},
                                        gens.gens_sc.gens_1, gens.gens_sc.gens_4, t, tape, &rx, &claims1, &blind_claim_postsc1);
  if (tm) tm->sc_phase_one = now_s() - t1;
  if (on_rx) (*on_rx)(rx);
  const Fq &tau_claim = claims1[0], &Az_claim = claims1[1], &Bz_claim = claims1[2], &Cz_claim = claims1[3];
  Fq Az_blind = tape.random_scalar("Az_blind"), Bz_blind = tape.random_scalar("Bz_blind");
  Fq Cz_blind = tape.random_scalar("Cz_blind"), prod_Az_Bz_blind = tape.random_scalar("prod_Az_Bz_blind");
  CP comm_Cz, comm_Az, comm_Bz, comm_prod;
  P.pok_claims_phase2 = knowledge_prove(c, gens.gens_sc.gens_1, t, tape, Cz_claim, Cz_blind, &comm_Cz);
  Fq prod = Az_claim * Bz_claim;
  P.proof_prod = product_prove(c, gens.gens_sc.gens_1, t, tape, Az_claim, Az_blind, Bz_claim, Bz_blind, prod, prod_Az_Bz_blind, &comm_Az, &comm_Bz,
                               &comm_prod);
  t.append_point("comm_Az_claim", comm_Az.data());
  t.append_point("comm_Bz_claim", comm_Bz.data());
  t.append_point("comm_Cz_claim", comm_Cz.data());
  t.append_point("comm_prod_Az_Bz_claims", comm_prod.data());
  P.claims_phase2[0] = comm_Az; P.claims_phase2[1] = comm_Bz; P.claims_phase2[2] = comm_Cz; P.claims_phase2[3] = comm_prod;
  Fq blind_expected_claim_postsc1 = tau_claim * (prod_Az_Bz_blind - Cz_blind);
  Fq claim_post_phase1 = (Az_claim * Bz_claim - Cz_claim) * tau_claim;
  P.proof_eq_sc_phase1 = equality_prove(c, gens.gens_sc.gens_1, t, tape, claim_post_phase1, blind_expected_claim_postsc1, claim_post_phase1,
                                        blind_claim_postsc1);

  double t2 = now_s();
  Fq r_A = t.challenge_scalar("challenge_Az"), r_B = t.challenge_scalar("challenge_Bz"), r_C = t.challenge_scalar("challenge_Cz");
  Fq claim_phase2 = r_A * Az_claim + r_B * Bz_claim + r_C * Cz_claim;
  Fq blind_claim_phase2 = r_A * Az_blind + r_B * Bz_blind + r_C * Cz_blind;
  DevTable poly_ABC;
  {
    DevTable evals_rx = tab_eq(c, rx);  // :274
    const sp_sparse* ms[3] = {inst.dA, inst.dB, inst.dC};
    FqVec w = {r_A, r_B, r_C};
    sp_table* tt;
    SPX(sp_sparse_eval_table(c, ms, U(w), 3, evals_rx.h, &tt));  // :275-283
    poly_ABC = DevTable(c, tt);
  }
  FqVec ry, claims2;
  Fq blind_claim_postsc2;
  P.sc_proof_phase2 = zk_sumcheck_prove(c, 0, claim_phase2, blind_claim_phase2, num_rounds_y, {z.h, poly_ABC.h}, gens.gens_sc.gens_1,
                                        gens.gens_sc.gens_3, t, tape, &ry, &claims2, &blind_claim_postsc2);
  if (tm) tm->sc_phase_two = now_s() - t2;
  if (on_ry) (*on_ry)(ry);

  double t3 = now_s();
  FqVec ry1(ry.begin() + 1, ry.end());
  Fq eval_vars_at_ry;
  bool eval_from_opening = false;  // unsharded: the evaluation comes out of the opening's own vector-matrix product (polyeval_prove)
  {
    std::vector<sp_ctx*> shards = residue_shard_ctxs(c);
    size_t W = shards.size(), lw = W >= 2 ? log_2(W) : 0;
    if (W < 2 && ctx_opt(c, "polyeval.eval_from_opening") != 0) {
      eval_from_opening = true;
    } else if (W >= 2 && ry1.size() > lw + 1 && !commit_shard_residue_off(c)) {
      // DensePolynomial::evaluate as W partial dot products over contiguous chunks (chunk g = the top log2 W index
      // bits): <Z, chi(r)> = sum_g chi_g(r[..lw]) <Z_g, chi(r[lw..])>, one scalar per shard gathered and combined here
      SPX(sp_ctx_sync(c));
      FqVec top = eq_evals_host(FqVec(ry1.begin(), ry1.begin() + lw)), low(ry1.begin() + lw, ry1.end());
      size_t chunk = poly_vars.len() / W;
      std::vector<DevTable> views;
      eval_vars_at_ry = fq_zero();
      for (size_t g = 0; g < W; g++) {
        views.push_back(tab_view(shards[g], poly_vars, g * chunk, chunk));
        Fq e;
        SPX(sp_evaluate(shards[g], views.back().h, U(low), low.size(), e.l));
        eval_vars_at_ry += top[g] * e;
      }
      commit_shard_note_gather(c, 32 * W);
    } else {
      SPX(sp_evaluate(c, poly_vars.h, U(ry1), ry1.size(), eval_vars_at_ry.l));  // :299
    }
  }
  Fq blind_eval = tape.random_scalar("blind_eval");
  P.proof_eval_vars_at_ry = polyeval_prove(c, poly_vars, &blinds_vars, ry1, eval_vars_at_ry, &blind_eval, gens.gens_pc, t, tape, &P.comm_vars_at_ry,
                                           eval_from_opening ? &eval_vars_at_ry : nullptr);
  if (tm) tm->polyeval = now_s() - t3;

  Fq blind_eval_Z_at_ry = (fq_one() - ry[0]) * blind_eval;
  Fq blind_expected_claim_postsc2 = claims2[1] * blind_eval_Z_at_ry;
  Fq claim_post_phase2 = claims2[0] * claims2[1];
  P.proof_eq_sc_phase2 = equality_prove(c, gens.gens_pc.gens.gens_1, t, tape, claim_post_phase2, blind_expected_claim_postsc2, claim_post_phase2,
                                        blind_claim_postsc2);
  *rx_out = rx;
  *ry_out = ry;
  if (tm) tm->r1cs_sat = now_s() - t0;
  return P;
}

VarsAssignment::VarsAssignment(Ctx& ctx, const Fq* vars, size_t n_) : c(ctx.h), n(n_) {
  REQUIRE(vars && n_ > 0);
  sp_table* t = nullptr;
  SPX(sp_table_upload(c, vars[0].l, n_, &t));
  tab = DevTable(c, t);
}
NIZK NIZK::prove(Ctx& ctx, const Instance& inst, const Fq* vars, size_t nvars_given, const FqVec& inputs, const NIZKGens& gens, Transcript& t,
                 const Fq* tape_seed, ProveTimes* tm, const sp_table* vars_resident) {  // lib.rs:501-546
  double t0 = now_s();
  const std::vector<uint8_t> digest = inst.compute_digest();  // lib.rs:514 absorbs inst.digest (r1cs.rs:154-158); computed once per instance, under its lock
  Fq shared_seed;  // lock-step ranks of a sharded proof must share one tape: rank 0 draws it (shard.cc)
  if (!tape_seed && commit_shard_shared_seed(ctx.h, &shared_seed)) tape_seed = &shared_seed;
  RandomTape tape = tape_seed ? RandomTape("proof", *tape_seed) : RandomTape("proof");  // random.rs:11-18
  std::function<void()> prefix = [&]() {
    t.append_protocol_name("Spartan NIZK proof");
    t.append_message("R1CSShapeDigest", digest.data(), digest.size());
  };
  NIZK P;
  P.r1cs_sat_proof = r1cs_prove(ctx.h, inst, vars, nvars_given, inputs, gens.gens_r1cs_sat, t, tape, &P.rx, &P.ry, tm, nullptr, &prefix, nullptr, vars_resident);
  if (tm) tm->total = now_s() - t0;
  return P;
}

#include "spark.inc"

// ------------------------------------------------------------------ bincode 1.3 (fixed-width LE ints, u64 lengths)
namespace {
struct Wr {
  std::vector<uint8_t> b;
  void u64(uint64_t x) { for (int i = 0; i < 8; i++) b.push_back((uint8_t)(x >> (8 * i))); }
  void fq(const Fq& x) { for (int i = 0; i < 4; i++) u64(x.l[i]); }  // Scalar serializes its raw Montgomery limbs (ristretto255.rs:198-199)
  void cp(const CP& c) { b.insert(b.end(), c.begin(), c.end()); }
  void fqv(const FqVec& v) { u64(v.size()); for (auto& x : v) fq(x); }
  void cpv(const std::vector<CP>& v) { u64(v.size()); for (auto& x : v) cp(x); }
};
void w_dpp(Wr& w, const DotProductProof& p) { w.cp(p.delta); w.cp(p.beta); w.fqv(p.z); w.fq(p.z_delta); w.fq(p.z_beta); }
void w_zksc(Wr& w, const ZKSumcheckInstanceProof& p) { w.cpv(p.comm_polys); w.cpv(p.comm_evals); w.u64(p.proofs.size()); for (auto& d : p.proofs) w_dpp(w, d); }
void w_eq(Wr& w, const EqualityProof& p) { w.cp(p.alpha); w.fq(p.z); }
void w_pe(Wr& w, const PolyEvalProof& p) {
  w.cpv(p.proof.bullet.L_vec); w.cpv(p.proof.bullet.R_vec); w.cp(p.proof.delta); w.cp(p.proof.beta); w.fq(p.proof.z1); w.fq(p.proof.z2);
}
void w_r1cs(Wr& w, const R1CSProof& p) {
  w.cpv(p.comm_vars.C);
  w_zksc(w, p.sc_proof_phase1);
  for (int i = 0; i < 4; i++) w.cp(p.claims_phase2[i]);
  w.cp(p.pok_claims_phase2.alpha); w.fq(p.pok_claims_phase2.z1); w.fq(p.pok_claims_phase2.z2);
  w.cp(p.proof_prod.alpha); w.cp(p.proof_prod.beta); w.cp(p.proof_prod.delta);
  for (int i = 0; i < 5; i++) w.fq(p.proof_prod.z[i]);
  w_eq(w, p.proof_eq_sc_phase1);
  w_zksc(w, p.sc_proof_phase2);
  w.cp(p.comm_vars_at_ry);
  w_pe(w, p.proof_eval_vars_at_ry);
  w_eq(w, p.proof_eq_sc_phase2);
}
void w_batched(Wr& w, const ProductCircuitEvalProofBatched& p) {
  w.u64(p.proof.size());
  for (auto& l : p.proof) {
    w.u64(l.proof.compressed_polys.size());
    for (auto& c : l.proof.compressed_polys) w.fqv(c);
    w.fqv(l.claims_prod_left); w.fqv(l.claims_prod_right);
  }
  for (int i = 0; i < 3; i++) w.fqv(p.claims_dotp[i]);
}
void w_evalproof(Wr& w, const SparseMatPolyEvalProof& p) {
  w.cpv(p.comm_derefs.C);
  const ProductLayerProof& L = p.proof_prod_layer;
  w.fq(L.row_init); w.fqv(L.row_read); w.fqv(L.row_write); w.fq(L.row_audit);
  w.fq(L.col_init); w.fqv(L.col_read); w.fqv(L.col_write); w.fq(L.col_audit);
  w.fqv(L.eval_val[0]); w.fqv(L.eval_val[1]);
  w_batched(w, L.proof_mem); w_batched(w, L.proof_ops);
  const HashLayerProof& h = p.proof_hash_layer;
  w.fqv(h.row_addr); w.fqv(h.row_read_ts); w.fq(h.row_audit_ts);
  w.fqv(h.col_addr); w.fqv(h.col_read_ts); w.fq(h.col_audit_ts);
  w.fqv(h.eval_val); w.fqv(h.eval_derefs[0]); w.fqv(h.eval_derefs[1]);
  w_pe(w, h.proof_ops); w_pe(w, h.proof_mem); w_pe(w, h.proof_derefs);
}
}  // namespace
namespace {
void w_mcg(Wr& w, const MultiCommitGens& g, const std::vector<uint8_t>& comp) {  // commitments.rs:7-12
  w.u64(g.n());
  w.u64(g.n());
  for (uint32_t i : g.G) w.b.insert(w.b.end(), comp.begin() + 32 * i, comp.begin() + 32 * i + 32);
  w.b.insert(w.b.end(), comp.begin() + 32 * g.h, comp.begin() + 32 * g.h + 32);
}
void w_pcg(Wr& w, const PolyCommitmentGens& g, const std::vector<uint8_t>& comp) {  // dense_mlpoly.rs:24-27, nizk/mod.rs:407-412
  w.u64(g.gens.n);
  w_mcg(w, g.gens.gens_n, comp);
  w_mcg(w, g.gens.gens_1, comp);
}
}  // namespace
std::vector<uint8_t> SNARKGens::serialize() const {
  Wr w;
  const std::vector<uint8_t>& cs = stream_sat.compressed;
  w_mcg(w, gens_r1cs_sat.gens_sc.gens_1, cs);  // R1CSSumcheckGens (r1csproof.rs:39-44)
  w_mcg(w, gens_r1cs_sat.gens_sc.gens_3, cs);
  w_mcg(w, gens_r1cs_sat.gens_sc.gens_4, cs);
  w_pcg(w, gens_r1cs_sat.gens_pc, cs);         // R1CSGens.gens_pc (:61-65)
  const std::vector<uint8_t>& ce = stream_eval.compressed;
  w_pcg(w, gens_r1cs_eval.gens_ops, ce);       // SparseMatPolyCommitmentGens (sparse_mlpoly.rs:284-289)
  w_pcg(w, gens_r1cs_eval.gens_mem, ce);
  w_pcg(w, gens_r1cs_eval.gens_derefs, ce);
  return w.b;
}
std::vector<uint8_t> ComputationCommitment::serialize() const {
  Wr w;
  w.u64(num_cons); w.u64(num_vars); w.u64(num_inputs);
  w.u64(comm.batch_size); w.u64(comm.num_ops); w.u64(comm.num_mem_cells);
  w.cpv(comm.comm_comb_ops.C);
  w.cpv(comm.comm_comb_mem.C);
  return w.b;
}
std::vector<uint8_t> ComputationDecommitment::serialize() const {
  Wr w;
  const MultiSparseMatPolynomialAsDense& d = dense;
  auto poly = [&](const DevTable& t) {  // DensePolynomial {num_vars, len, Z}
    size_t n = t.len();
    FqVec z(n);
    SPX(sp_table_download(t.c, t.h, 0, n, U(z)));
    w.u64(log_2(n)); w.u64(n); w.fqv(z);
  };
  auto polys = [&](const std::vector<DevTable>& v) { w.u64(v.size()); for (auto& t : v) poly(t); };
  auto at = [&](const AddrTimestamps& a) {
    // ops_addr_usize: Vec<Vec<usize>> — the same numbers as ops_addr (DensePolynomial::from_usize, sparse_mlpoly.rs:246-248), read back from it
    w.u64(a.ops_addr.size());
    for (auto& t : a.ops_addr) {
      size_t n = t.len();
      FqVec z(n);
      SPX(sp_table_download(t.c, t.h, 0, n, U(z)));
      w.u64(n);
      for (auto& x : z) w.u64(sp::fq_from_mont(x).l[0]);
    }
    polys(a.ops_addr); polys(a.read_ts); poly(a.audit_ts);
  };
  w.u64(d.batch_size); polys(d.val); at(d.row); at(d.col); poly(d.comb_ops); poly(d.comb_mem);
  return w.b;
}
std::vector<uint8_t> serialize_r1cs_proof(const R1CSProof& p) { Wr w; w_r1cs(w, p); return w.b; }
std::vector<uint8_t> NIZK::serialize() const { Wr w; w_r1cs(w, r1cs_sat_proof); w.fqv(rx); w.fqv(ry); return w.b; }
std::vector<uint8_t> SNARK::serialize() const {
  Wr w;
  w_r1cs(w, r1cs_sat_proof);
  for (int i = 0; i < 3; i++) w.fq(inst_evals[i]);
  w_evalproof(w, r1cs_eval_proof);
  return w.b;
}

}  // namespace spz
