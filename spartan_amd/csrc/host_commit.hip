// spartan_amd: the 1..16-term commitments of the Sigma protocols, on the CALLING THREAD's core (host code; no kernel here).
//
// Every round of the two zero-knowledge sum-checks (sumcheck.rs:428-776) and every Sigma protocol of nizk/mod.rs commits to
// two..five scalars under fixed generators (Scalar::commit / UniPoly::commit, commitments.rs:73-93) and needs the encoded point
// before the transcript can move on. That is a chain of ~100 dependent mixed additions followed by one inverse square root:
// a lone wavefront runs it at ~1 us per F_p multiplication (measured, bench/ubench_fpmul: one multiplication in flight per
// SIMD), the host core that is waiting for the answer anyway at ~15 ns. So these commitments are not sent to the GPU: they
// are computed here, from signed 10-bit window tables of the handful of generators involved (same layout and recoding as the
// device tables, msm.hpp), with the same point arithmetic the kernels compile (curve.hpp, host instantiation).
// They are part of the C ABI (sp_host_*): the Rust crate with `--features gpu` calls the same code the C++ driver calls
// (rust_shim/seams/sumcheck.rs, nizk.rs) instead of falling back to dalek for them.
// The device path for them still exists (sp_msm_indexed); the proofs are byte-identical either way
// (tests/test_gpu_proofs.py), the latency is not ("small commitments").
#include <atomic>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>

#include "internal.hpp"

namespace {
constexpr int kHostWbits = 10;  // 26 windows x 512 entries x 96 B = 1.25 MiB per generator: the ~8 generators in use stay in the core's L3 slice

struct StreamTables {
  std::mutex mu;
  std::map<uint32_t, std::unique_ptr<Niels[]>> tab;
};
std::mutex g_reg_mu;
std::map<const void*, std::shared_ptr<StreamTables>> g_reg;  // keyed by the generator set's cache entry (core.hip)

// window table of one point: entry (w, m) = m * 2^(c w) * P for m = 1..2^(c-1), affine Niels form (msm_tidx layout, pt = 0)
std::unique_ptr<Niels[]> build_table(const uint8_t comp[32]) {
  const MsmGeom g = msm_geom(kHostWbits);
  Pt base;
  if (!pt_decompress(comp, &base)) return nullptr;
  std::vector<Pt> e(g.pt_entries);
  for (int w = 0; w < g.nwin; w++) {
    Pt acc = base;
    for (int m = 1; m <= g.tent; m++) {
      e[msm_tidx(g, 0, w, m)] = acc;
      if (m < g.tent) acc = pt_add(acc, base);
    }
    base = pt_dbl(acc);  // next window's base: 2^c * base = 2 * (2^(c-1) * base), the last entry
  }
  std::vector<Fp> pre(g.pt_entries);  // one inversion for all Z (Montgomery's trick)
  Fp run = fp_one();
  for (size_t i = 0; i < g.pt_entries; i++) { pre[i] = run; run = fp_mul(run, e[i].Z); }
  Fp inv = fp_invert(run);
  std::unique_ptr<Niels[]> t(new Niels[g.pt_entries]);
  for (size_t i = g.pt_entries; i-- > 0;) {
    Fp zinv = fp_mul(inv, pre[i]);
    inv = fp_mul(inv, e[i].Z);
    t[i] = pt_to_niels(e[i], zinv);
  }
  return t;
}

std::shared_ptr<StreamTables> stream_of(const sp_gens* g) {
  std::lock_guard<std::mutex> lk(g_reg_mu);
  auto& p = g_reg[g->cache_entry];
  if (!p) p = std::make_shared<StreamTables>();
  return p;
}
// nullptr: index out of range or the point does not decode (cannot happen for a set the device accepted)
const Niels* table_of(const sp_gens* g, StreamTables& st, uint32_t idx) {
  std::lock_guard<std::mutex> lk(st.mu);
  auto it = st.tab.find(idx);
  if (it != st.tab.end()) return it->second.get();
  size_t n = 0;
  const uint8_t* comp = gens_compressed_bytes(g, &n);
  if (!comp || (size_t)idx >= n) return nullptr;
  auto t = build_table(comp + 32 * (size_t)idx);
  if (!t) return nullptr;
  const Niels* p = t.get();
  st.tab.emplace(idx, std::move(t));
  return p;
}

// acc += s * P, P given by its table. The signed digits are produced first and their entries requested from the cache
// hierarchy together, then the additions run.
inline void accumulate(Pt& acc, const Fq& s_mont, const Niels* t) {
  if (fq_is_zero(s_mont)) return;
  const MsmGeom g = msm_geom(kHostWbits);
  const Fq s = fq_from_mont(s_mont);
  int dig[32];
  int carry = 0;
  for (int w = 0; w < g.nwin; w++) {
    int d = (int)msm_field(s, w, g.wbits) + carry;
    carry = d >= g.tent;
    d -= carry << g.wbits;
    dig[w] = d;
    if (d) {
      const Niels* p = t + (size_t)w * g.tent + (size_t)((d < 0 ? -d : d) - 1);
      __builtin_prefetch(p);
      __builtin_prefetch((const char*)p + 64);
    }
  }
  for (int w = 0; w < g.nwin; w++) {
    int d = dig[w];
    if (d) acc = pt_madd(acc, t[(size_t)w * g.tent + (size_t)((d < 0 ? -d : d) - 1)], d < 0);
  }
}
inline Fq limbs_at(const uint64_t* p, size_t i) {
  Fq x;
  memcpy(x.l, p + 4 * i, 32);
  return x;
}
static_assert(sizeof(sp_host_point) == sizeof(Pt), "sp_host_point is an extended point");

int32_t row_point(const sp_gens* g, StreamTables& st, const Niels** tabs, const uint32_t* idx, size_t cols, const uint64_t* S, Pt* out) {
  Pt acc = pt_identity();
  for (size_t k = 0; k < cols; k++) {
    Fq sc = limbs_at(S, k);
    if (fq_is_zero(sc)) continue;
    if (!tabs[k] && !(tabs[k] = table_of(g, st, idx[k]))) return SP_EINVAL;
    accumulate(acc, sc, tabs[k]);
  }
  *out = acc;
  return SP_OK;
}
}  // namespace

void host_commit_forget(const void* cache_entry) {
  std::lock_guard<std::mutex> lk(g_reg_mu);
  g_reg.erase(cache_entry);
}

// The tape-only halves of a ZK sum-check's commitments, computed ahead of the rounds by a helper thread while the proving
// thread is in its first evaluation (see spartan_hip.h). Completion is published per round; a waiter spins briefly (the
// helper is ahead of the rounds after the first one) and then sleeps on a condition variable.
struct sp_zk_ahead {
  const sp_gens* g;
  std::vector<uint32_t> idx_u;
  size_t W, nn, rounds;
  std::vector<uint64_t> blinds_poly, blinds_evals, d, r_delta, r_beta;
  std::vector<uint8_t> delta;          // 32 * rounds
  std::vector<Pt> be_h, rb_h, bp_hn;   // rounds each
  std::atomic<size_t> done{0};
  std::atomic<int32_t> failed{0};
  std::mutex mu;
  std::condition_variable cv;
  std::thread th;
};

extern "C" {

int32_t sp_host_commit_small(const sp_gens* g, const uint32_t* idx, size_t cols, const uint64_t* S, size_t rows, const sp_host_point* const* addend,
                             uint8_t* out) {
  if (!g || !idx || !S || !out || cols == 0 || cols > 16 || rows == 0) return SP_EINVAL;
  auto st = stream_of(g);
  const Niels* tabs[16] = {nullptr};
  Pt accs[8];  // encoded together, up to four at a time (pt_compress_many: the inverse-square-root chains of a call's rows overlap)
  for (size_t r0 = 0; r0 < rows; r0 += 8) {
    const size_t nr = rows - r0 < 8 ? rows - r0 : 8;
    for (size_t r = r0; r < r0 + nr; r++) {
      Pt acc;
      SPCHK(row_point(g, *st, tabs, idx, cols, S + 4 * r * cols, &acc));
      if (addend && addend[r]) {
        Pt a;
        memcpy(&a, addend[r], sizeof(Pt));
        acc = pt_add(acc, a);
      }
      accs[r - r0] = acc;
    }
    pt_compress_many(accs, nr, out + 32 * r0);
  }
  return SP_OK;
}
int32_t sp_host_commit_point(const sp_gens* g, const uint32_t* idx, size_t cols, const uint64_t* S, sp_host_point* out) {
  if (!g || !idx || !S || !out || cols == 0 || cols > 16) return SP_EINVAL;
  auto st = stream_of(g);
  const Niels* tabs[16] = {nullptr};
  Pt acc;
  SPCHK(row_point(g, *st, tabs, idx, cols, S, &acc));
  memcpy(out, &acc, sizeof(Pt));
  return SP_OK;
}
// pts[set * rows + r]: adds the `nsets` points of every row and encodes the sums (RFC 9496) — the combine step of a column-sharded
// commitment (sp_commit_rows_partial: one set per shard, one more for the blind terms), all on the calling thread.
int32_t sp_host_points_sum_encode(const sp_host_point* pts, size_t nsets, size_t rows, uint8_t* out) {
  if (!pts || !out || nsets == 0 || rows == 0) return SP_EINVAL;
  Pt accs[8];
  for (size_t r0 = 0; r0 < rows; r0 += 8) {
    const size_t nr = rows - r0 < 8 ? rows - r0 : 8;
    for (size_t r = r0; r < r0 + nr; r++) {
      Pt acc;
      memcpy(&acc, &pts[r], sizeof(Pt));
      for (size_t k = 1; k < nsets; k++) {
        Pt p;
        memcpy(&p, &pts[k * rows + r], sizeof(Pt));
        acc = pt_add(acc, p);
      }
      accs[r - r0] = acc;
    }
    pt_compress_many(accs, nr, out + 32 * r0);
  }
  return SP_OK;
}
// Device-free form for tests: npts compressed points, rows x npts scalars.
int32_t sp_host_commit_probe(const uint8_t* compressed, size_t npts, const uint64_t* S, size_t rows, uint8_t* out) {
  if (!compressed || !S || !out || npts == 0 || npts > 16) return SP_EINVAL;
  std::vector<std::unique_ptr<Niels[]>> own;
  const Niels* tabs[16];
  for (size_t k = 0; k < npts; k++) {
    own.push_back(build_table(compressed + 32 * k));
    if (!own.back()) return SP_EPOINT;
    tabs[k] = own.back().get();
  }
  for (size_t r = 0; r < rows; r++) {
    Pt acc = pt_identity();
    for (size_t k = 0; k < npts; k++) accumulate(acc, limbs_at(S, r * npts + k), tabs[k]);
    pt_compress(acc, out + 32 * r);
  }
  return SP_OK;
}

int32_t sp_host_zk_ahead_begin(const sp_gens* g, const uint32_t* idx_u, size_t W, size_t nn, size_t rounds, const uint64_t* blinds_poly,
                               const uint64_t* blinds_evals, const uint64_t* d, const uint64_t* r_delta, const uint64_t* r_beta, sp_zk_ahead** out) {
  if (!g || !idx_u || !blinds_poly || !blinds_evals || !d || !r_delta || !r_beta || !out || rounds == 0 || nn == 0 || W != nn + 3 || W > 16) return SP_EINVAL;
  sp_zk_ahead* a = new (std::nothrow) sp_zk_ahead();
  if (!a) return SP_ENOMEM;
  a->g = g; a->W = W; a->nn = nn; a->rounds = rounds;
  a->idx_u.assign(idx_u, idx_u + W);
  a->blinds_poly.assign(blinds_poly, blinds_poly + 4 * rounds);
  a->blinds_evals.assign(blinds_evals, blinds_evals + 4 * rounds);
  a->d.assign(d, d + 4 * rounds * nn);
  a->r_delta.assign(r_delta, r_delta + 4 * rounds);
  a->r_beta.assign(r_beta, r_beta + 4 * rounds);
  a->delta.resize(32 * rounds);
  a->be_h.resize(rounds); a->rb_h.resize(rounds); a->bp_hn.resize(rounds);
  try {
    a->th = std::thread([a]() {
      auto st = stream_of(a->g);
      const Niels* tabs[16] = {nullptr};
      const size_t nn = a->nn, W = a->W;
      const uint32_t ihn = a->idx_u[nn], ih1 = a->idx_u[nn + 2];
      const Niels *thn = table_of(a->g, *st, ihn), *th1 = table_of(a->g, *st, ih1);
      int32_t rc = (thn && th1) ? SP_OK : SP_EINVAL;
      std::vector<uint64_t> row(4 * W);
      for (size_t j = 0; j < a->rounds && rc == SP_OK; j++) {
        Pt p = pt_identity();
        accumulate(p, limbs_at(a->blinds_poly.data(), j), thn);
        a->bp_hn[j] = p;
        p = pt_identity();
        accumulate(p, limbs_at(a->blinds_evals.data(), j), th1);
        a->be_h[j] = p;
        std::fill(row.begin(), row.end(), 0);  // delta = <d_j, G_n> + r_delta_j * h_n: nothing but the tape enters it
        memcpy(row.data(), a->d.data() + 4 * j * nn, 32 * nn);
        memcpy(row.data() + 4 * nn, a->r_delta.data() + 4 * j, 32);
        rc = row_point(a->g, *st, tabs, a->idx_u.data(), W, row.data(), &p);
        if (rc != SP_OK) break;
        pt_compress(p, a->delta.data() + 32 * j);
        p = pt_identity();
        accumulate(p, limbs_at(a->r_beta.data(), j), th1);
        a->rb_h[j] = p;
        {
          std::lock_guard<std::mutex> lk(a->mu);
          a->done.store(j + 1, std::memory_order_release);
        }
        a->cv.notify_all();
      }
      if (rc != SP_OK) {
        std::lock_guard<std::mutex> lk(a->mu);
        a->failed.store(rc, std::memory_order_release);
        a->cv.notify_all();
      }
    });
  } catch (...) {
    delete a;
    return SP_ENOMEM;
  }
  *out = a;
  return SP_OK;
}
int32_t sp_host_zk_ahead_wait(sp_zk_ahead* a, size_t j, uint8_t delta[32], sp_host_point* be_h, sp_host_point* rb_h, sp_host_point* bp_hn) {
  if (!a || j >= a->rounds) return SP_EINVAL;
  for (int spins = 0; spins < 20000 && a->done.load(std::memory_order_acquire) <= j && !a->failed.load(std::memory_order_acquire); spins++) {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#else
    std::this_thread::yield();
#endif
  }
  if (a->done.load(std::memory_order_acquire) <= j) {  // the helper is behind (an oversubscribed host): sleep instead of burning the core it needs
    std::unique_lock<std::mutex> lk(a->mu);
    a->cv.wait(lk, [&] { return a->done.load(std::memory_order_acquire) > j || a->failed.load(std::memory_order_acquire) != 0; });
  }
  if (a->done.load(std::memory_order_acquire) <= j) return a->failed.load(std::memory_order_acquire);
  if (delta) memcpy(delta, a->delta.data() + 32 * j, 32);
  if (be_h) memcpy(be_h, &a->be_h[j], sizeof(Pt));
  if (rb_h) memcpy(rb_h, &a->rb_h[j], sizeof(Pt));
  if (bp_hn) memcpy(bp_hn, &a->bp_hn[j], sizeof(Pt));
  return SP_OK;
}
void sp_host_zk_ahead_free(sp_zk_ahead* a) {
  if (!a) return;
  if (a->th.joinable()) a->th.join();
  delete a;
}

}  // extern "C"
