// spartan_amd host driver: the 1..8-term commitments of the Sigma protocols, on the proving thread's core.
//
// Every round of the two zero-knowledge sum-checks (sumcheck.rs:428-776) and every Sigma protocol of nizk/mod.rs commits to
// two..five scalars under fixed generators (Scalar::commit / UniPoly::commit, commitments.rs:73-93) and needs the encoded point
// before the transcript can move on. That is a chain of ~100 dependent mixed additions followed by one inverse square root:
// a lone wavefront runs it at ~1 us per F_p multiplication (measured, bench/ubench_fpmul: one multiplication in flight per
// SIMD), the host core that is waiting for the answer anyway at ~15 ns. So these commitments are not sent to the GPU: they
// are computed here, from signed 10-bit window tables of the handful of generators involved (same layout and recoding as the
// device tables, msm.hpp), with the same point arithmetic the kernels compile (curve.hpp, host instantiation).
// The device path for them still exists (sp_msm_indexed) and is selected by SPARTAN_SMALL_MSM=device: the proofs are
// byte-identical either way (tests/test_gpu_proofs.py), the latency is not ("small commitments").
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <vector>
#include "libspartan.hpp"
#include "../csrc/msm.hpp"

namespace spz {
using namespace sp;

namespace {
constexpr int kHostWbits = 10;  // 26 windows x 512 entries x 96 B = 1.25 MiB per generator: the ~8 generators in use stay in the core's L3 slice

struct StreamTables {
  std::vector<uint8_t> compressed;  // the stream's points (a copy of GensStream::compressed: the stream object may move)
  std::mutex mu;
  std::map<uint32_t, std::unique_ptr<Niels[]>> tab;
};
std::mutex g_reg_mu;
std::map<const sp_gens*, std::shared_ptr<StreamTables>> g_reg;
int g_mode = -1;  // -1: read SPARTAN_SMALL_MSM on first use; 0: device; 1: host

// window table of one point: entry (w, m) = m * 2^(c w) * P for m = 1..2^(c-1), affine Niels form (msm_tidx layout, pt = 0)
std::unique_ptr<Niels[]> build_table(const uint8_t comp[32]) {
  const MsmGeom g = msm_geom(kHostWbits);
  Pt base;
  if (!pt_decompress(comp, &base)) throw Error("small_msm: generator does not decode");
  std::vector<Pt> e(g.pt_entries);
  for (int w = 0; w < g.nwin; w++) {
    Pt acc = base;
    for (int m = 1; m <= g.tent; m++) {
      e[msm_tidx(g, 0, w, m)] = acc;
      if (m < g.tent) acc = pt_add(acc, base);
    }
    // next window's base: 2^c * base = 2 * (2^(c-1) * base), the last entry
    base = pt_dbl(acc);
  }
  // one inversion for all Z (Montgomery's trick)
  std::vector<Fp> pre(g.pt_entries);
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

const Niels* table_of(StreamTables& st, uint32_t idx) {
  std::lock_guard<std::mutex> lk(st.mu);
  auto it = st.tab.find(idx);
  if (it != st.tab.end()) return it->second.get();
  if ((size_t)idx * 32 + 32 > st.compressed.size()) throw Error("small_msm: generator index out of range");
  auto t = build_table(st.compressed.data() + 32 * (size_t)idx);
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
}  // namespace

void small_msm_set_mode(int mode) { g_mode = mode; }
bool small_msm_on_host() {
  if (g_mode < 0) {
    const char* e = getenv("SPARTAN_SMALL_MSM");
    g_mode = (e && strcmp(e, "device") == 0) ? 0 : 1;
  }
  return g_mode == 1;
}
void small_msm_register(const sp_gens* g, const std::vector<uint8_t>& compressed) {
  std::lock_guard<std::mutex> lk(g_reg_mu);
  auto st = std::make_shared<StreamTables>();
  st->compressed = compressed;
  g_reg[g] = std::move(st);
}
bool small_msm_has(const sp_gens* g) {
  std::lock_guard<std::mutex> lk(g_reg_mu);
  return g_reg.count(g) != 0;
}
void small_msm_forget(const sp_gens* g) {
  std::lock_guard<std::mutex> lk(g_reg_mu);
  g_reg.erase(g);
}
// rows x cols scalars (Montgomery form, row-major) over generators idx[0..cols) of the stream behind g -> rows encoded points.
// false: g is not a registered stream (the caller uses the device path).
bool small_msm_rows(const sp_gens* g, const uint32_t* idx, size_t cols, const Fq* scalars, size_t rows, uint8_t* out) {
  std::shared_ptr<StreamTables> st;
  {
    std::lock_guard<std::mutex> lk(g_reg_mu);
    auto it = g_reg.find(g);
    if (it == g_reg.end()) return false;
    st = it->second;
  }
  const Niels* tabs[16];
  if (cols > 16) return false;
  for (size_t k = 0; k < cols; k++) tabs[k] = table_of(*st, idx[k]);
  for (size_t r = 0; r < rows; r++) {
    Pt acc = pt_identity();
    for (size_t k = 0; k < cols; k++) accumulate(acc, scalars[r * cols + k], tabs[k]);
    pt_compress(acc, out + 32 * r);
  }
  return true;
}
static_assert(sizeof(HostPt) == sizeof(Pt), "HostPt is Pt");
static std::shared_ptr<StreamTables> stream_of(const sp_gens* g) {
  std::lock_guard<std::mutex> lk(g_reg_mu);
  auto it = g_reg.find(g);
  return it == g_reg.end() ? nullptr : it->second;
}
bool small_msm_point(const sp_gens* g, const uint32_t* idx, size_t cols, const Fq* scalars, HostPt* out) {
  auto st = stream_of(g);
  if (!st || cols > 16) return false;
  Pt acc = pt_identity();
  for (size_t k = 0; k < cols; k++)
    if (!fq_is_zero(scalars[k])) accumulate(acc, scalars[k], table_of(*st, idx[k]));
  memcpy(out, &acc, sizeof(Pt));
  return true;
}
bool small_msm_rows_plus(const sp_gens* g, const uint32_t* idx, size_t cols, const Fq* scalars, size_t rows, const HostPt* const* addend, uint8_t* out) {
  auto st = stream_of(g);
  if (!st || cols > 16) return false;
  const Niels* tabs[16];
  for (size_t k = 0; k < cols; k++) tabs[k] = nullptr;
  for (size_t r = 0; r < rows; r++) {
    Pt acc = pt_identity();
    for (size_t k = 0; k < cols; k++) {
      const Fq& sc = scalars[r * cols + k];
      if (fq_is_zero(sc)) continue;
      if (!tabs[k]) tabs[k] = table_of(*st, idx[k]);
      accumulate(acc, sc, tabs[k]);
    }
    if (addend && addend[r]) {
      Pt a;
      memcpy(&a, addend[r], sizeof(Pt));
      acc = pt_add(acc, a);
    }
    pt_compress(acc, out + 32 * r);
  }
  return true;
}
// test hook (no device): commitments of rows x npts scalars under npts compressed points
int small_msm_probe(const uint8_t* compressed, size_t npts, const uint64_t* scalars, size_t rows, uint8_t* out) {
  if (npts == 0 || npts > 16) return SP_EINVAL;
  try {
    std::vector<std::unique_ptr<Niels[]>> own;
    const Niels* tabs[16];
    for (size_t k = 0; k < npts; k++) { own.push_back(build_table(compressed + 32 * k)); tabs[k] = own.back().get(); }
    for (size_t r = 0; r < rows; r++) {
      Pt acc = pt_identity();
      for (size_t k = 0; k < npts; k++) {
        Fq s;
        memcpy(s.l, scalars + 4 * (r * npts + k), 32);
        accumulate(acc, s, tabs[k]);
      }
      pt_compress(acc, out + 32 * r);
    }
  } catch (const Error&) {
    return SP_EPOINT;
  }
  return SP_OK;
}

}  // namespace spz
