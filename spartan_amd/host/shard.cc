// spartan_amd host driver: row-sharded commitments across the GPUs of one node — inside the library.
//
// Every rank runs the same proof in lock-step (same instance, tape seed and transcript). For a DensePolynomial::commit
// (dense_mlpoly.rs:179-204) of L rows, rank r computes rows [r L/W, (r+1) L/W) on its GPU; the 32-byte compressed
// commitments are exchanged with ONE all-gather of bytes per commitment. Rows are independent MSMs over shared generators,
// so there is no elliptic-curve reduction to do (RCCL has no such operator; "all-reduce of partial bucket sums" would be
// an all-gather plus local point additions, and is only needed when a single row is too wide for one GPU).
// Three transports behind one code path (partition, blind offsets, result layout are shared):
//   rccl     ncclAllGather on device buffers, on the context's stream. librccl.so is dlopen'ed on first use: a single-GPU
//            deployment, and the CPU-only test box, never load it.
//   callback the caller moves the bytes (tests: torch.distributed over gloo)
//   virtual  W shards on ONE physical GPU: W sub-contexts (own streams, shared generator tables) compute their row ranges
//            concurrently and the "gather" is a memcpy — the partitioning logic under test without a second GPU
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>

#include <chrono>
#include <map>
#include <mutex>

#include "libspartan.hpp"

namespace spz {
namespace {

// ---- the slice of RCCL's API this file uses (rccl/rccl.h), bound at run time
typedef struct { char internal[128]; } ncclUniqueId_t;
typedef void* ncclComm_p;
typedef int (*fn_ncclGetUniqueId)(ncclUniqueId_t*);
typedef int (*fn_ncclCommInitRank)(ncclComm_p*, int, ncclUniqueId_t, int);
typedef int (*fn_ncclAllGather)(const void*, void*, size_t, int /*ncclDataType_t: ncclUint8 = 1*/, ncclComm_p, hipStream_t);
typedef int (*fn_ncclCommDestroy)(ncclComm_p);
typedef const char* (*fn_ncclGetErrorString)(int);
struct Rccl {
  void* lib = nullptr;
  fn_ncclGetUniqueId GetUniqueId = nullptr;
  fn_ncclCommInitRank CommInitRank = nullptr;
  fn_ncclAllGather AllGather = nullptr;
  fn_ncclCommDestroy CommDestroy = nullptr;
  fn_ncclGetErrorString GetErrorString = nullptr;
};
Rccl& rccl() {
  static Rccl r = [] {
    Rccl x;
    for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
      x.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (x.lib) break;
    }
    if (x.lib) {
      x.GetUniqueId = (fn_ncclGetUniqueId)dlsym(x.lib, "ncclGetUniqueId");
      x.CommInitRank = (fn_ncclCommInitRank)dlsym(x.lib, "ncclCommInitRank");
      x.AllGather = (fn_ncclAllGather)dlsym(x.lib, "ncclAllGather");
      x.CommDestroy = (fn_ncclCommDestroy)dlsym(x.lib, "ncclCommDestroy");
      x.GetErrorString = (fn_ncclGetErrorString)dlsym(x.lib, "ncclGetErrorString");
      if (!x.GetUniqueId || !x.CommInitRank || !x.AllGather || !x.CommDestroy) { dlclose(x.lib); x.lib = nullptr; }
    }
    return x;
  }();
  return r;
}
void need_rccl() {
  if (!rccl().lib) throw Error("librccl.so could not be loaded: RCCL sharding is unavailable on this machine");
}
void nccl_ok(int rc, const char* what) {
  if (rc != 0) throw Error(std::string(what) + " failed: " + (rccl().GetErrorString ? rccl().GetErrorString(rc) : "nccl error") + " (" + std::to_string(rc) + ")");
}
void hip_ok(hipError_t e, const char* what) {
  if (e != hipSuccess) throw Error(std::string(what) + " failed: " + hipGetErrorString(e));
}

struct ShardState {
  int mode = 0;  // 0 none, 1 callback, 2 rccl, 3 virtual
  int rank = 0, world = 1;
  int dev = 0;  // the GPU of the owning context: every HIP/RCCL object of this state is created with it current
  CommitGatherFn gather = nullptr;
  void* user = nullptr;
  ncclComm_p comm = nullptr;
  hipStream_t stream = nullptr;  // rccl: the collective's stream
  uint8_t* dbuf = nullptr;       // rccl: device staging [send | recv]
  size_t dbuf_bytes = 0;
  std::vector<sp_ctx*> vctx;     // virtual: sub-contexts 1..W-1 (shard 0 runs on the owning context)
  ShardStats stats;
  // Switches that decide whether a rank ENTERS a collective. They are read from the environment ONCE, when the sharding is configured,
  // kept here, and — for the multi-process transports — compared across the ranks before the first proof (a rank whose environment
  // differs would otherwise deadlock in ncclAllGather or diverge from the others' transcripts).
  bool no_shard_cols = false;    // option shard.cols = 0: few-row commitments are not column-sharded
  bool no_residue = false;       // option shard.residues = 0: sum-check tables are never residue-sharded
  bool device_encode = false;    // option encode.device: sp_commit_rows_partial is unavailable -> no column sharding
  int residue_min_log2 = 22;     // multi-process transports: tables of >= 2^k entries are residue-sharded (option shard.residue_min_log2; DESIGN.md section 6:
                                 // a round must outlast the ~26 us exchange, i.e. >= 2^22 entries per table; 0 = always, 64 = never)
};
std::mutex g_mu;
std::map<sp_ctx*, ShardState> g_state;

// the sharding-related options of the context (options.hpp), resolved ONCE when the sharding is configured and compared across the ranks
void read_switches(sp_ctx* c, ShardState& s) {
  s.no_shard_cols = !ctx_opt(c, "shard.cols");
  s.no_residue = !ctx_opt(c, "shard.residues");
  s.device_encode = ctx_opt(c, "encode.device") != 0;
  s.residue_min_log2 = (int)ctx_opt(c, "shard.residue_min_log2");
  if (ctx_opt(c, "shard.residue_transport")) s.residue_min_log2 = 0;  // the round-3 opt-in: every sum-check, whatever its size
}
void release(ShardState& s) {
  if (s.comm || s.dbuf || s.stream) (void)hipSetDevice(s.dev);
  for (sp_ctx* v : s.vctx) sp_ctx_destroy(v);
  s.vctx.clear();
  if (s.comm && rccl().CommDestroy) (void)rccl().CommDestroy(s.comm);
  s.comm = nullptr;
  if (s.dbuf) (void)hipFree(s.dbuf);
  s.dbuf = nullptr;
  if (s.stream) (void)hipStreamDestroy(s.stream);
  s.stream = nullptr;
}

void gather_bytes(ShardState& s, uint8_t* all, size_t per);
// every lock-step rank must have resolved the switches the same way: one 8-byte exchange when the sharding is configured
void check_switches_agree(ShardState& s) {
  if (s.world <= 1) return;
  const size_t W = (size_t)s.world;
  std::vector<uint8_t> all(8 * W, 0);
  uint8_t* mine = &all[8 * (size_t)s.rank];
  mine[0] = s.no_shard_cols; mine[1] = s.no_residue; mine[2] = s.device_encode; mine[3] = (uint8_t)s.residue_min_log2; mine[4] = 0xA5;
  gather_bytes(s, all.data(), 8);
  for (size_t r = 0; r < W; r++)
    if (memcmp(&all[8 * r], mine, 8) != 0)
      throw Error("set_commit_shard: rank " + std::to_string(r) + " resolved the sharding options (shard.cols / shard.residues / "
                  "encode.device / shard.residue_min_log2) differently from rank " + std::to_string(s.rank) + ": the ranks would not enter the same collectives");
}

}  // namespace

bool commit_shard_active(sp_ctx* c) {
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_state.find(c);
  return it != g_state.end() && it->second.mode != 0 && it->second.world > 1;
}
void commit_shard_forget(sp_ctx* c) {
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_state.find(c);
  if (it == g_state.end()) return;
  release(it->second);
  g_state.erase(it);
}
void set_commit_shard(Ctx& c, int rank, int world, CommitGatherFn gather, void* user) {
  commit_shard_forget(c.h);
  if (world <= 1) return;
  if (!gather || rank < 0 || rank >= world) throw Error("set_commit_shard: bad arguments");
  ShardState s;
  s.mode = 1; s.rank = rank; s.world = world; s.gather = gather; s.user = user;
  read_switches(c.h, s);
  check_switches_agree(s);
  std::lock_guard<std::mutex> lk(g_mu);
  g_state[c.h] = s;
}
void rccl_unique_id(uint8_t out[128]) {
  need_rccl();
  ncclUniqueId_t id;
  nccl_ok(rccl().GetUniqueId(&id), "ncclGetUniqueId");
  memcpy(out, id.internal, 128);
}
void set_commit_shard_rccl(Ctx& c, int rank, int world, const uint8_t unique_id[128]) {
  commit_shard_forget(c.h);
  if (world < 1 || rank < 0 || rank >= world || !unique_id) throw Error("set_commit_shard_rccl: bad arguments");
  need_rccl();
  ShardState s;
  s.mode = 2; s.rank = rank; s.world = world;
  s.dev = sp_ctx_device(c.h);
  ncclUniqueId_t id;
  memcpy(id.internal, unique_id, 128);
  hip_ok(hipSetDevice(s.dev), "hipSetDevice");  // the communicator and its stream belong to the context's GPU, whatever the calling thread had current
  hip_ok(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking), "hipStreamCreate");
  read_switches(c.h, s);
  try {
    nccl_ok(rccl().CommInitRank(&s.comm, world, id, rank), "ncclCommInitRank");
    check_switches_agree(s);
  } catch (...) {
    release(s);
    throw;
  }
  std::lock_guard<std::mutex> lk(g_mu);
  g_state[c.h] = s;
}
void set_commit_shard_virtual(Ctx& c, int nshards) {
  commit_shard_forget(c.h);
  if (nshards <= 1) return;
  ShardState s;
  s.mode = 3; s.rank = 0; s.world = nshards;
  read_switches(c.h, s);
  int dev = s.dev = sp_ctx_device(c.h);
  for (int k = 1; k < nshards; k++) {
    sp_ctx* v = nullptr;
    if (sp_ctx_create(dev, &v) != SP_OK) { release(s); throw Error("set_commit_shard_virtual: sp_ctx_create failed"); }
    (void)sp_ctx_copy_options(v, c.h);  // a virtual shard runs with its parent's settings
    s.vctx.push_back(v);
  }
  std::lock_guard<std::mutex> lk(g_mu);
  g_state[c.h] = s;
}
ShardStats commit_shard_stats(Ctx& c, bool reset) {
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_state.find(c.h);
  if (it == g_state.end()) return ShardStats();
  ShardStats r = it->second.stats;
  if (reset) it->second.stats = ShardStats();
  return r;
}

// A RandomTape seed shared by the lock-step ranks of a sharded proof. With the production setting (no caller seed) every
// rank would seed its tape from its own OS entropy: each would blind its row slice with its own tape, the gathered
// commitment would mix blinds of different tapes and the ranks' transcripts would diverge. So the ranks agree on one seed:
// EVERY rank contributes 64 bytes of OS entropy, the contributions travel over the transport that moves the commitments
// (callback: one gather; RCCL: one all-gather), and the seed is from_bytes_wide(SHAKE256("spartan_amd shared tape seed" || all)):
// no single rank (and no transport slot left unfilled) can fix it, and an all-zero or short gather is refused.
// CONTRACT of the transport (set_commit_shard's callback, the RCCL communicator): the seed material fixes every blind of the proof —
// whoever reads it can recover the witness from the proof — so the transport must be confidential and node-local (xGMI / shared
// memory between the ranks of one node; never a network hop). The buffers are wiped after use.
// Returns false when no multi-rank transport is configured (single GPU, virtual shards): the caller seeds as usual.
bool commit_shard_shared_seed(sp_ctx* c, Fq* seed) {
  ShardState* sp = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_state.find(c);
    if (it == g_state.end() || (it->second.mode != 1 && it->second.mode != 2) || it->second.world <= 1) return false;
    sp = &it->second;
  }
  ShardState& s = *sp;
  const size_t W = (size_t)s.world, per = 64;
  std::vector<uint8_t> buf(per * W, 0);
  {
    Fq a = RandomTape::os_random_scalar(), b = RandomTape::os_random_scalar();
    memcpy(&buf[per * (size_t)s.rank], a.l, 32);
    memcpy(&buf[per * (size_t)s.rank + 32], b.l, 32);
    volatile uint64_t* wa = a.l; volatile uint64_t* wb = b.l;
    for (int i = 0; i < 4; i++) { wa[i] = 0; wb[i] = 0; }
  }
  gather_bytes(s, buf.data(), per);
  bool ok = true;
  for (size_t r = 0; r < W && ok; r++) {
    bool zero = true;
    for (size_t k = 0; k < per; k++) zero = zero && buf[per * r + k] == 0;
    ok = !zero;  // a slot nobody filled: a transport that returned success without moving that rank's bytes
  }
  if (ok) {
    Shake256 sh;
    static const char dom[] = "spartan_amd shared tape seed";
    sh.absorb(dom, sizeof dom - 1);
    sh.absorb(buf.data(), buf.size());
    uint8_t wide[64];
    sh.squeeze(wide, 64);
    uint64_t w[8];
    memcpy(w, wide, 64);
    *seed = sp::fq_from_u512(w);
    volatile uint8_t* vw = wide; for (size_t i = 0; i < 64; i++) vw[i] = 0;
    volatile uint64_t* v8 = w; for (int i = 0; i < 8; i++) v8[i] = 0;
  }
  volatile uint8_t* vb = buf.data(); for (size_t i = 0; i < buf.size(); i++) vb[i] = 0;
  if (s.mode == 2 && s.dbuf) (void)hipMemset(s.dbuf, 0, s.dbuf_bytes < per * (W + 1) ? s.dbuf_bytes : per * (W + 1));
  if (!ok) throw Error("commit shard transport returned an unfilled tape-seed slot: the ranks cannot agree on a RandomTape seed");
  s.stats.gathers++; s.stats.bytes += per * W;
  return true;
}

// Cost of one small all-gather on the context's RCCL communicator (bench/shard_probe.py: the per-round exchange of the residue-sharded
// sum-checks is 96 bytes per rank): `iters` times H2D of `bytes`, ncclAllGather, D2H, stream sync — what sharded_commit_rows does per
// commitment. Returns microseconds per exchange, or < 0 when no RCCL transport is configured.
double rccl_allgather_probe(sp_ctx* c, size_t bytes, int iters) {
  ShardState* sp = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_state.find(c);
    if (it == g_state.end() || it->second.mode != 2) return -1.0;
    sp = &it->second;
  }
  ShardState& s = *sp;
  size_t W = (size_t)s.world, need = bytes + bytes * W;
  hip_ok(hipSetDevice(s.dev), "hipSetDevice");
  if (s.dbuf_bytes < need) {
    if (s.dbuf) hip_ok(hipFree(s.dbuf), "hipFree");
    s.dbuf = nullptr;
    hip_ok(hipMalloc((void**)&s.dbuf, need), "hipMalloc");
    s.dbuf_bytes = need;
  }
  std::vector<uint8_t> host(need, 1);
  auto once = [&]() {
    hip_ok(hipMemcpyAsync(s.dbuf, host.data(), bytes, hipMemcpyHostToDevice, s.stream), "hipMemcpyAsync");
    nccl_ok(rccl().AllGather(s.dbuf, s.dbuf + bytes, bytes, 1, s.comm, s.stream), "ncclAllGather");
    hip_ok(hipMemcpyAsync(host.data(), s.dbuf + bytes, bytes * W, hipMemcpyDeviceToHost, s.stream), "hipMemcpyAsync");
    hip_ok(hipStreamSynchronize(s.stream), "hipStreamSynchronize");
  };
  for (int i = 0; i < 20; i++) once();
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < iters; i++) once();
  return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / iters * 1e6;
}

// The sub-contexts of a virtual sharding (shard 0 = the owning context), for the residue-sharded sum-checks, bound and evaluate of
// prover.cc; empty when no virtual sharding is configured. (With one process per GPU the same partition runs with the partial sums
// travelling over the gather transport; virtual shards exercise the partition and its arithmetic on one GPU.)
std::vector<sp_ctx*> residue_shard_ctxs(sp_ctx* c) {
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_state.find(c);
  std::vector<sp_ctx*> v;
  if (it == g_state.end() || it->second.mode != 3 || it->second.world <= 1) return v;
  v.push_back(c);
  for (sp_ctx* x : it->second.vctx) v.push_back(x);
  return v;
}
void commit_shard_note_gather(sp_ctx* c, size_t bytes) {
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_state.find(c);
  if (it == g_state.end()) return;
  it->second.stats.gathers++;
  it->second.stats.bytes += bytes;
}

// all-gather of `per` bytes from every rank into all[world * per] (rank order) over the context's transport (callback or RCCL)
namespace {
void gather_bytes(ShardState& s, uint8_t* all, size_t per) {
  const size_t W = (size_t)s.world, lo = per * (size_t)s.rank;
  if (s.mode == 1) {
    if (s.gather(s.user, all, per * W, lo, per) != 0) throw Error("commit shard gather failed");
    return;
  }
  size_t need = per + per * W;
  hip_ok(hipSetDevice(s.dev), "hipSetDevice");
  if (s.dbuf_bytes < need) {
    if (s.dbuf) hip_ok(hipFree(s.dbuf), "hipFree");
    s.dbuf = nullptr;
    hip_ok(hipMalloc((void**)&s.dbuf, need), "hipMalloc");
    s.dbuf_bytes = need;
  }
  hip_ok(hipMemcpyAsync(s.dbuf, all + lo, per, hipMemcpyHostToDevice, s.stream), "hipMemcpyAsync");
  nccl_ok(rccl().AllGather(s.dbuf, s.dbuf + per, per, 1 /*ncclUint8*/, s.comm, s.stream), "ncclAllGather");
  hip_ok(hipMemcpyAsync(all, s.dbuf + per, per * W, hipMemcpyDeviceToHost, s.stream), "hipMemcpyAsync");
  hip_ok(hipStreamSynchronize(s.stream), "hipStreamSynchronize");
}
}  // namespace

// The multi-process transports (callback, RCCL) as seen by the residue-sharded sum-checks of prover.cc: this rank's place among the
// lock-step ranks, and an all-gather of `per` bytes per rank (all[world * per], rank order; this rank's slice is filled in by the caller).
bool commit_shard_transport(sp_ctx* c, int* rank, int* world, int* residue_min_log2) {
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_state.find(c);
  if (it == g_state.end() || (it->second.mode != 1 && it->second.mode != 2) || it->second.world <= 1) return false;
  *rank = it->second.rank; *world = it->second.world;
  if (residue_min_log2) *residue_min_log2 = it->second.no_residue ? 64 : it->second.residue_min_log2;
  return true;
}
bool commit_shard_residue_off(sp_ctx* c) {
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_state.find(c);
  return it != g_state.end() && it->second.no_residue;
}
void commit_shard_gather(sp_ctx* c, uint8_t* all, size_t per) {
  ShardState* sp = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_state.find(c);
    if (it == g_state.end() || (it->second.mode != 1 && it->second.mode != 2)) throw Error("commit_shard_gather: no multi-process transport");
    sp = &it->second;
  }
  gather_bytes(*sp, all, per);
  sp->stats.gathers++; sp->stats.bytes += per * (size_t)sp->world;
}

// A commitment with fewer rows than shards (Ls <= 8: a small instance's witness, a single-row commit) sharded by COLUMNS — SURVEY 8e's
// rendering of the north-star's "partial bucket sums": shard k sums the generators [k Rs/W, (k+1) Rs/W) of every row into one partial
// point per row (sp_commit_rows_partial), the W x Ls points (128 bytes each) are gathered — RCCL has no elliptic-curve reduction, so
// the "all-reduce" is an all-gather and a local addition — and every rank adds them, adds the blind terms and encodes
// (sp_host_points_sum_encode): the same points as the unsharded sum, hence the same bytes. Off with option shard.cols = 0.
static bool sharded_commit_cols(ShardState& s, sp_ctx* c, const sp_gens* g, size_t g_off, size_t h_idx, const sp_table* Z, size_t Ls, size_t Rs,
                                const uint64_t* blinds, uint8_t* out /*32*Ls*/) {
  const size_t W = (size_t)s.world;
  // (device_encode: sp_commit_rows_partial leaves row sums as points for the HOST to add and encode, which that diagnostic setting forbids:
  // the commitment then takes the unsharded path on every rank, as it did before column sharding existed)
  if (W <= 1 || Ls == 0 || Ls > 8 || Rs % W != 0 || s.no_shard_cols || s.device_encode) return false;
  const size_t per = Rs / W;
  auto chk = [](int32_t rc, const char* what) { if (rc != SP_OK) throw Error(std::string(what) + " failed: " + sp_strerror(rc)); };
  std::vector<sp_host_point> pts((W + 1) * Ls);  // [shard][row], then the blind terms
  if (s.mode == 3) {
    chk(sp_ctx_sync(c), "sp_ctx_sync");
    for (size_t k = 0; k < W; k++) {
      sp_ctx* ck = k == 0 ? c : s.vctx[k - 1];
      chk(sp_commit_rows_partial(ck, g, g_off + k * per, Z, k * per, Rs, Ls, per, &pts[k * Ls]), "sp_commit_rows_partial");
    }
  } else {
    const size_t k = (size_t)s.rank;
    chk(sp_commit_rows_partial(c, g, g_off + k * per, Z, k * per, Rs, Ls, per, &pts[k * Ls]), "sp_commit_rows_partial");
    gather_bytes(s, (uint8_t*)pts.data(), sizeof(sp_host_point) * Ls);
  }
  size_t nsets = W;
  if (blinds) {
    const uint32_t hh[1] = {(uint32_t)h_idx};
    for (size_t r = 0; r < Ls; r++) chk(sp_host_commit_point(g, hh, 1, blinds + 4 * r, &pts[W * Ls + r]), "sp_host_commit_point");
    nsets = W + 1;
  }
  chk(sp_host_points_sum_encode(pts.data(), nsets, Ls, out), "sp_host_points_sum_encode");
  s.stats.gathers++; s.stats.bytes += sizeof(sp_host_point) * Ls * W;
  return true;
}

// DensePolynomial::commit_inner over the shards of the context; returns false when the commitment is not sharded (no
// sharding configured, or too few rows per shard) and the caller takes the single-GPU path.
bool sharded_commit_rows(sp_ctx* c, const sp_gens* g, size_t g_off, size_t h_idx, const sp_table* Z, size_t Ls, size_t Rs, const uint64_t* blinds,
                         uint8_t* out /*32*Ls*/) {
  ShardState* sp = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_state.find(c);
    if (it == g_state.end() || it->second.mode == 0) return false;
    sp = &it->second;  // entries are only removed by the owning thread (set_* / ~Ctx); shard configuration and sharded proving of one context are single-threaded, like the context itself
  }
  ShardState& s = *sp;
  size_t W = (size_t)s.world;
  if (W <= 1 && s.mode != 2) return false;
  if (Ls <= 8) return sharded_commit_cols(s, c, g, g_off, h_idx, Z, Ls, Rs, blinds, out);  // fewer rows than a shard is worth: by columns
  if (Ls % W != 0 || Ls / W <= 8) return false;  // rows are independent MSMs: shard only when every rank gets a real batch
  size_t per = Ls / W;
  auto chk = [](int32_t rc, const char* what) { if (rc != SP_OK) throw Error(std::string(what) + " failed: " + sp_strerror(rc)); };
  if (s.mode == 3) {
    // virtual shards: shard k on its own context (own streams) of the same GPU, all in flight together
    chk(sp_ctx_sync(c), "sp_ctx_sync");  // Z as produced by everything queued on the owning context
    std::vector<sp_job*> jobs(W, nullptr);
    try {
      for (size_t k = 0; k < W; k++) {
        sp_ctx* ck = k == 0 ? c : s.vctx[k - 1];
        chk(sp_commit_rows_dev_start(ck, g, g_off, h_idx, Z, k * per * Rs, per, Rs, blinds ? blinds + 4 * k * per : nullptr, &jobs[k]), "sp_commit_rows_dev_start");
      }
      for (size_t k = 0; k < W; k++) {
        sp_job* j = jobs[k];
        jobs[k] = nullptr;
        chk(sp_job_wait(j, out + 32 * k * per), "sp_job_wait");
      }
    } catch (...) {
      std::vector<uint8_t> sink(32 * per);
      for (sp_job* j : jobs) if (j) (void)sp_job_wait(j, sink.data());
      throw;
    }
    s.stats.gathers++; s.stats.bytes += 32 * Ls;
    return true;
  }
  size_t lo = per * (size_t)s.rank;
  chk(sp_commit_rows_dev(c, g, g_off, h_idx, Z, lo * Rs, per, Rs, blinds ? blinds + 4 * lo : nullptr, out + 32 * lo), "sp_commit_rows_dev");
  gather_bytes(s, out, 32 * per);  // the compressed commitments, 32*per bytes from each rank, rank order
  s.stats.gathers++; s.stats.bytes += 32 * Ls;
  return true;
}

}  // namespace spz
