// spartan_amd: the QUEUE form of the fixed-base row MSM (round 6) — self-contained wavefronts that pull work from a device-side queue.
//
// Replaces the same reference code as the other forms — the rows of DensePolynomial::commit_inner (src/dense_mlpoly.rs:164-177), i.e.
// [Scalar]::commit = vartime_multiscalar_mul over MultiCommitGens + blind * h (src/commitments.rs:80-92, src/group.rs:98-117), and
// Derefs::commit (src/sparse_mlpoly.rs:64-67) — over the same wide-window tables (15/14-bit signed windows, one 128-byte line per entry).
//
// What the earlier forms measured: the strip form (core.hip) keeps its table entries in flight in REGISTERS (48 of
// its 164 VGPRs) and still waits on memory 42-47 % of its wave cycles; the ring form (msm_lds.hip) moved the gathers to LDS-DMA issued by
// loader wavefronts and reached 0.75-0.82 of the mixed-addition ceiling, but its unit of scheduling is a 1024-thread workgroup in
// lock-step (one barrier per tile, ~1 ms of indivisible work per CU) — faster per launch, no faster in the proof. Here:
//
//   * the unit of execution is ONE WAVEFRONT = 64 rows of the matrix. It owns a private ring of D slots in LDS (6 KB each: the 96 bytes of
//     one table entry per lane, chunk-major so that the LDS side of a DMA is wave-uniform base + lane * 16 and the lane's reads are
//     conflict-free) and gathers ITS OWN entries with global_load_lds_dwordx4, per-lane source address, D - 1 tiles ahead of the mixed
//     addition that consumes them. No other wavefront ever reads its slots: there is NO barrier in the loop — a counted s_waitcnt vmcnt
//     is the only synchronisation — and no register holds an entry in flight;
//   * the unit of scheduling is an ITEM = (64-row group, run of msm.q_units (column, window) units), ~0.2 ms of work. Every row group has
//     its own queue of runs (an atomic head in device memory). A wavefront ATTACHES to a group — its home group first — takes a slot of
//     that group's partial sums, and pulls runs of THAT group into ONE accumulator until the group's queue is empty; only then does it
//     write its partial sum (one per attachment, not one per item) and look for another group that still has runs (a vector scan of the
//     heads: work stealing at the tail). A launch is therefore correct and balanced on ANY number of workgroups of ANY size — the
//     background launch on a share of the CUs, the foreground launch whose workgroups start as CUs free up — it ends within one short
//     item of the first idle wavefront (the work-conserving schedule VERDICT r5 asked for), and the partial sums stay ~workers / groups
//     per row however fine the items are (first version: one partial per item — 3 893 per row at 2^22, 8 ms of reduction passes);
//   * digits come from the same signed recoding as every other form (msm.hpp), produced D - 1 tiles ahead; a zero digit adds the neutral
//     entry (1, 1, 0) from LDS, so the addition is branch-free; when no lane of the wavefront has anything left in the current scalar
//     (ballot), the stream jumps to the next column without issuing the remaining gathers (short scalars: SNARK::encode's addresses
//     and timestamps, src/sparse_mlpoly.rs:483-503).
//
// Partial sums go to partial[row][run]; the cross-run reduction and the encodes are the existing kernels of core.hip.
#include "internal.hpp"

struct MsmQArgs {
  const Fq* Z; size_t z_row_stride, rows, cols;
  const Niels* table; size_t g_off; const uint32_t* idx; const Fq* blinds; size_t h_idx;
  Pt* partial;              // [rows][S]: slot s of a row group belongs to the s-th wavefront that attached to it
  unsigned* heads;          // [ngroups] next run of each row group (zero when the launch starts)
  unsigned* nslots;         // [ngroups] attachments so far (may run past S: refused ones count too; the reduction takes min(nslots, S))
  unsigned nb, ngroups, len, S;  // runs per row; 64-row groups; units per run; partial-sum slots per row
  unsigned long long* issued;     // profiling runs: tiles issued by all wavefronts (x 64 = mixed additions actually performed: the ballot skips
                                  // the upper windows of short scalars), or null
  MsmGeom geom;   // window geometry of the set's tables (mixed widths: msm.hpp)
};

typedef __attribute__((address_space(3))) void* q_lds_ptr_t;

constexpr unsigned MSMQ_SLOT = 6 * 1024;  // one tile of one wavefront: 64 entries of 96 bytes, row-major

__device__ __forceinline__ Fp q_lds_fp(const uint8_t* p) {  // 32 bytes of an entry out of LDS (two ds_read_b128)
  uint4 a = *reinterpret_cast<const uint4*>(p), b = *reinterpret_cast<const uint4*>(p + 16);
  return Fp{{(uint64_t)a.x | ((uint64_t)a.y << 32), (uint64_t)a.z | ((uint64_t)a.w << 32), (uint64_t)b.x | ((uint64_t)b.y << 32), (uint64_t)b.z | ((uint64_t)b.w << 32)}};
}
// One tile of one wavefront: six LDS-DMA instructions bring the 64 lanes' table entries (96 bytes each) into a 6 KB slot. The LDS side of an
// LDS-DMA is fixed — M0 + lane * 16 — but the SOURCE address is per lane, so the wavefront fetches TRANSPOSED: lane l of instruction i
// brings piece (64 i + l) % 6 of the entry of row (64 i + l) / 6. Six consecutive lanes then read the six consecutive 16-byte pieces of ONE
// entry — one 128-byte line, coalesced by the texture addresser into one request — and the slot comes out row-major (row r at r * 96). The
// first version had every lane fetch its own six pieces (six requests for the same line, 384 per tile instead of ~70): its gathers alone ran
// at 28-30 G entries/s against an addition side of 24 G/s in the same loop (profiles/r6_queue_diag.txt: the two sides overlapped badly).
// Issued in INLINE ASSEMBLY on purpose: for a __builtin_amdgcn_global_load_lds the compiler knows of a pending write to LDS and puts
// `s_waitcnt vmcnt(0)` in front of the next ds_read of the kernel's LDS array — every gather would be waited for by the addition issued right
// behind it, the ring would be one tile deep whatever D says (first build of this kernel: `.s` inspected). What the compiler does not see it
// does not wait for; the waits are counted by hand (q_wait_tiles). M0 is compiler-reserved and not preserved around a statement: saved and
// restored inside it.
__device__ __forceinline__ void q_gather_tile(const uint8_t* a0, const uint8_t* a1, const uint8_t* a2, const uint8_t* a3, const uint8_t* a4,
                                              const uint8_t* a5, unsigned lds_dst /* wave-uniform LDS byte address */) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %7\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\t"
      "s_add_u32 m0, m0, 0x400\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %2, off\n\t"
      "s_add_u32 m0, m0, 0x400\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %3, off\n\t"
      "s_add_u32 m0, m0, 0x400\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %4, off\n\t"
      "s_add_u32 m0, m0, 0x400\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %5, off\n\t"
      "s_add_u32 m0, m0, 0x400\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %6, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "s"(lds_dst)
      : "memory", "scc");
}
// At most `tiles` of this wavefront's tiles (6 LDS-DMA instructions each) may still be in flight. VMEM loads of a wavefront complete in
// order, so a load the COMPILER issued in between (the next column's scalar) only makes this wait stricter than needed — and the compiler's
// own counted waits, which do not know of the asm gathers queued behind its loads, are stricter than needed for the same reason: never weaker.
__device__ __forceinline__ void q_wait_tiles(int tiles) {
  switch (tiles) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break;
  }
}

// D = 2 ring slots per wavefront: one tile of gathers in flight under the addition of the previous one. (Three slots with two wavefronts
// per SIMD measured no better — the gathers are not waited for: profiles/r6_queue_diag.txt — and were retired with the variant that ran the
// background launch on a fenced share of the CUs: without the fence the latency kernels were dispatched onto the background CUs and lost
// every issue arbitration there, second sum-check 1.2 -> 4.1 ms; with it the reserved CUs idled. The co-resident form replaced both.)
constexpr int MSMQ_D = 2;
// SP_Q_DIAG (timing experiments of `make variant NAME=qdiagN FLAGS=-DSP_Q_DIAG=N` only; WRONG RESULTS unless 0): bit 0 no gathers (every
// addition takes the neutral entry), bit 1 no additions (the gathers and their waits alone), bit 2 gathers issued but not waited for (the
// additions read whatever the slot holds: the cost of ISSUING the gathers without their latency) — what each side of a tile costs in this loop
#ifndef SP_Q_DIAG
#define SP_Q_DIAG 0
#endif
__global__ void __launch_bounds__(768) k_msm_q(MsmQArgs A) {
  constexpr int D = MSMQ_D;
  extern __shared__ __attribute__((aligned(16))) uint8_t q_lds[];
  const unsigned tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
#if SP_Q_DIAG & 8   // bit 3: the shader clock this kernel runs at (shader cycles over the 100 MHz wall clock), printed by one wavefront
  const unsigned long long dg_c0 = clock64(), dg_w0 = wall_clock64();
#endif
  uint8_t* const ring = q_lds + wave * (D * MSMQ_SLOT);
  // the ring's LDS byte address, provably wave-uniform for the "s" operand of the gather statement
  const unsigned ring_lds = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)(q_lds_ptr_t)ring);
  uint8_t* const ident = q_lds + nwaves * (D * MSMQ_SLOT);  // the neutral entry (1, 1, 0), 96 bytes, shared and read-only
  if (tid < 6) {
    uint4 v = make_uint4(0, 0, 0, 0);
    if (tid == 0 || tid == 2) v.x = 1;  // yp = 1, ym = 1 (limb 0), t2d = 0
    reinterpret_cast<uint4*>(ident)[tid] = v;
  }
  __syncthreads();  // the only barrier of the kernel
  // the transposed fetch: which row's entry, and which 16-byte piece of it, this lane brings in gather instruction i
  int sel[6];        // byte address of the source lane for ds_bpermute
  unsigned poff[6];  // byte offset of the piece inside the entry
#pragma unroll
  for (int i = 0; i < 6; i++) {
    const unsigned t = 64u * i + lane;
    sel[i] = (int)((t / 6u) * 4u);
    poff[i] = (t % 6u) * 16u;
  }
  const int nwin = A.geom.nwin, n0 = msm_n0(A.geom);
  const size_t ncol = A.cols + (A.blinds ? 1 : 0);
  const size_t U = ncol * (size_t)nwin;
  // this wavefront's walk over the row groups: position 0 is its home group (wavefronts are dealt to the groups round-robin), positions only
  // ever advance — a group it has left is exhausted, a group that had no slot for it is left to the wavefronts attached there
  const unsigned home = (blockIdx.x * nwaves + wave) % A.ngroups;
  unsigned long long n_issued = 0;
  for (unsigned pos = 0; pos < A.ngroups;) {
    {  // the next position whose group still has runs (lanes look at 64 heads at a time)
      unsigned found = A.ngroups;
      for (unsigned base = pos; base < A.ngroups && found == A.ngroups; base += 64) {
        const unsigned q = base + lane;
        unsigned h = A.nb;
        if (q < A.ngroups) h = __hip_atomic_load(A.heads + (home + q) % A.ngroups, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long ok = __ballot(h < A.nb);
        if (ok) found = base + (unsigned)__builtin_ctzll(ok);
      }
      pos = found;
      if (pos >= A.ngroups) break;
    }
    const unsigned grp = (home + pos) % A.ngroups;
    pos++;
    unsigned slot = 0;
    if (lane == 0) slot = atomicAdd(A.nslots + grp, 1u);
    slot = (unsigned)__builtin_amdgcn_readfirstlane((int)slot);
    if (slot >= A.S) continue;  // every slot of this group is taken (by wavefronts that will finish it)
    const size_t row = (size_t)grp * 64 + lane;
    const bool live = row < A.rows;
    auto ld_scalar = [&](size_t jj) {  // Montgomery form of column jj of this lane's row (the blind is column `cols`)
      if (!live) return fq_zero();
      return ld_fq(jj < A.cols ? A.Z + row * A.z_row_stride + jj : A.blinds + row);
    };
    auto col_base = [&](size_t jj) {  // the window tables of column jj's generator (wave-uniform)
      const size_t pt = jj < A.cols ? (A.idx ? (size_t)A.idx[jj] : A.g_off + jj) : A.h_idx;
      return reinterpret_cast<const uint8_t*>(A.table + pt * A.geom.pt_entries);
    };
    Pt acc = pt_identity();
    for (;;) {
      unsigned bk = 0;
      if (lane == 0) bk = atomicAdd(A.heads + grp, 1u);
      bk = (unsigned)__builtin_amdgcn_readfirstlane((int)bk);
      if (bk >= A.nb) break;
      const size_t u = (size_t)bk * A.len;
      size_t u1 = u + A.len;
      if (u1 > U) u1 = U;
      // ---- the ring: tiles issued and not yet consumed, oldest first; flags of tile (issued - 1 - k) in bits 2k+1:2k of hist
      unsigned hist = 0, slot_w = 0, slot_r = 0;
      int inflight = 0;
      auto consume = [&]() {  // the mixed addition of the oldest tile in flight
        const int p = inflight - 1;       // tiles issued after it
        if (!(SP_Q_DIAG & 5)) q_wait_tiles(p);  // (bit 2: gathers issued, never waited for)
        const unsigned fl = (hist >> (2 * p)) & 3u;
        const bool neg = fl & 1u, zero = (SP_Q_DIAG & 1) ? true : (fl & 2u) != 0;
        if (SP_Q_DIAG & 2) { acc.X.v[0] ^= fl; slot_r = slot_r + 1 == (unsigned)D ? 0 : slot_r + 1; inflight--; return; }
        // this lane's entry (y+x, y-x, 2dxy) at lane * 96 of the slot; a zero digit adds the neutral entry (the same point in other
        // coordinates, so the canonical bytes of the sum do not change — and the addition runs outside any divergent branch)
        const uint8_t* c0 = zero ? ident : ring + slot_r * MSMQ_SLOT + lane * 96u;
        const unsigned a_off = neg ? 0u : 32u, b_off = neg ? 32u : 0u;  // p - n = p + (-n): -n swaps y+x with y-x and negates 2dxy
        Fp Am = fp_mul(fp_sub(acc.Y, acc.X), q_lds_fp(c0 + a_off));
        Fp Bm = fp_mul(fp_add(acc.Y, acc.X), q_lds_fp(c0 + b_off));
        Fp t2 = q_lds_fp(c0 + 64);
        Fp Cm = fp_mul(acc.T, fp_select(t2, fp_neg(t2), neg));
        Fp Dd = fp_add(acc.Z, acc.Z);
        Fp E = fp_sub(Bm, Am), H = fp_add(Bm, Am);
        Fp F = fp_sub(Dd, Cm), G = fp_add(Dd, Cm);
        acc = Pt{fp_mul(E, F), fp_mul(G, H), fp_mul(F, G), fp_mul(E, H)};
        slot_r = slot_r + 1 == (unsigned)D ? 0 : slot_r + 1;
        inflight--;
      };
      // ---- the digit stream, column by column. The next column's scalar is requested a column ahead by an ORDINARY load that lives across
      // the window loop untouched (it must not be carried through that loop: the compiler would copy its registers — and wait for the
      // load — at every tile); its use at the top of the next column is where the compiler drains the wavefront's loads, once per column.
      const size_t j0 = u / (size_t)nwin, j1 = (u1 - 1) / (size_t)nwin;
      const int w_first = (int)(u % (size_t)nwin), w_last = (int)((u1 - 1) % (size_t)nwin) + 1;
      Fq raw = ld_scalar(j0);
      for (size_t j = j0; j <= j1; j++) {
        Fq s = fq_from_mont(raw);  // canonical integer < q < 2^253 (scalar/mod.rs:32-36 does the same for dalek)
        uint64_t s0 = s.l[0], s1 = s.l[1], s2 = s.l[2], s3 = s.l[3];
        int carry = 0;
        raw = j + 1 < ncol ? ld_scalar(j + 1) : fq_zero();
        auto shift = [&](int c) {
          s0 = (s0 >> c) | (s1 << (64 - c));
          s1 = (s1 >> c) | (s2 << (64 - c));
          s2 = (s2 >> c) | (s3 << (64 - c));
          s3 >>= c;
        };
        int w = 0;
        if (j == j0)
          for (; w < w_first; w++) {  // a run may start inside a scalar: the carry into window w depends on all lower windows
            const int c = A.geom.wbits + (w >= n0 ? 1 : 0);
            int d = (int)(s0 & ((1u << c) - 1)) + carry;
            carry = d >= (1 << (c - 1));
            shift(c);
          }
        const int w_end = j == j1 ? w_last : nwin;
        const uint8_t* cbase = col_base(j);
        for (; w < w_end; w++) {
          // nothing left in this scalar on any lane of the wavefront: no gathers for its upper windows (short scalars)
          if (__all((s0 | s1 | s2 | s3) == 0 && carry == 0)) break;
          const int c = A.geom.wbits + (w >= n0 ? 1 : 0);   // the top windows are one bit wider (mixed widths)
          int d = (int)(s0 & ((1u << c) - 1)) + carry;
          carry = d >= (1 << (c - 1));
          d -= carry << c;
          const uint32_t m = (uint32_t)(d < 0 ? -d : d);
          shift(c);
          if (!(SP_Q_DIAG & 1)) {
            const uint8_t* sub = cbase + msm_woff(A.geom, w) * sizeof(Niels);  // the tile's (generator, window) sub-table (wave-uniform)
            const int eidx = (int)(m ? m - 1 : 0);               // this lane's entry in it
            const uint8_t* a[6];
#pragma unroll
            for (int i = 0; i < 6; i++)
              a[i] = sub + (((size_t)(unsigned)__builtin_amdgcn_ds_bpermute(sel[i], eidx)) << 7) + poff[i];
            q_gather_tile(a[0], a[1], a[2], a[3], a[4], a[5], ring_lds + slot_w * MSMQ_SLOT);
          }
          slot_w = slot_w + 1 == (unsigned)D ? 0 : slot_w + 1;
          hist = (hist << 2) | (d < 0 ? 1u : 0u) | (m == 0 ? 2u : 0u);
          inflight++;
          n_issued++;
          if (inflight == D) consume();  // (the tile just issued went into the slot the previous addition has finished reading)
        }
      }
      while (inflight) consume();
    }
    if (live) A.partial[row * A.S + slot] = acc;  // (the neutral element if another wavefront took the group's last run first)
  }
  if (A.issued && lane == 0 && n_issued) atomicAdd(A.issued, n_issued);
#if SP_Q_DIAG & 8
  if (blockIdx.x == 7 && tid == 0) {
    const unsigned long long c = clock64() - dg_c0, w = wall_clock64() - dg_w0;
    printf("k_msm_q clock: %llu shader cycles in %.1f us -> %.0f MHz\n", c, w * 0.01, c / (w * 0.01));
  }
#endif
}

// ------------------------------------------------------------------------------------------------ host side
// wavefronts per workgroup (4 / 8 / 12 = 1 / 2 / 3 per SIMD) and ring depth of a launch: 12 x 2 x 6 KB or 8 x 3 x 6 KB = 144 KB of the CU's
// 160 KB, so a CU holds exactly one workgroup and a launch of n workgroups occupies n CUs — the partition a CU mask would give
// role of a launch (MsmQRole, internal.hpp): ALONE — the chip to itself: msm.q_waves wavefronts per workgroup on every CU;
// CORESIDENT — the background launch, and a foreground launch that meets one in flight: msm.q_bg_waves wavefronts (8: two per SIMD, 96 KB of
// LDS) on EVERY CU, leaving each CU half of its registers and 64 KB of LDS for the main stream's latency kernels, which outrank the MSM's
// wavefronts in the issue arbitration (SP_FG_PRIO, internal.hpp) — no CU is held in reserve for them
static void msm_q_shape(const sp_ctx* c, int role, unsigned* waves, unsigned* depth, size_t* wgs) {
  unsigned wv = (unsigned)c->opt.v[role != MSMQ_ALONE ? OPT_MSM_Q_BG_WAVES : OPT_MSM_Q_WAVES], d = MSMQ_D;
  *wgs = (size_t)c->n_cus;
  if (wv * d * MSMQ_SLOT + 96 > 160 * 1024) d = 2;
  *waves = wv; *depth = d;
}
// The cut of a launch: runs of msm.q_units (column, window) units per row group, and S partial-sum slots per row — one for each wavefront that
// can attach to a group: its share of the launch's resident wavefronts plus 64 for the ones that come stealing at the tail.
MsmQRuns msm_q_cut(const sp_ctx* c, const sp_gens* g, size_t rows, size_t cols, bool has_blinds, int role) {
  unsigned waves, depth;
  size_t wgs;
  msm_q_shape(c, role, &waves, &depth, &wgs);
  const size_t workers = wgs * waves, ngroups = (rows + 63) / 64;
  const size_t units = (cols + (has_blinds ? 1 : 0)) * (size_t)g->geom.nwin;
  size_t len = (size_t)c->opt.v[OPT_MSM_Q_UNITS];
  const size_t share = (workers + ngroups - 1) / ngroups;  // wavefronts whose home is one group
  if (len * share > units) len = (units + share - 1) / share;  // a small launch: at least one run per resident wavefront
  if (len < 4) len = 4;
  MsmQRuns r;
  r.len = (unsigned)len;
  r.nb = (unsigned)((units + len - 1) / len);
  r.S = (unsigned)(share + 64);
  if (r.S > workers) r.S = (unsigned)workers;
  return r;
}
void msm_q_enqueue(sp_ctx* c, hipStream_t st, const sp_gens* g, const Fq* dZ, size_t z_stride, size_t rows, size_t cols, size_t g_off,
                   const uint32_t* didx, const Fq* dblinds, size_t h_idx, Pt* partial, const MsmQRuns& r, int role,
                   const unsigned** counts_out, const unsigned long long** issued_out) {
  MsmQArgs A;
  A.Z = dZ; A.z_row_stride = z_stride; A.rows = rows; A.cols = cols;
  A.table = g->table; A.g_off = g_off; A.idx = didx; A.blinds = dblinds; A.h_idx = h_idx;
  A.partial = partial;
  A.nb = r.nb; A.len = r.len; A.S = r.S;
  A.ngroups = (unsigned)((rows + 63) / 64);
  A.geom = g->geom;
  // the queues: one of a ring of counter blocks owned by the launching context (heads, then attachment counts), zeroed in stream order in
  // front of the launch; *counts_out tells the reduction how many slots of each group hold a sum
  A.heads = c->q_heads + (size_t)MSMQ_BLOCK_WORDS * (c->q_next++ % MSMQ_BLOCKS);
  A.nslots = A.heads + MSMQ_MAX_GROUPS;
  (void)hipMemsetAsync(A.heads, 0, 4 * (size_t)MSMQ_BLOCK_WORDS, st);
  *counts_out = A.nslots;
  A.issued = c->prof_on ? reinterpret_cast<unsigned long long*>(A.heads + 2 * MSMQ_MAX_GROUPS) : nullptr;  // (zeroed with the block)
  if (issued_out) *issued_out = A.issued;
  // profiling runs read the counter at the next drain: drain before the ring of blocks wraps onto a block that has not been read
  if (c->prof_on && c->q_next % MSMQ_BLOCKS == 0) prof_drain(c);
  unsigned waves, depth;
  size_t wgs;
  msm_q_shape(c, role, &waves, &depth, &wgs);
  const size_t need = ((size_t)A.nb * A.ngroups + waves - 1) / waves;  // no more workgroups than there are items for
  if (wgs > need) wgs = need;
  size_t lds = (size_t)waves * depth * MSMQ_SLOT + 96;
  if (lds < 81920 + 96) lds = 81920 + 96;  // more than half of a CU's LDS: one MSM workgroup per CU whatever its size
  auto launch = [&](auto kern) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(kern, dim3((unsigned)wgs), dim3(64 * waves), (unsigned)lds, st, A);
  };
  launch(k_msm_q);
}
