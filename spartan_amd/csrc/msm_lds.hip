// spartan_amd: the LDS-staged small-window form of the fixed-base row MSM (BASELINE.json north_star: "scalars and generator
// windows staged in LDS ... built on gfx950 wavefront ballot/shuffle primitives").
//
// Replaces the same reference code as the wide-window forms of core.hip — the rows of DensePolynomial::commit_inner
// (src/dense_mlpoly.rs:164-177), i.e. [Scalar]::commit = vartime_multiscalar_mul over MultiCommitGens + blind * h
// (src/commitments.rs:80-92, src/group.rs:98-117) — with the opposite trade: the wide form spends HBM (15-bit windows, 118 GB of
// tables at 2^20, one random 128-byte gather per mixed addition) to do 17-19 additions per scalar; this form keeps 10-bit signed
// windows (26 additions per scalar, 1.25 MB of table per generator: 6.5 GB at 2^20) and never gathers from HBM at all:
//
//   * a workgroup is up to 1024 ROWS of the matrix (one lane per row, 16 wavefronts = 4 per SIMD, the whole CU) and walks a run of
//     (column, window) TILES. All its lanes need the same (generator, window) sub-table for a tile: 512 entries x 96 B = 48 KB,
//     which is streamed from the table into LDS by global_load_lds_dwordx4 (LDS-DMA: coalesced 1 KB per wave-instruction, no staging
//     registers) into one half of a double buffer while the lanes work out of the other half;
//   * the lane's gather is a 96-byte LDS read at its own digit; the mixed addition (7 F_p multiplications) follows; one workgroup
//     barrier per tile hands the buffers over (the DMA of tile t+1 is issued during the addition of tile t — by loader wavefronts, or
//     from inside the addition — and has until the next barrier to land);
//   * HBM sees a sequential stream: rows/1024 x cols x 26 x 48 KB per commit (1.3 GB for the 2^20 witness, ~1.3 TB/s while the
//     kernel runs) instead of 107-126 B of random gather per addition at 88-93 % L2 miss.
//
// The unit of work is the tile, and a row-block's cols x nwin tiles are cut into nb equal runs (as in the balanced form of core.hip),
// so a launch is exactly as many workgroups as the chip (or the background share of it) holds and all finish together; a run may start
// in the middle of a scalar (the signed recoding's carry into its first window is rebuilt from the lower windows).
// Scalars leave Montgomery form once per column (one Montgomery reduction per 26 additions); the next column's scalar is requested
// one column ahead. Partial sums go to partial[row][run]; the cross-run reduction and the encodes are the existing kernels of core.hip
// (DPP point-addition trees, k_msm_reduce / k_pt_encode).
#include "internal.hpp"


static_assert(sizeof(NielsP) == 96, "packed Niels entry");

struct MsmLdsArgs {
  const Fq* Z; size_t z_row_stride, rows, cols;
  const NielsP* table; size_t g_off; const uint32_t* idx; const Fq* blinds; size_t h_idx;
  Pt* partial;              // [rows][nb]
  unsigned nb, nrb, rows_per_wg;  // runs per row-block; row-blocks; rows of a row-block (<= blockDim.x)
  unsigned n_wg;            // nb * nrb; a launch of fewer workgroups (the persistent background form) walks them with a grid stride
  int wbits, nwin, tent;
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

// DIAG (timing experiments of `make variant NAME=ldsdiagN FLAGS=-DSP_LDS_DIAG=N` only; WRONG RESULTS unless 0): bit 0 no DMA, bit 1 no barrier,
// bit 2 no LDS gather (one fixed entry), bit 3 no addition — what each stage of a tile costs (bench/msm_lds_probe.py ... diag).
// First measurement (profiles/r5_lds_diag_v1.txt, 1536 x 4096): the additions alone 7.1 of 8.1 ms; the DMA costs 0.8 ms although it has a
// whole addition to land in — its ISSUE does: right after the barrier every wave of the CU queues its 3-4 LDS-DMA pieces (~100 cycles of
// issue each) before any of them starts multiplying. So the pieces are issued from INSIDE the addition, one after each of its first
// multiplications, by whichever wave gets there: the other waves of the SIMD multiply meanwhile.
__device__ __forceinline__ Fp lds_fp(const uint8_t* p) {  // 32 bytes of an entry out of LDS (two ds_read_b128)
  const uint4* e = reinterpret_cast<const uint4*>(p);
  uint4 a = e[0], b = e[1];
  return Fp{{(uint64_t)a.x | ((uint64_t)a.y << 32), (uint64_t)a.z | ((uint64_t)a.w << 32), (uint64_t)b.x | ((uint64_t)b.y << 32), (uint64_t)b.z | ((uint64_t)b.w << 32)}};
}
// = pt_madd (curve.hpp) against the entry at `e` in LDS: branch-free, each field of the entry read just before the multiplication that
// needs it (8 live registers of entry instead of 24), hook(k) after multiplication k
template <class Hook>
__device__ __forceinline__ Pt pt_madd_lds(const Pt& p, const uint8_t* e, bool neg, Hook&& hook) {
  // p - n = p + (-n), and -n swaps y+x with y-x and negates 2dxy: no branch on the sign anywhere
  Fp A = fp_mul(fp_sub(p.Y, p.X), lds_fp(e + (neg ? 0 : 32)));   // n.ym, or n.yp for a subtraction
  hook(0);
  Fp B = fp_mul(fp_add(p.Y, p.X), lds_fp(e + (neg ? 32 : 0)));
  hook(1);
  Fp t2 = lds_fp(e + 64);
  Fp C = fp_mul(p.T, fp_select(t2, fp_neg(t2), neg));
  hook(2);
  Fp Dd = fp_add(p.Z, p.Z);
  Fp E = fp_sub(B, A), H = fp_add(B, A);
  Fp F = fp_sub(Dd, C), G = fp_add(Dd, C);
  Fp X3 = fp_mul(E, F);
  hook(3);
  return Pt{X3, fp_mul(G, H), fp_mul(F, G), fp_mul(E, H)};
}

// one run of tiles (bk of nb) for one row-block rb; blockDim.x lanes, lanes >= rows_per_wg (or past the last row) only help with the DMA
template <int DIAG>
__device__ __forceinline__ void msm_lds_run(const MsmLdsArgs& A, unsigned rb, unsigned bk, uint8_t* lds) {
  const unsigned T = blockDim.x, tid = threadIdx.x, lane = tid & 63;
  const size_t row = (size_t)rb * A.rows_per_wg + tid;
  const bool live = tid < A.rows_per_wg && row < A.rows;
  const int nwin = A.nwin, c = A.wbits;
  const size_t ncol = A.cols + (A.blinds ? 1 : 0);
  const size_t U = ncol * (size_t)nwin;
  const size_t u0 = U * bk / A.nb, u1 = U * (bk + 1) / A.nb;
  const unsigned sub_bytes = (unsigned)A.tent * 96u;   // a multiple of 1 KB for every width >= 5: the number of DMA pieces is wave-uniform
  const unsigned npieces = ((sub_bytes >> 10) + (T >> 6) - 1) / (T >> 6);  // 1 KB pieces per wavefront when every wavefront issues its share
  // LDS: [buffer 0][buffer 1][the neutral entry (1, 1, 0): what a lane with a zero digit adds]
  const unsigned ident_off = 2 * sub_bytes;
  if (tid < 6) {
    uint4 v = make_uint4(0, 0, 0, 0);
    if (tid == 0 || tid == 2) v.x = 1;  // yp = 1, ym = 1 (limb 0), t2d = 0
    reinterpret_cast<uint4*>(lds + ident_off)[tid] = v;
  }
  auto scalar_ptr = [&](size_t jj) { return jj < A.cols ? A.Z + row * A.z_row_stride + jj : A.blinds + row; };
  auto col_base = [&](size_t jj) {  // the window tables of column jj's generator (wave-uniform)
    const size_t pt = jj < A.cols ? (A.idx ? (size_t)A.idx[jj] : A.g_off + jj) : A.h_idx;
    return reinterpret_cast<const uint8_t*>(A.table + pt * (size_t)nwin * (size_t)A.tent);
  };
  // Who issues the LDS-DMA of a sub-table (1 KB pieces: 64 lanes x 16 bytes, LDS side = wave-uniform base + lane * 16):
  //  * a workgroup with wavefronts to spare (rows_per_wg <= 768: the launch adds up to four LOADER wavefronts that own no rows) has the
  //    loaders issue every piece while the row wavefronts do nothing but multiply — the loader / consumer split; the loaders also fill the
  //    CU's register file, so no other kernel's workgroup lands on a CU that runs this one (the background launch relies on that);
  //  * a full workgroup (1024 rows) has every wavefront issue its share from inside its addition, one piece after each of the first
  //    multiplications (pt_madd_lds hooks): the issue cost (~100 cycles a piece) is covered by the other wavefronts of the SIMD.
  const unsigned wave = tid >> 6, nwaves = T >> 6, nlive = (A.rows_per_wg + 63) >> 6;
  const unsigned nload = nwaves > nlive ? nwaves - nlive : 0;
  const bool loader = wave >= nlive;  // wave-uniform
  const unsigned kb = sub_bytes >> 10;
  auto piece = [&](const uint8_t* src, unsigned buf, unsigned p) {  // piece p of the sub-table at src into buffer buf
    __builtin_amdgcn_global_load_lds((glb_ptr_t)(src + p * 1024u + lane * 16u), (lds_ptr_t)(lds + buf * sub_bytes + p * 1024u), 16, 0, 0);
  };
  auto dma_piece = [&](const uint8_t* src, unsigned buf, unsigned k) {  // hooked form: this wavefront's k-th piece
    const unsigned p = k * nwaves + wave;
    if (nload == 0 && p < kb) piece(src, buf, p);
  };
  auto dma_all = [&](const uint8_t* src, unsigned buf) {  // everything this wavefront owes to a sub-table, at once
    if (nload) { if (loader) for (unsigned p = wave - nlive; p < kb; p += nload) piece(src, buf, p); }
    else for (unsigned p = wave; p < kb; p += nwaves) piece(src, buf, p);
  };
  Pt acc = pt_identity();
  if (u1 > u0) {
    size_t j = u0 / (size_t)nwin;
    int w = (int)(u0 % (size_t)nwin);
    uint64_t s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    int carry = 0;
    const uint32_t mask = (1u << c) - 1;
    auto take = [&](const Fq& raw) {
      Fq s = fq_from_mont(raw);  // canonical integer < q < 2^253 (scalar/mod.rs:32-36 does the same for dalek)
      s0 = s.l[0]; s1 = s.l[1]; s2 = s.l[2]; s3 = s.l[3];
      carry = 0;
    };
    auto shift = [&]() {
      s0 = (s0 >> c) | (s1 << (64 - c));
      s1 = (s1 >> c) | (s2 << (64 - c));
      s2 = (s2 >> c) | (s3 << (64 - c));
      s3 >>= c;
    };
    // the tile after the current one: column, window and where its sub-table lies (all wave-uniform, advanced without divisions)
    size_t jn = j;
    int wn = w;
    const uint8_t* cbase = col_base(jn);
    if (!(DIAG & 1)) dma_all(cbase + (size_t)wn * sub_bytes, 0);
    auto advance_next = [&]() {
      if (++wn == nwin) { wn = 0; jn++; if (jn < ncol) cbase = col_base(jn); }
    };
    advance_next();
    Fq raw_next = fq_zero();
    if (live) {
      take(ld_fq(scalar_ptr(j)));
      if (j + 1 < ncol) raw_next = ld_fq(scalar_ptr(j + 1));
      for (int k = 0; k < w; k++) {  // the carry into window w depends on all lower windows
        int d = (int)(s0 & mask) + carry;
        carry = d >= A.tent;
        shift();
      }
    }
    for (size_t u = u0; u < u1; u++) {
      const unsigned buf = (unsigned)(u - u0) & 1u;
      bool fetch_next = false;
      if (w == nwin) {
        j++; w = 0;
        if (live) take(raw_next);
        fetch_next = true;
      }
      int d = (int)(s0 & mask) + carry;
      carry = d >= A.tent;
      d -= carry << c;
      const uint32_t m = (uint32_t)(d < 0 ? -d : d);
      shift();
      w++;
      if (!(DIAG & 2)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this lane's share of tile u has landed in LDS
        __syncthreads();                                   // ... and everyone's; and every lane is done reading the other buffer (tile u - 1)
      }
      if (fetch_next && live && j + 1 < ncol) raw_next = ld_fq(scalar_ptr(j + 1));
      const bool more = !(DIAG & 1) && u + 1 < u1;
      const uint8_t* src = cbase + (size_t)wn * sub_bytes;
      // a zero digit (and a lane without a row) adds the neutral entry: the same point in other coordinates, so the canonical bytes of
      // the sum do not change — and the addition, with the DMA pieces inside it, runs outside any divergent branch
      const unsigned eoff = (m == 0 || (DIAG & 4)) ? ident_off : buf * sub_bytes + (m - 1) * 96u;
      const uint8_t* e = lds + eoff;
      if (loader) {  // (wave-uniform) no rows here: the next sub-table, then the barrier
        if (more) dma_all(src, buf ^ 1u);
      } else if (DIAG & 8) {
        acc.X.v[0] ^= lds_fp(e).v[0] ^ lds_fp(e + 32).v[1] ^ lds_fp(e + 64).v[2];
        if (more) dma_all(src, buf ^ 1u);
      } else {
        acc = pt_madd_lds(acc, e, d < 0, [&](unsigned k) { if (more) dma_piece(src, buf ^ 1u, k); });
        if (more) for (unsigned k = 4; k < npieces; k++) dma_piece(src, buf ^ 1u, k);  // narrow workgroups: the rest of the pieces
      }
      advance_next();
    }
    __syncthreads();  // persistent form: the next run's first DMA must not overtake this run's last reads
  }
  if (live) A.partial[row * A.nb + bk] = acc;
}

template <int DIAG>
__global__ void __launch_bounds__(1024) k_msm_lds(MsmLdsArgs A) {
  extern __shared__ __attribute__((aligned(16))) uint8_t msm_lds_buf[];
  // XCD-aware order (workgroup b runs on XCD b % 8, each XCD has its own L2): the row-blocks of one run stream the SAME sub-tables, so
  // they are given to the same XCD, back to back — the second one's DMA hits in L2
  const bool xcd = gridDim.x == A.n_wg && A.nrb > 1 && A.n_wg % (8 * A.nrb) == 0;
  for (unsigned wg = blockIdx.x; wg < A.n_wg; wg += gridDim.x) {
    const unsigned rb = xcd ? (wg / 8) % A.nrb : wg % A.nrb, bk = xcd ? (wg / (8 * A.nrb)) * 8 + wg % 8 : wg / A.nrb;
    msm_lds_run<DIAG>(A, rb, bk, msm_lds_buf);
  }
}

// ------------------------------------------------------------------------------------------------ host side
// rows per workgroup and row-blocks of an LDS-form launch
static void msm_lds_shape(size_t rows, unsigned* nrb, unsigned* rows_per_wg, unsigned* threads) {
  size_t b = (rows + 1023) / 1024;
  size_t per = (rows + b - 1) / b;
  size_t t = (per + 63) / 64 * 64;
  for (int k = 0; k < 4 && t + 64 <= 1024; k++) t += 64;  // up to four loader wavefronts (they own no rows: the LDS-DMA of the sub-tables)
  *nrb = (unsigned)b; *rows_per_wg = (unsigned)per; *threads = (unsigned)t;
}
size_t msm_lds_runs(const sp_gens* g, size_t rows, size_t cols, bool has_blinds, size_t wg_slots) {
  unsigned nrb, per, thr;
  msm_lds_shape(rows, &nrb, &per, &thr);
  size_t units = (cols + (has_blinds ? 1 : 0)) * (size_t)g->geom_lds.nwin;
  size_t nb = wg_slots / nrb;
  if (nb < 1) nb = 1;
  if (nb > units / 4) nb = units / 4;  // at least four tiles per run
  if (nb < 1) nb = 1;
  return nb;
}
void msm_lds_enqueue(sp_ctx* c, hipStream_t st, const sp_gens* g, const Fq* dZ, size_t z_stride, size_t rows, size_t cols, size_t g_off,
                     const uint32_t* didx, const Fq* dblinds, size_t h_idx, Pt* partial, size_t nb, unsigned grid_limit) {
  MsmLdsArgs A;
  A.Z = dZ; A.z_row_stride = z_stride; A.rows = rows; A.cols = cols;
  A.table = g->table_lds; A.g_off = g_off; A.idx = didx; A.blinds = dblinds; A.h_idx = h_idx;
  A.partial = partial;
  unsigned thr;
  msm_lds_shape(rows, &A.nrb, &A.rows_per_wg, &thr);
  A.nb = (unsigned)nb; A.n_wg = A.nb * A.nrb;
  A.wbits = g->geom_lds.wbits; A.nwin = g->geom_lds.nwin; A.tent = g->geom_lds.tent;
  unsigned grid = A.n_wg;
  if (grid_limit && grid > grid_limit) grid = grid_limit;
  size_t lds = 2 * (size_t)A.tent * 96 + 96;
  if (lds < 81920) lds = 81920;  // more than half of a CU's LDS: a CU never holds two of these (narrow windows)
  auto launch = [&](auto kern) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);  // more than the default 64 KB of dynamic LDS
    hipLaunchKernelGGL(kern, dim3(grid), dim3(thr), (unsigned)lds, st, A);
  };
#ifdef SP_LDS_DIAG   // variant build only (make variant NAME=ldsdiagN FLAGS=-DSP_LDS_DIAG=N): the stage mask is compiled in
  launch(k_msm_lds<SP_LDS_DIAG>);
  return;
#endif
  launch(k_msm_lds<0>);
}
