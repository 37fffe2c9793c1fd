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
//     barrier per tile hands the buffers over (the DMA of tile t+1 is issued right after the barrier of tile t and has the whole
//     addition to land);
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

#include <atomic>

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

// one run of tiles for one row-block; blockDim.x lanes, lanes >= rows_per_wg (or past the last row) only help with the DMA
__device__ __forceinline__ void msm_lds_run(const MsmLdsArgs& A, unsigned wg, uint8_t* lds) {
  const unsigned T = blockDim.x, tid = threadIdx.x, lane = tid & 63;
  const unsigned rb = wg % A.nrb, bk = wg / A.nrb;
  const size_t row = (size_t)rb * A.rows_per_wg + tid;
  const bool live = tid < A.rows_per_wg && row < A.rows;
  const int nwin = A.nwin, c = A.wbits;
  const size_t ncol = A.cols + (A.blinds ? 1 : 0);
  const size_t U = ncol * (size_t)nwin;
  const size_t u0 = U * bk / A.nb, u1 = U * (bk + 1) / A.nb;
  const unsigned sub_bytes = (unsigned)A.tent * 96u;   // a multiple of 1 KB for every width >= 5: the DMA loop's trip count is wave-uniform
  auto scalar_ptr = [&](size_t jj) { return jj < A.cols ? A.Z + row * A.z_row_stride + jj : A.blinds + row; };
  auto dma = [&](size_t u, unsigned buf) {
    const size_t jj = u / (size_t)nwin;
    const int ww = (int)(u % (size_t)nwin);
    const size_t pt = jj < A.cols ? (A.idx ? (size_t)A.idx[jj] : A.g_off + jj) : A.h_idx;
    const uint8_t* src = reinterpret_cast<const uint8_t*>(A.table + (pt * (size_t)nwin + (size_t)ww) * (size_t)A.tent);
    uint8_t* dst = lds + buf * sub_bytes;
    for (unsigned off = tid * 16u; off < sub_bytes; off += T * 16u)
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(src + off), (lds_ptr_t)(dst + (off - lane * 16u)), 16, 0, 0);
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
    dma(u0, 0);
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
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this lane's share of tile u has landed in LDS
      __syncthreads();                                   // ... and everyone's; and every lane is done reading the other buffer (tile u - 1)
      if (u + 1 < u1) dma(u + 1, buf ^ 1u);
      if (fetch_next && live && j + 1 < ncol) raw_next = ld_fq(scalar_ptr(j + 1));
      if (live && m != 0) {
        const uint4* e = reinterpret_cast<const uint4*>(lds + buf * sub_bytes + (size_t)(m - 1) * 96u);
        uint4 a0 = e[0], a1 = e[1], a2 = e[2], a3 = e[3], a4 = e[4], a5 = e[5];
        Niels n;
        n.yp = Fp{{(uint64_t)a0.x | ((uint64_t)a0.y << 32), (uint64_t)a0.z | ((uint64_t)a0.w << 32), (uint64_t)a1.x | ((uint64_t)a1.y << 32), (uint64_t)a1.z | ((uint64_t)a1.w << 32)}};
        n.ym = Fp{{(uint64_t)a2.x | ((uint64_t)a2.y << 32), (uint64_t)a2.z | ((uint64_t)a2.w << 32), (uint64_t)a3.x | ((uint64_t)a3.y << 32), (uint64_t)a3.z | ((uint64_t)a3.w << 32)}};
        n.t2d = Fp{{(uint64_t)a4.x | ((uint64_t)a4.y << 32), (uint64_t)a4.z | ((uint64_t)a4.w << 32), (uint64_t)a5.x | ((uint64_t)a5.y << 32), (uint64_t)a5.z | ((uint64_t)a5.w << 32)}};
        acc = pt_madd(acc, n, d < 0);
      }
    }
    __syncthreads();  // persistent form: the next run's first DMA must not overtake this run's last reads
  }
  if (live) A.partial[row * A.nb + bk] = acc;
}

__global__ void __launch_bounds__(1024) k_msm_lds(MsmLdsArgs A) {
  extern __shared__ __attribute__((aligned(16))) uint8_t msm_lds_buf[];
  for (unsigned wg = blockIdx.x; wg < A.n_wg; wg += gridDim.x) msm_lds_run(A, wg, msm_lds_buf);
}

// ------------------------------------------------------------------------------------------------ host side
// rows per workgroup and row-blocks of an LDS-form launch
static void msm_lds_shape(size_t rows, unsigned* nrb, unsigned* rows_per_wg, unsigned* threads) {
  size_t b = (rows + 1023) / 1024;
  size_t per = (rows + b - 1) / b;
  size_t t = (per + 63) / 64 * 64;
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
  size_t lds = 2 * (size_t)A.tent * 96;
  if (lds < 81920 && grid_limit) lds = 81920;  // background share: more than half of a CU's LDS, so that a CU never holds two of these
  static std::atomic<uint64_t> attr_set{0};  // per device: the kernel may claim more than the default 64 KB of dynamic LDS
  const uint64_t bit = 1ull << (c->dev & 63);
  if (!(attr_set.load() & bit)) {
    (void)hipFuncSetAttribute((const void*)k_msm_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set.fetch_or(bit);
  }
  hipLaunchKernelGGL(k_msm_lds, dim3(grid), dim3(thr), (unsigned)lds, st, A);
}
