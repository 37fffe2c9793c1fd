// spartan_amd: F_q streaming kernels (eq tables, sum-check evaluate/bind, vector-matrix, dot products).
#include "internal.hpp"

// ------------------------------------------------------------------------------------------------ F_q streaming kernels
// chi table (EqPolynomial::evals, dense_mlpoly.rs:68-84). r[0] <-> most significant index bit. Thread t owns the 2^TOPB
// entries whose low (ell - TOPB) index bits equal t: it multiplies out the factors of those low variables once (a serial
// chain), then doubles over the TOPB top variables in registers. Entry k of a thread sits at k * 2^(ell-TOPB) + t, so every
// store instruction of a wave covers 64 consecutive scalars (2 KiB, fully coalesced).
constexpr int EQ_TOPB = 4;
__device__ __forceinline__ Fq eq_suffix(const Fq* __restrict__ r, size_t ell, int topb, size_t lo) {
  Fq acc = fq_one();
  int nl = (int)ell - topb;
  for (int k = 0; k < nl; k++) {
    Fq rk = ld_fq(r + topb + k);
    bool bit = (lo >> (nl - 1 - k)) & 1;
    acc = fq_mul(acc, bit ? rk : fq_sub(fq_one(), rk));
  }
  return acc;
}
template <int TOPB>
__device__ __forceinline__ void eq_expand_top(Fq (&v)[1 << TOPB], const Fq* __restrict__ r, const Fq& suffix) {
  v[0] = suffix;
#pragma unroll
  for (int k = 0; k < TOPB; k++) {
    Fq rk = ld_fq(r + k);
#pragma unroll
    for (int i = (1 << k) - 1; i >= 0; i--) {
      Fq hi = fq_mul(v[i], rk);
      v[2 * i + 1] = hi;
      v[2 * i] = fq_sub(v[i], hi);
    }
  }
}
// Short tables (ell <= EQ_SMALL_ELL) are pure latency, and there the SIZE of the code is what costs: an earlier kernel
// that expanded 16 entries per thread in registers was 59 KB of straight-line code, which a lone wavefront fetches cold
// (~20 us measured, whatever ell is). The kernel below is a few KB of rolled code; r arrives in the host-mapped page and is
// copied to LDS once per block.
constexpr size_t EQ_SMALL_ELL = 13;
// One thread per entry, but not one chain of ell multiplications per thread (a lone wavefront needs 1-2 us per dependent
// F_q multiplication: 13 of them were ~19 us per table, on the critical path of every product-circuit layer). The factors
// are grouped: the low 8 index bits split 4 + 4 into two 16-entry tables built by 32 threads (3 multiplications deep),
// the block's high bits (<= 5) are multiplied out by one more thread, then hi*ta (16 threads) and one multiplication per
// entry: at most 6 deep. The product of the same factors in another order is the same field element.
// The three product chains (the two 16-entry tables and the block's high bits) run in three different wavefronts: as branches of ONE
// wavefront they were executed one after the other (10 dependent multiplications instead of 4: 14 -> ~8 us per table, 70+ tables per
// proof). The challenge vector travels in the kernel arguments (r_host == null) instead of being read from the host-mapped page.
struct EqR { Fq r[13]; };
__global__ void __launch_bounds__(256) k_eq_expand_small(const Fq* __restrict__ r_host, EqR rin, size_t ell, Fq* __restrict__ out) { SP_FG_PRIO();
  __shared__ Fq r[16];
  __shared__ Fq ta[16], tb[16];
  __shared__ Fq hi;
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  if (r_host) {
    if (t < (int)ell) r[t] = ld_fq(r_host + t);
  } else {
#pragma unroll
    for (int k = 0; k < 13; k++)
      if (t == k) r[k] = rin.r[k];
  }
  __syncthreads();
  const int nlo = ell < 8 ? (int)ell : 8, nb = nlo / 2, na = nlo - nb, nhi = (int)ell - nlo;
  auto factor = [&](int k, bool bit) {  // chi factor of variable k: r[k] or 1 - r[k]
    Fq rk = r[k], f = fq_sub(fq_one(), rk);
#pragma unroll
    for (int w = 0; w < 4; w++) f.l[w] = bit ? rk.l[w] : f.l[w];
    return f;
  };
  if (wave == 0 && lane < 16) {
    Fq acc = fq_one();
    if (lane < (1 << na))
      for (int k = 0; k < na; k++) { Fq f = factor(nhi + k, (lane >> (na - 1 - k)) & 1); acc = k ? fq_mul(acc, f) : f; }
    ta[lane] = acc;
  } else if (wave == 1 && lane < 16) {
    Fq acc = fq_one();
    if (lane < (1 << nb))
      for (int k = 0; k < nb; k++) { Fq f = factor(nhi + na + k, (lane >> (nb - 1 - k)) & 1); acc = k ? fq_mul(acc, f) : f; }
    tb[lane] = acc;
  } else if (wave == 2 && lane == 0) {
    Fq acc = fq_one();
    for (int k = 0; k < nhi; k++) { Fq f = factor(k, (blockIdx.x >> (nhi - 1 - k)) & 1); acc = k ? fq_mul(acc, f) : f; }
    hi = acc;
  }
  __syncthreads();
  if (nhi > 0 && t < (1 << na)) ta[t] = fq_mul(ta[t], hi);
  __syncthreads();
  size_t i = (size_t)blockIdx.x * 256 + t;
  if (i >> ell) return;
  Fq v = ta[(t >> nb) & ((1 << na) - 1)];
  if (nb > 0) v = fq_mul(v, tb[t & ((1 << nb) - 1)]);
  st_fq(out + i, v);
}

// Long tables: chi(r)[i] = chi(r_hi)[i >> lo] * chi(r_lo)[i & (2^lo - 1)] — two short tables (kernel above) and ONE
// multiplication per entry in a streaming kernel with a few hundred bytes of code, instead of the 59 KB unrolled kernel
// (whose first wave on every CU spends ~20 us fetching it). The product of the same factors in another order is the same
// field element, hence the same canonical limbs.
__global__ void __launch_bounds__(256) k_eq_outer(const Fq* __restrict__ hi, const Fq* __restrict__ lo, int lo_ell, size_t len, Fq* __restrict__ out) { SP_FG_PRIO();
  size_t mask = ((size_t)1 << lo_ell) - 1;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < len; i += (size_t)gridDim.x * blockDim.x)
    st_fq(out + i, fq_mul(ld_fq(hi + (i >> lo_ell)), ld_fq(lo + (i & mask))));
}
// <Z, chi(r)> without materialising chi; per-block partials.
template <int TOPB>
__global__ void __launch_bounds__(256) k_evaluate(const Fq* __restrict__ Z, const Fq* __restrict__ r_host, size_t ell, Fq* __restrict__ partials) { SP_FG_PRIO();
  __shared__ Fq sm[256];
  __shared__ Fq r[40];
  if (threadIdx.x < ell) r[threadIdx.x] = ld_fq(r_host + threadIdx.x);
  __syncthreads();
  size_t nthreads = (size_t)1 << (ell - TOPB);
  Fq acc[1] = {fq_zero()};
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < nthreads; t += (size_t)gridDim.x * blockDim.x) {
    Fq v[1 << TOPB];
    eq_expand_top<TOPB>(v, r, eq_suffix(r, ell, TOPB, t));
#pragma unroll
    for (int k = 0; k < (1 << TOPB); k++) acc[0] = fq_add(acc[0], fq_mul(v[k], ld_fq(Z + ((size_t)k << (ell - TOPB)) + t)));
  }
  block_sum_fq<1>(acc, sm);
  if (threadIdx.x == 0) st_fq(partials + blockIdx.x, acc[0]);
}

struct Tabs4 {
  Fq* p[4];
};
// sum-check evaluations at t = 0, 2, 3 of the line through (T[i], T[i+half]); per-block partial sums.
template <int KIND>
__device__ __forceinline__ void sc_point(const Fq& a0, const Fq& a1, const Fq& b0, const Fq& b1, const Fq& c0, const Fq& c1, const Fq& d0,
                                         const Fq& d1, Fq& e0, Fq& e2, Fq& e3) {
  Fq a2 = fq_sub(fq_dbl(a1), a0), b2 = fq_sub(fq_dbl(b1), b0);
  if (KIND == 0) {
    e0 = fq_add(e0, fq_mul(a0, b0));
    e2 = fq_add(e2, fq_mul(a2, b2));
    return;
  }
  Fq a3 = fq_sub(fq_add(a2, a1), a0), b3 = fq_sub(fq_add(b2, b1), b0);
  Fq c2 = fq_sub(fq_dbl(c1), c0), c3 = fq_sub(fq_add(c2, c1), c0);
  if (KIND == 1) {
    e0 = fq_add(e0, fq_mul(fq_mul(a0, b0), c0));
    e2 = fq_add(e2, fq_mul(fq_mul(a2, b2), c2));
    e3 = fq_add(e3, fq_mul(fq_mul(a3, b3), c3));
    return;
  }
  Fq d2 = fq_sub(fq_dbl(d1), d0), d3 = fq_sub(fq_add(d2, d1), d0);
  e0 = fq_add(e0, fq_mul(a0, fq_sub(fq_mul(b0, c0), d0)));
  e2 = fq_add(e2, fq_mul(a2, fq_sub(fq_mul(b2, c2), d2)));
  e3 = fq_add(e3, fq_mul(a3, fq_sub(fq_mul(b3, c3), d3)));
}
template <int KIND>
__global__ void __launch_bounds__(256) k_sc_eval(Tabs4 T, size_t half, Fq* __restrict__ partials) { SP_FG_PRIO();
  __shared__ Fq sm[256];
  Fq e[3] = {fq_zero(), fq_zero(), fq_zero()};
  Fq z = fq_zero();
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < half; i += (size_t)gridDim.x * blockDim.x) {
    Fq a0 = ld_fq(T.p[0] + i), a1 = ld_fq(T.p[0] + half + i), b0 = ld_fq(T.p[1] + i), b1 = ld_fq(T.p[1] + half + i);
    Fq c0 = z, c1 = z, d0 = z, d1 = z;
    if (KIND >= 1) { c0 = ld_fq(T.p[2] + i); c1 = ld_fq(T.p[2] + half + i); }
    if (KIND == 2) { d0 = ld_fq(T.p[3] + i); d1 = ld_fq(T.p[3] + half + i); }
    sc_point<KIND>(a0, a1, b0, b1, c0, c1, d0, d1, e[0], e[1], e[2]);
  }
  block_sum_fq<3>(e, sm);
  if (threadIdx.x == 0) {
    st_fq(partials + 3 * blockIdx.x + 0, e[0]);
    st_fq(partials + 3 * blockIdx.x + 1, e[1]);
    st_fq(partials + 3 * blockIdx.x + 2, e[2]);
  }
}
// fused bind(r) + evaluate next round. quarter = len/4. Thread i < quarter reads T[i], T[i+q], T[i+2q], T[i+3q],
// writes the bound values T'[i] = T[i] + r (T[i+2q]-T[i]) and T'[i+q], and evaluates the round on (T'[i], T'[i+q]).
template <int KIND>
__global__ void __launch_bounds__(256) k_sc_bind_eval(Tabs4 T, size_t quarter, Fq r, Fq* __restrict__ partials) { SP_FG_PRIO();
  __shared__ Fq sm[256];
  constexpr int NT = KIND == 0 ? 2 : (KIND == 1 ? 3 : 4);
  Fq e[3] = {fq_zero(), fq_zero(), fq_zero()};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < quarter; i += (size_t)gridDim.x * blockDim.x) {
    Fq lo[4], hi[4];
#pragma unroll
    for (int k = 0; k < NT; k++) {
      Fq x0 = ld_fq(T.p[k] + i), x1 = ld_fq(T.p[k] + quarter + i), x2 = ld_fq(T.p[k] + 2 * quarter + i), x3 = ld_fq(T.p[k] + 3 * quarter + i);
      lo[k] = fq_add(x0, fq_mul(r, fq_sub(x2, x0)));
      hi[k] = fq_add(x1, fq_mul(r, fq_sub(x3, x1)));
      st_fq(T.p[k] + i, lo[k]);
      st_fq(T.p[k] + quarter + i, hi[k]);
    }
    Fq z = fq_zero();
    sc_point<KIND>(lo[0], hi[0], lo[1], hi[1], NT > 2 ? lo[2] : z, NT > 2 ? hi[2] : z, NT > 3 ? lo[3] : z, NT > 3 ? hi[3] : z, e[0], e[1], e[2]);
  }
  block_sum_fq<3>(e, sm);
  if (threadIdx.x == 0) {
    st_fq(partials + 3 * blockIdx.x + 0, e[0]);
    st_fq(partials + 3 * blockIdx.x + 1, e[1]);
    st_fq(partials + 3 * blockIdx.x + 2, e[2]);
  }
}
// the same for any number of tables of one length (pointer list in the host-mapped page); heads != nullptr with half == 1:
// the bound value (the table's only remaining entry) also goes to the result area — bound_poly_var_top + [0] in one launch
__global__ void __launch_bounds__(256) k_bind_top_list(Fq* const* __restrict__ ptrs, size_t ntabs, size_t half, Fq r, Fq* __restrict__ heads, DoneSig sig) { SP_FG_PRIO();
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < ntabs * half; idx += (size_t)gridDim.x * blockDim.x) {
    size_t t = idx / half, i = idx % half;
    Fq* p = ptrs[t];
    Fq x0 = ld_fq(p + i), x1 = ld_fq(p + half + i);
    Fq v = fq_add(x0, fq_mul(r, fq_sub(x1, x0)));
    st_fq(p + i, v);
    if (heads) st_fq(heads + t, v);
  }
  signal_done(sig);
}
// Latency form of k_sc_bind_eval for short tables (quarter <= 8192), kinds 0 (A*B) and 2 (A*(B*C-D)): as in
// k_cubic_bind_eval_tiny (spark.hip) the multiplications of one index are spread over 8 lanes — lane 2k+h binds half h of
// table k, then lanes 0..2 evaluate t = 0, 2, 3 with one instruction stream (operands chosen by selects). Block = 32
// indices x 8 lanes; partials[blk][3].
template <int KIND>
__global__ void __launch_bounds__(256) k_sc_bind_eval_tiny(Tabs4 T, size_t quarter, Fq r, Fq* __restrict__ partials, DoneSig sig) { SP_FG_PRIO();
  constexpr int NT = KIND == 0 ? 2 : 4;
  __shared__ Fq bound[32][8];  // [index][table*2 + half]
  __shared__ Fq red[3][32];
  int li = threadIdx.x >> 3, role = threadIdx.x & 7;
  size_t i = (size_t)blockIdx.x * 32 + li;
  bool live = i < quarter;
  if (role < 2 * NT && live) {
    int k = role >> 1, half = role & 1;
    Fq* ptr = T.p[k];
    Fq x0 = ld_fq(ptr + (size_t)half * quarter + i), x2 = ld_fq(ptr + (size_t)(2 + half) * quarter + i);
    Fq v = fq_add(x0, fq_mul(r, fq_sub(x2, x0)));
    bound[li][role] = v;
    st_fq(ptr + (size_t)half * quarter + i, v);
  }
  __syncthreads();
  if (role < 3) {
    Fq e = fq_zero();
    if (live && (KIND == 2 || role < 2)) {
      // value of each table's line at this lane's point: t = 0 -> x0 ; t = 2 -> 2 x1 - x0 ; t = 3 -> 3 x1 - 2 x0
      Fq pt[4];
#pragma unroll
      for (int k = 0; k < NT; k++) {
        Fq x0 = bound[li][2 * k], x1 = bound[li][2 * k + 1];
        Fq x2 = fq_sub(fq_dbl(x1), x0), x3 = fq_sub(fq_add(x2, x1), x0);
#pragma unroll
        for (int w = 0; w < 4; w++) pt[k].l[w] = role == 0 ? x0.l[w] : (role == 1 ? x2.l[w] : x3.l[w]);
      }
      if (KIND == 0) e = fq_mul(pt[0], pt[1]);
      else e = fq_mul(pt[0], fq_sub(fq_mul(pt[1], pt[2]), pt[3]));
    }
    red[role][li] = e;
  }
  __syncthreads();
  for (int s = 16; s > 0; s >>= 1) {
    if (role < 3 && li < s) red[role][li] = fq_add(red[role][li], red[role][li + s]);
    __syncthreads();
  }
  if (threadIdx.x < 3) st_fq(partials + (size_t)blockIdx.x * 3 + threadIdx.x, red[threadIdx.x][0]);
  signal_done(sig);
}
__global__ void __launch_bounds__(256) k_bind_top(Tabs4 T, int ntabs, size_t half, Fq r) { SP_FG_PRIO();
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < half; i += (size_t)gridDim.x * blockDim.x) {
    for (int k = 0; k < ntabs; k++) {
      Fq x0 = ld_fq(T.p[k] + i), x1 = ld_fq(T.p[k] + half + i);
      st_fq(T.p[k] + i, fq_add(x0, fq_mul(r, fq_sub(x1, x0))));
    }
  }
}
// partials[nblk][K] -> out[K] ; single block
__global__ void __launch_bounds__(256) k_reduce_partials(const Fq* __restrict__ partials, size_t nblk, int K, Fq* __restrict__ out, DoneSig sig) { SP_FG_PRIO();
  __shared__ Fq sm[256];
  for (int k = 0; k < K; k++) {
    Fq acc[1] = {fq_zero()};
    for (size_t b = threadIdx.x; b < nblk; b += 256) acc[0] = fq_add(acc[0], ld_fq(partials + b * K + k));
    block_sum_fq<1>(acc, sm);
    if (threadIdx.x == 0) st_fq(out + k, acc[0]);
  }
  signal_done(sig);
}
// DensePolynomial::bound (dense_mlpoly.rs:206-213): out[i] = sum_j L[j] * Z[j*R + i], in two stages.
// Stage 1, grid (ceil(R/64), nchunks): a block covers 64 columns x one chunk of rows; its 256 threads are 64 columns x 4 row
// lanes (a wave reads 2 KiB of one row: coalesced), each thread multiplies its few rows (the row chunks are sized so that a
// thread has 4..8 rows and the launch has thousands of blocks, also for the 1024 x 1024 witness), then the 4 row lanes are
// added in LDS -> partial[chunk][col]. Stage 2, grid ceil(R/32): 32 columns x 8 chunk lanes per block add the chunks.
__global__ void __launch_bounds__(256) k_vecmat(const Fq* __restrict__ L, size_t Lsz, const Fq* __restrict__ Z, size_t R, size_t jchunk,
                                                Fq* __restrict__ partial) { SP_FG_PRIO();
  __shared__ Fq sm[4][64];
  const int cl = threadIdx.x & 63, g = threadIdx.x >> 6;
  size_t i = (size_t)blockIdx.x * 64 + cl;
  size_t j0 = (size_t)blockIdx.y * jchunk, j1 = j0 + jchunk;
  if (j1 > Lsz) j1 = Lsz;
  Fq acc = fq_zero();
  if (i < R)
    for (size_t j = j0 + g; j < j1; j += 4) acc = fq_add(acc, fq_mul(ld_fq(L + j), ld_fq(Z + j * R + i)));
  sm[g][cl] = acc;
  __syncthreads();
  if (g == 0 && i < R) st_fq(partial + (size_t)blockIdx.y * R + i, fq_add(fq_add(sm[0][cl], sm[1][cl]), fq_add(sm[2][cl], sm[3][cl])));
}
__global__ void __launch_bounds__(256) k_colsum(const Fq* __restrict__ partial, size_t nchunks, size_t R, Fq* __restrict__ out) { SP_FG_PRIO();
  __shared__ Fq sm[8][32];
  const int cl = threadIdx.x & 31, g = threadIdx.x >> 5;
  size_t i = (size_t)blockIdx.x * 32 + cl;
  Fq acc = fq_zero();
  if (i < R)
    for (size_t c = g; c < nchunks; c += 8) acc = fq_add(acc, ld_fq(partial + c * R + i));
  sm[g][cl] = acc;
  __syncthreads();
  for (int s = 4; s > 0; s >>= 1) {
    if (g < s) sm[g][cl] = fq_add(sm[g][cl], sm[g + s][cl]);
    __syncthreads();
  }
  if (g == 0 && i < R) st_fq(out + i, sm[0][cl]);
}
// rows per chunk of stage 1: 4 row lanes x (4 rows for small matrices, 8 for large ones)
static size_t vecmat_jchunk(size_t Lsz, size_t R) { return Lsz * R <= ((size_t)1 << 22) ? 16 : 32; }
__global__ void __launch_bounds__(256) k_dot(const Fq* __restrict__ a, const Fq* __restrict__ b, size_t n, Fq* __restrict__ partials) { SP_FG_PRIO();
  __shared__ Fq sm[256];
  Fq acc[1] = {fq_zero()};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    acc[0] = fq_add(acc[0], fq_mul(ld_fq(a + i), ld_fq(b + i)));
  block_sum_fq<1>(acc, sm);
  if (threadIdx.x == 0) st_fq(partials + blockIdx.x, acc[0]);
}
// out[k*count + e] = *(ptrs[k] + e): pointer list read from the host-mapped page
__global__ void k_gather_elems(const Fq* const* __restrict__ ptrs, size_t n, size_t count, Fq* __restrict__ out) { SP_FG_PRIO();
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n * count) st_fq(out + t, ld_fq(ptrs[t / count] + t % count));
}


// partials[nblk][K] -> out[K]. A few hundred partials at most are written by the kernel straight into the host-mapped
// result page and added by the calling thread (no second launch); longer lists are added by k_reduce_partials.
static void host_sum(const Fq* p, size_t nblk, int K, uint64_t* out) {
  Fq* o = (Fq*)out;
  for (int k = 0; k < K; k++) {
    Fq acc = p[k];
    for (size_t b = 1; b < nblk; b++) acc = fq_add(acc, p[b * K + k]);
    o[k] = acc;
  }
}
int32_t reduce_and_fetch(sp_ctx* c, Fq* partials, size_t nblk, int K, uint64_t* out) {
  if (partials != (Fq*)hres(c)) {
    DoneSig sig = sig_make(c, 1);
    {
      ProfScope ps(c, PF_REDUCE, 32.0 * (double)(nblk * K));
      hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(256), 0, c->stream, (const Fq*)partials, nblk, K, (Fq*)hres(c), sig);
    }
    SPCHK(sig_wait(c, sig));
    memcpy(out, hres(c), 32 * K);
    return hipGetLastError() == hipSuccess ? SP_OK : SP_EHIP;
  }
  SPCHK(sync_spin(c));
  if (hipGetLastError() != hipSuccess) return SP_EHIP;
  host_sum((const Fq*)hres(c), nblk, K, out);
  return SP_OK;
}

extern "C" {

// (a one-chain-per-entry kernel, ell multiplications deep, measured the same per proof as this 4+4 product form: removed)
// r_host: the same ell scalars on the host (ell <= 13: they go into the kernel arguments); dr: their copy in the host-mapped page (the fallback)
static void launch_eq_small(sp_ctx* c, const Fq* dr, const uint64_t* r_host, size_t ell, Fq* out) {
  size_t len = (size_t)1 << ell;
  const bool inline_args = c->opt.v[OPT_SUMCHECK_INLINE_ARGS] != 0;  // A/B switch
  EqR in;
  const bool inl = inline_args && r_host && ell <= 13;
  if (inl) memcpy(in.r, r_host, 32 * ell);
  hipLaunchKernelGGL(k_eq_expand_small, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, c->stream, inl ? (const Fq*)nullptr : dr, in, ell, out);
}
int32_t sp_eq_expand(sp_ctx* c, const uint64_t* r, size_t ell, sp_table** out) {
  if (!c || !r || !out || ell == 0 || ell > 40) return SP_EINVAL;
  HIPCHK(hipSetDevice(c->dev));
  // the challenge vector goes into the next slot of a small ring and the call returns without waiting; a slot is reused
  // only after a completed wait on the stream (sync_epoch moved on), which guarantees its kernel has read it
  static_assert(EQ_SLOTS == 8 && 32 * 40 <= EQ_SLOT_BYTES, "eq ring layout");
  unsigned slot = c->eq_next;
  c->eq_next = (slot + 1) % EQ_SLOTS;
  if (c->eq_slot_epoch[slot] == c->sync_epoch + 1) SPCHK(sync_spin(c));
  const Fq* dr = (const Fq*)stage_small(c, HMAP_GEN + slot * EQ_SLOT_BYTES, r, 32 * ell);
  c->eq_slot_epoch[slot] = c->sync_epoch + 1;
  size_t len = (size_t)1 << ell;
  SPCHK(table_new(c, len, false, out));
  {
    ProfScope ps(c, PF_EQ_EXPAND, 32.0 * (double)len);
    dim3 blk(256);
    if (ell <= EQ_SMALL_ELL) {
      launch_eq_small(c, dr, r, ell, (*out)->d);
    } else {
      size_t hi_ell = ell - ell / 2, lo_ell = ell / 2, nhi = (size_t)1 << hi_ell, nlo = (size_t)1 << lo_ell;
      Fq* tmp = nullptr;  // [chi(r_hi) | chi(r_lo)]; handed back to the pool right away: reuse is ordered by the stream
      int32_t rc = pool_alloc(c, 32 * (nhi + nlo), (void**)&tmp);
      if (rc != SP_OK) { sp_table_free(*out); *out = nullptr; return rc; }
      launch_eq_small(c, dr, r, hi_ell, tmp);
      launch_eq_small(c, dr + hi_ell, r + 4 * hi_ell, lo_ell, tmp + nhi);
      hipLaunchKernelGGL(k_eq_outer, dim3((unsigned)grid_for(len, 4096)), blk, 0, c->stream, (const Fq*)tmp, (const Fq*)(tmp + nhi), (int)lo_ell, len,
                         (*out)->d);
      pool_release(c, tmp, 32 * (nhi + nlo));
    }
  }
  return hipGetLastError() == hipSuccess ? SP_OK : SP_EHIP;
}

static int32_t tabs_check(sp_ctx* c, sp_table* const* tabs, size_t ntabs, size_t need, Tabs4* T, size_t* len) {
  if (!c || !tabs || ntabs != need) return SP_EINVAL;
  size_t l = tabs[0] ? tabs[0]->len : 0;
  for (size_t k = 0; k < ntabs; k++) {
    if (!tabs[k] || tabs[k]->len != l) return SP_EINVAL;
    T->p[k] = tabs[k]->d;
  }
  for (size_t k = ntabs; k < 4; k++) T->p[k] = nullptr;
  if (l < 2 || !is_pow2(l)) return SP_EINVAL;
  *len = l;
  return SP_OK;
}
int32_t sp_sumcheck_eval(sp_ctx* c, int kind, sp_table* const* tabs, size_t ntabs, uint64_t* out_evals) {
  if (kind < 0 || kind > 2 || !out_evals) return SP_EINVAL;
  Tabs4 T;
  size_t len;
  SPCHK(tabs_check(c, tabs, ntabs, kind == 0 ? 2 : (kind == 1 ? 3 : 4), &T, &len));
  HIPCHK(hipSetDevice(c->dev));
  size_t half = len / 2, nblk = half <= 256 ? 1 : grid_for(half, 1024);
  SPCHK(ensure(&c->scratch, &c->scratch_cap, 32 * (nblk * 3 + 3)));
  Fq* partials = partials_dst(c, nblk, 3);
  {
    ProfScope ps(c, PF_SC_EVAL, 32.0 * (double)len * (double)ntabs, nullptr, (kind == 0 ? 2.0 : 6.0) * (double)half);
    if (kind == 0) hipLaunchKernelGGL(k_sc_eval<0>, dim3((unsigned)nblk), dim3(256), 0, c->stream, T, half, partials);
    if (kind == 1) hipLaunchKernelGGL(k_sc_eval<1>, dim3((unsigned)nblk), dim3(256), 0, c->stream, T, half, partials);
    if (kind == 2) hipLaunchKernelGGL(k_sc_eval<2>, dim3((unsigned)nblk), dim3(256), 0, c->stream, T, half, partials);
  }
  uint64_t e[12];
  SPCHK(reduce_and_fetch(c, partials, nblk, 3, e));
  memcpy(out_evals, e, 32);
  memcpy(out_evals + 4, e + 4, 32);
  if (kind != 0) memcpy(out_evals + 8, e + 8, 32);
  return SP_OK;
}
int32_t sp_table_bind_top(sp_ctx* c, sp_table* const* tabs, size_t ntabs, const uint64_t r[4]) {
  if (!c || !tabs || !r || ntabs == 0) return SP_EINVAL;
  HIPCHK(hipSetDevice(c->dev));
  Fq rr;
  memcpy(rr.l, r, 32);
  for (size_t k0 = 0; k0 < ntabs; k0 += 4) {
    size_t nk = ntabs - k0 < 4 ? ntabs - k0 : 4;
    Tabs4 T = {{nullptr, nullptr, nullptr, nullptr}};
    size_t len = tabs[k0] ? tabs[k0]->len : 0;
    for (size_t k = 0; k < nk; k++) {
      if (!tabs[k0 + k] || tabs[k0 + k]->len != len) return SP_EINVAL;
      T.p[k] = tabs[k0 + k]->d;
    }
    if (len < 2 || !is_pow2(len)) return SP_EINVAL;
    size_t half = len / 2;
    {
      ProfScope ps(c, PF_SC_BIND, 48.0 * (double)len * (double)nk);
      hipLaunchKernelGGL(k_bind_top, dim3((unsigned)grid_for(half)), dim3(256), 0, c->stream, T, (int)nk, half, rr);
    }
    for (size_t k = 0; k < nk; k++) tabs[k0 + k]->len = half;
  }
  return hipGetLastError() == hipSuccess ? SP_OK : SP_EHIP;
}
static Fq limbs4(const uint64_t* p) {
  Fq x;
  memcpy(x.l, p, 32);
  return x;
}
static int32_t bind_top_list(sp_ctx* c, sp_table* const* tabs, size_t ntabs, const uint64_t r[4], uint64_t* out_heads) {
  if (!c || !tabs || !r || ntabs == 0 || 8 * ntabs > HMAP_GEN || 32 * ntabs > HMAP_SIZE - HMAP_IN) return SP_EINVAL;
  size_t len = tabs[0] ? tabs[0]->len : 0;
  if (len < 2 || !is_pow2(len) || (out_heads && len != 2)) return SP_EINVAL;
  std::vector<Fq*> ptrs(ntabs);
  for (size_t k = 0; k < ntabs; k++) {
    if (!tabs[k] || tabs[k]->len != len) return SP_EINVAL;
    for (size_t m = 0; m < k; m++)
      if (tabs[m] == tabs[k]) return SP_EINVAL;  // a table listed twice would be bound twice
    ptrs[k] = tabs[k]->d;
  }
  HIPCHK(hipSetDevice(c->dev));
  Fq* const* dp = (Fq* const*)stage_small(c, 0, ptrs.data(), 8 * ntabs);
  size_t half = len / 2;
  DoneSig sig = sig_make(c, grid_for(ntabs * half));
  {
    ProfScope ps(c, PF_SC_BIND, 48.0 * (double)len * (double)ntabs);
    hipLaunchKernelGGL(k_bind_top_list, dim3((unsigned)grid_for(ntabs * half)), dim3(256), 0, c->stream, dp, ntabs, half, limbs4(r),
                       out_heads ? (Fq*)hres(c) : (Fq*)nullptr, sig);
  }
  for (size_t k = 0; k < ntabs; k++) tabs[k]->len = half;
  SPCHK(sig_wait(c, sig));  // also without heads: the pointer list sits in the shared input page
  if (out_heads) memcpy(out_heads, hres(c), 32 * ntabs);
  return hipGetLastError() == hipSuccess ? SP_OK : SP_EHIP;
}
int32_t sp_table_bind_top_heads(sp_ctx* c, sp_table* const* tabs, size_t ntabs, const uint64_t r[4], uint64_t* out_heads) {
  if (!out_heads) return SP_EINVAL;
  return bind_top_list(c, tabs, ntabs, r, out_heads);
}
int32_t sp_sumcheck_bind_eval(sp_ctx* c, int kind, sp_table* const* tabs, size_t ntabs, const uint64_t r[4], uint64_t* out_evals) {
  if (kind < 0 || kind > 2 || !out_evals || !r) return SP_EINVAL;
  Tabs4 T;
  size_t len;
  SPCHK(tabs_check(c, tabs, ntabs, kind == 0 ? 2 : (kind == 1 ? 3 : 4), &T, &len));
  if (len < 4) return SP_EINVAL;
  HIPCHK(hipSetDevice(c->dev));
  Fq rr;
  memcpy(rr.l, r, 32);
  size_t quarter = len / 4;
  bool tiny = kind != 1 && quarter <= 8192;  // latency-bound rounds: one index per 8 lanes
  size_t nblk = tiny ? (quarter + 31) / 32 : grid_for(quarter, 1024);
  SPCHK(ensure(&c->scratch, &c->scratch_cap, 32 * (nblk * 3 + 3)));
  Fq* partials = partials_dst(c, nblk, 3);
  {
    ProfScope ps(c, PF_SC_BIND_EVAL, 48.0 * (double)len * (double)ntabs, nullptr, (kind == 0 ? 6.0 : (kind == 1 ? 12.0 : 14.0)) * (double)quarter);
    if (tiny && kind == 0) hipLaunchKernelGGL(k_sc_bind_eval_tiny<0>, dim3((unsigned)nblk), dim3(256), 0, c->stream, T, quarter, rr, partials, sig_none());
    else if (tiny) hipLaunchKernelGGL(k_sc_bind_eval_tiny<2>, dim3((unsigned)nblk), dim3(256), 0, c->stream, T, quarter, rr, partials, sig_none());
    else if (kind == 0) hipLaunchKernelGGL(k_sc_bind_eval<0>, dim3((unsigned)nblk), dim3(256), 0, c->stream, T, quarter, rr, partials);
    else if (kind == 1) hipLaunchKernelGGL(k_sc_bind_eval<1>, dim3((unsigned)nblk), dim3(256), 0, c->stream, T, quarter, rr, partials);
    else hipLaunchKernelGGL(k_sc_bind_eval<2>, dim3((unsigned)nblk), dim3(256), 0, c->stream, T, quarter, rr, partials);
  }
  for (size_t k = 0; k < ntabs; k++) tabs[k]->len = len / 2;
  uint64_t e[12];
  SPCHK(reduce_and_fetch(c, partials, nblk, 3, e));
  memcpy(out_evals, e, 32);
  memcpy(out_evals + 4, e + 4, 32);
  if (kind != 0) memcpy(out_evals + 8, e + 8, 32);
  return SP_OK;
}
// The same round in two halves: _start queues the bind and the evaluation and returns; _collect waits for the sums. The
// caller (the ZK sum-check of the host driver) computes the round's commitments on its own core in between. No other call
// on this context may come between the two: the sums travel through the context's result page.
int32_t sp_sumcheck_bind_eval_start(sp_ctx* c, int kind, sp_table* const* tabs, size_t ntabs, const uint64_t r[4]) {
  if (kind < 0 || kind > 2 || !r) return SP_EINVAL;
  Tabs4 T;
  size_t len;
  SPCHK(tabs_check(c, tabs, ntabs, kind == 0 ? 2 : (kind == 1 ? 3 : 4), &T, &len));
  if (len < 4 || c->pend_eval.active) return SP_EINVAL;
  HIPCHK(hipSetDevice(c->dev));
  Fq rr;
  memcpy(rr.l, r, 32);
  size_t quarter = len / 4;
  bool tiny = kind != 1 && quarter <= 8192;
  size_t nblk = tiny ? (quarter + 31) / 32 : grid_for(quarter, 1024);
  SPCHK(ensure(&c->scratch, &c->scratch_cap, 32 * (nblk * 3 + 3)));
  Fq* partials = tiny ? partials_dst(c, nblk, 3) : (Fq*)c->scratch;
  const bool on_host = partials == (Fq*)hres(c);
  DoneSig sig = sig_make(c, on_host ? nblk : 1);
  {
    ProfScope ps(c, PF_SC_BIND_EVAL, 48.0 * (double)len * (double)ntabs, nullptr, (kind == 0 ? 6.0 : (kind == 1 ? 12.0 : 14.0)) * (double)quarter);
    if (tiny && kind == 0) hipLaunchKernelGGL(k_sc_bind_eval_tiny<0>, dim3((unsigned)nblk), dim3(256), 0, c->stream, T, quarter, rr, partials, on_host ? sig : sig_none());
    else if (tiny) hipLaunchKernelGGL(k_sc_bind_eval_tiny<2>, dim3((unsigned)nblk), dim3(256), 0, c->stream, T, quarter, rr, partials, on_host ? sig : sig_none());
    else if (kind == 0) hipLaunchKernelGGL(k_sc_bind_eval<0>, dim3((unsigned)nblk), dim3(256), 0, c->stream, T, quarter, rr, partials);
    else if (kind == 1) hipLaunchKernelGGL(k_sc_bind_eval<1>, dim3((unsigned)nblk), dim3(256), 0, c->stream, T, quarter, rr, partials);
    else hipLaunchKernelGGL(k_sc_bind_eval<2>, dim3((unsigned)nblk), dim3(256), 0, c->stream, T, quarter, rr, partials);
  }
  for (size_t k = 0; k < ntabs; k++) tabs[k]->len = len / 2;
  if (!on_host) {
    ProfScope ps(c, PF_REDUCE, 32.0 * (double)(nblk * 3));
    hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(256), 0, c->stream, (const Fq*)partials, nblk, 3, (Fq*)hres(c), sig);
  }
  c->pend_eval.active = true;
  c->pend_eval.kind = kind;
  c->pend_eval.nblk = nblk;
  c->pend_eval.on_host = on_host;
  c->pend_eval.seq = sig.flag ? sig.seq : sync_post(c);
  return SP_OK;
}
int32_t sp_sumcheck_bind_eval_collect(sp_ctx* c, uint64_t* out_evals) {
  if (!c || !out_evals || !c->pend_eval.active) return SP_EINVAL;
  c->pend_eval.active = false;
  SPCHK(sync_wait(c, c->pend_eval.seq));
  if (hipGetLastError() != hipSuccess) return SP_EHIP;
  uint64_t e[12];
  if (c->pend_eval.on_host) host_sum((const Fq*)hres(c), c->pend_eval.nblk, 3, e);
  else memcpy(e, hres(c), 96);
  memcpy(out_evals, e, 32);
  memcpy(out_evals + 4, e + 4, 32);
  if (c->pend_eval.kind != 0) memcpy(out_evals + 8, e + 8, 32);
  return SP_OK;
}
int32_t sp_sumcheck_bind_eval_commit(sp_ctx* c, int kind, sp_table* const* tabs, size_t ntabs, const uint64_t r[4], uint64_t* out_evals,
                                     const sp_gens* g, const uint32_t* idx, size_t cols, const uint64_t* S, size_t rows, uint8_t* out_points) {
  if (kind < 0 || kind > 2 || !out_evals || !r || !out_points) return SP_EINVAL;
  Tabs4 T;
  size_t len;
  SPCHK(tabs_check(c, tabs, ntabs, kind == 0 ? 2 : (kind == 1 ? 3 : 4), &T, &len));
  if (len < 4) return SP_EINVAL;
  HIPCHK(hipSetDevice(c->dev));
  if (c->device_encode) {  // diagnostic mode (every encode on the GPU): the two halves one after the other
    SPCHK(sp_sumcheck_bind_eval(c, kind, tabs, ntabs, r, out_evals));
    return sp_msm_indexed(c, g, idx, cols, S, rows, out_points);
  }
  // the commitments do not depend on the tables: they run on the side stream while the main stream binds and evaluates
  constexpr size_t SUMS_OFF = HMAP_SIZE - HMAP_IN - 1024;  // evaluations / partial sums at the head of the result area, row sums at its tail
  SPCHK(msm_small_enqueue(c, c->stream_side, g, idx, cols, S, rows, hres(c) + SUMS_OFF));
  HIPCHK(hipEventRecord(c->side_ev, c->stream_side));
  Fq rr;
  memcpy(rr.l, r, 32);
  size_t quarter = len / 4;
  bool tiny = kind != 1 && quarter <= 8192;  // latency-bound rounds: one index per 8 lanes
  size_t nblk = tiny ? (quarter + 31) / 32 : grid_for(quarter, 1024);
  SPCHK(ensure(&c->scratch, &c->scratch_cap, 32 * (nblk * 3 + 3)));
  Fq* partials = partials_dst(c, nblk, 3);
  {
    ProfScope ps(c, PF_SC_BIND_EVAL, 48.0 * (double)len * (double)ntabs, nullptr, (kind == 0 ? 6.0 : (kind == 1 ? 12.0 : 14.0)) * (double)quarter);
    if (tiny && kind == 0) hipLaunchKernelGGL(k_sc_bind_eval_tiny<0>, dim3((unsigned)nblk), dim3(256), 0, c->stream, T, quarter, rr, partials, sig_none());
    else if (tiny) hipLaunchKernelGGL(k_sc_bind_eval_tiny<2>, dim3((unsigned)nblk), dim3(256), 0, c->stream, T, quarter, rr, partials, sig_none());
    else if (kind == 0) hipLaunchKernelGGL(k_sc_bind_eval<0>, dim3((unsigned)nblk), dim3(256), 0, c->stream, T, quarter, rr, partials);
    else if (kind == 1) hipLaunchKernelGGL(k_sc_bind_eval<1>, dim3((unsigned)nblk), dim3(256), 0, c->stream, T, quarter, rr, partials);
    else hipLaunchKernelGGL(k_sc_bind_eval<2>, dim3((unsigned)nblk), dim3(256), 0, c->stream, T, quarter, rr, partials);
  }
  for (size_t k = 0; k < ntabs; k++) tabs[k]->len = len / 2;
  if (partials != (Fq*)hres(c)) {
    ProfScope ps(c, PF_REDUCE, 32.0 * (double)(nblk * 3));
    hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(256), 0, c->stream, (const Fq*)partials, nblk, 3, (Fq*)hres(c), sig_none());
  }
  HIPCHK(hipStreamWaitEvent(c->stream, c->side_ev, 0));  // one completion for both streams
  SPCHK(sync_spin(c));
  if (hipGetLastError() != hipSuccess) return SP_EHIP;
  uint64_t e[12];
  if (partials == (Fq*)hres(c)) host_sum((const Fq*)hres(c), nblk, 3, e);
  else memcpy(e, hres(c), 96);
  memcpy(out_evals, e, 32);
  memcpy(out_evals + 4, e + 4, 32);
  if (kind != 0) memcpy(out_evals + 8, e + 8, 32);
  Pt sums[8];
  memcpy(sums, hres(c) + SUMS_OFF, sizeof(Pt) * rows);
  pt_compress_many(sums, rows, out_points);
  return SP_OK;
}
int32_t sp_vecmat(sp_ctx* c, const uint64_t* L, size_t Lsz, const sp_table* Z, uint64_t* out) {
  if (!c || !L || !Z || !out || Lsz == 0 || Z->len % Lsz) return SP_EINVAL;
  HIPCHK(hipSetDevice(c->dev));
  size_t R = Z->len / Lsz;
  SPCHK(ensure_dstage(c, 32 * Lsz));
  SPCHK(stage_in(c, 0, L, 32 * Lsz));
  size_t jchunk = vecmat_jchunk(Lsz, R), nchunks = (Lsz + jchunk - 1) / jchunk;
  SPCHK(ensure(&c->scratch, &c->scratch_cap, 32 * (nchunks * R + R)));
  Fq* partial = (Fq*)c->scratch;
  Fq* dres = partial + nchunks * R;
  {
    ProfScope ps(c, PF_VECMAT, 32.0 * (double)Z->len + 32.0 * (double)R, nullptr, (double)Z->len);
    hipLaunchKernelGGL(k_vecmat, dim3((unsigned)((R + 63) / 64), (unsigned)nchunks), dim3(256), 0, c->stream, (const Fq*)c->dstage, Lsz,
                       (const Fq*)Z->d, R, jchunk, partial);
    hipLaunchKernelGGL(k_colsum, dim3((unsigned)((R + 31) / 32)), dim3(256), 0, c->stream, (const Fq*)partial, nchunks, R, dres);
  }
  SPCHK(fetch_out(c, dres, out, 32 * R));
  return hipGetLastError() == hipSuccess ? SP_OK : SP_EHIP;
}
static int32_t vecmat_enqueue(sp_ctx* c, const Fq* dL, size_t Lsz, const sp_table* Z, sp_table** out) {
  size_t R = Z->len / Lsz;
  size_t jchunk = vecmat_jchunk(Lsz, R), nchunks = (Lsz + jchunk - 1) / jchunk;
  SPCHK(ensure(&c->scratch, &c->scratch_cap, 32 * (nchunks * R + R)));
  SPCHK(table_new(c, R, false, out));
  Fq* partial = (Fq*)c->scratch;
  {
    ProfScope ps(c, PF_VECMAT, 32.0 * (double)Z->len + 32.0 * (double)R, nullptr, (double)Z->len);
    hipLaunchKernelGGL(k_vecmat, dim3((unsigned)((R + 63) / 64), (unsigned)nchunks), dim3(256), 0, c->stream, dL, Lsz, (const Fq*)Z->d, R, jchunk, partial);
    hipLaunchKernelGGL(k_colsum, dim3((unsigned)((R + 31) / 32)), dim3(256), 0, c->stream, (const Fq*)partial, nchunks, R, (*out)->d);
  }
  return hipGetLastError() == hipSuccess ? SP_OK : SP_EHIP;
}
int32_t sp_vecmat_dev(sp_ctx* c, const uint64_t* L, size_t Lsz, const sp_table* Z, sp_table** out) {
  if (!c || !L || !Z || !out || Lsz == 0 || Z->len % Lsz) return SP_EINVAL;
  HIPCHK(hipSetDevice(c->dev));
  const void* dL = nullptr;
  SPCHK(vm_stage(c, L, 32 * Lsz, &dL));  // queued, not waited for: the caller goes on (the chi vector of the other half, the transcript)
  return vecmat_enqueue(c, (const Fq*)dL, Lsz, Z, out);
}
// The same with L already a device table (the chi vector of the left half generated by sp_eq_expand: PolyEvalProof::prove without
// blinds never needs it on the host). Queued, not waited for; L may be freed right after the call (its reuse is ordered by the stream).
int32_t sp_vecmat_tab(sp_ctx* c, const sp_table* L, const sp_table* Z, sp_table** out) {
  if (!c || !L || !Z || !out || L->len == 0 || Z->len % L->len) return SP_EINVAL;
  HIPCHK(hipSetDevice(c->dev));
  return vecmat_enqueue(c, (const Fq*)L->d, L->len, Z, out);
}
int32_t sp_dot(sp_ctx* c, const sp_table* a, size_t a_off, const sp_table* b, size_t b_off, size_t n, uint64_t out[4]) {
  if (!c || !a || !b || !out || n == 0 || a_off + n > a->cap || b_off + n > b->cap) return SP_EINVAL;
  HIPCHK(hipSetDevice(c->dev));
  size_t nblk = grid_for(n, 1024);
  SPCHK(ensure(&c->scratch, &c->scratch_cap, 32 * (nblk + 1)));
  Fq* partials = partials_dst(c, nblk, 1);
  {
    ProfScope ps(c, PF_DOT, 64.0 * (double)n, nullptr, (double)n);
    hipLaunchKernelGGL(k_dot, dim3((unsigned)nblk), dim3(256), 0, c->stream, (const Fq*)(a->d + a_off), (const Fq*)(b->d + b_off), n, partials);
  }
  return reduce_and_fetch(c, partials, nblk, 1, out);
}
int32_t sp_evaluate(sp_ctx* c, const sp_table* Z, const uint64_t* r, size_t ell, uint64_t out[4]) {
  if (!c || !Z || !r || !out || ell == 0 || ell > 40 || Z->len != ((size_t)1 << ell)) return SP_EINVAL;
  HIPCHK(hipSetDevice(c->dev));
  const Fq* dr = (const Fq*)stage_small(c, 0, r, 32 * ell);
  int topb = ell < (size_t)EQ_TOPB ? (int)ell : EQ_TOPB;
  size_t nthreads = Z->len >> topb, nblk = grid_for(nthreads, 1024);
  SPCHK(ensure(&c->scratch, &c->scratch_cap, 32 * (nblk + 1)));
  Fq* partials = partials_dst(c, nblk, 1);
  {
    ProfScope ps(c, PF_DOT, 32.0 * (double)Z->len);
    dim3 grid((unsigned)nblk), blk(256);
    const Fq* dz = (const Fq*)Z->d;
    switch (topb) {
      case 1: hipLaunchKernelGGL(k_evaluate<1>, grid, blk, 0, c->stream, dz, dr, ell, partials); break;
      case 2: hipLaunchKernelGGL(k_evaluate<2>, grid, blk, 0, c->stream, dz, dr, ell, partials); break;
      case 3: hipLaunchKernelGGL(k_evaluate<3>, grid, blk, 0, c->stream, dz, dr, ell, partials); break;
      default: hipLaunchKernelGGL(k_evaluate<EQ_TOPB>, grid, blk, 0, c->stream, dz, dr, ell, partials); break;
    }
  }
  return reduce_and_fetch(c, partials, nblk, 1, out);
}
static int32_t gather_elems(sp_ctx* c, sp_table* const* tabs, const size_t* offs, size_t ntabs, size_t count, uint64_t* out) {
  if (!c || !tabs || !out || ntabs == 0 || count == 0) return SP_EINVAL;
  if (32 * ntabs * count > HMAP_SIZE - HMAP_IN || 8 * ntabs > HMAP_GEN) return SP_EINVAL;
  HIPCHK(hipSetDevice(c->dev));
  std::vector<const Fq*> ptrs(ntabs);
  for (size_t k = 0; k < ntabs; k++) {
    size_t off = offs ? offs[k] : 0;
    if (!tabs[k] || off + count > tabs[k]->cap) return SP_EINVAL;
    ptrs[k] = tabs[k]->d + off;
  }
  const Fq* const* dp = (const Fq* const*)stage_small(c, 0, ptrs.data(), 8 * ntabs);
  size_t n = ntabs * count;
  hipLaunchKernelGGL(k_gather_elems, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, c->stream, dp, ntabs, count, (Fq*)hres(c));
  SPCHK(fetch_small(c, out, 32 * n));
  return hipGetLastError() == hipSuccess ? SP_OK : SP_EHIP;
}
int32_t sp_table_heads(sp_ctx* c, sp_table* const* tabs, size_t ntabs, uint64_t* out) { return gather_elems(c, tabs, nullptr, ntabs, 1, out); }
// ---- sharding helpers (sum-check tables by index residue, bound by row blocks) --------------------------------
__global__ void __launch_bounds__(256) k_residue_split(const Fq* __restrict__ src, size_t W, size_t g, size_t n, Fq* __restrict__ dst) { SP_FG_PRIO();
  for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (size_t)gridDim.x * blockDim.x) st_fq(dst + k, ld_fq(src + k * W + g));
}
__global__ void __launch_bounds__(256) k_add_into(Fq* __restrict__ dst, const Fq* __restrict__ src, size_t n) { SP_FG_PRIO();
  for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (size_t)gridDim.x * blockDim.x) st_fq(dst + k, fq_add(ld_fq(dst + k), ld_fq(src + k)));
}
int32_t sp_table_residue_split(sp_ctx* c, const sp_table* src, size_t W, size_t g, sp_table** out) {
  if (!c || !src || !out || W == 0 || g >= W || src->len % W != 0 || src->len / W == 0 || src->ctx->dev != c->dev) return SP_EINVAL;
  HIPCHK(hipSetDevice(c->dev));
  size_t n = src->len / W;
  SPCHK(table_new(c, n, false, out));
  ProfScope ps(c, PF_MISC, 64.0 * (double)n);
  hipLaunchKernelGGL(k_residue_split, dim3((unsigned)grid_for(n)), dim3(256), 0, c->stream, (const Fq*)src->d, W, g, n, (*out)->d);
  return hipGetLastError() == hipSuccess ? SP_OK : SP_EHIP;
}
int32_t sp_table_set_len(sp_table* t, size_t len) {
  if (!t || len == 0 || len > t->cap) return SP_EINVAL;
  t->len = len;
  return SP_OK;
}
int32_t sp_table_add_into(sp_ctx* c, sp_table* dst, const sp_table* src) {
  if (!c || !dst || !src || dst->len != src->len) return SP_EINVAL;
  HIPCHK(hipSetDevice(c->dev));
  ProfScope ps(c, PF_MISC, 96.0 * (double)dst->len);
  hipLaunchKernelGGL(k_add_into, dim3((unsigned)grid_for(dst->len)), dim3(256), 0, c->stream, dst->d, (const Fq*)src->d, dst->len);
  return hipGetLastError() == hipSuccess ? SP_OK : SP_EHIP;
}
int32_t sp_table_gather(sp_ctx* c, sp_table* const* tabs, const size_t* offs, size_t ntabs, size_t count, uint64_t* out) {
  return gather_elems(c, tabs, offs, ntabs, count, out);
}
// ---- hand-over of residue-sharded sum-check tables (the batched cubic sum-checks of SPARK): when the tables have become short
// enough that a round costs less than the exchange, every shard packs its sub-tables into one buffer (one DMA), the buffers are gathered
// (host: in-process for virtual shards, the commit transport between ranks), and the owner scatters them back into the full tables.
__global__ void __launch_bounds__(256) k_tables_pack(const Fq* const* __restrict__ ptrs, size_t ntabs, size_t count, Fq* __restrict__ out) { SP_FG_PRIO();
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < ntabs * count; t += (size_t)gridDim.x * blockDim.x)
    st_fq(out + t, ld_fq(ptrs[t / count] + t % count));
}
__global__ void __launch_bounds__(256) k_tables_unpack_residues(Fq* const* __restrict__ ptrs, size_t ntabs, size_t W, size_t sub, const Fq* __restrict__ in) { SP_FG_PRIO();
  // in[(g * ntabs + t) * sub + k] -> tabs[t][k * W + g]
  for (size_t x = (size_t)blockIdx.x * blockDim.x + threadIdx.x; x < W * ntabs * sub; x += (size_t)gridDim.x * blockDim.x) {
    size_t k = x % sub, t = (x / sub) % ntabs, g = x / (sub * ntabs);
    st_fq(ptrs[t] + k * W + g, ld_fq(in + x));
  }
}
int32_t sp_tables_pack(sp_ctx* c, sp_table* const* tabs, size_t ntabs, size_t count, uint64_t* out) {
  if (!c || !tabs || !out || ntabs == 0 || count == 0) return SP_EINVAL;
  HIPCHK(hipSetDevice(c->dev));
  std::vector<const Fq*> ptrs(ntabs);
  for (size_t k = 0; k < ntabs; k++) {
    if (!tabs[k] || count > tabs[k]->len) return SP_EINVAL;
    ptrs[k] = tabs[k]->d;
  }
  const size_t pb = (8 * ntabs + 255) & ~(size_t)255, db = 32 * ntabs * count;
  SPCHK(ensure_dstage(c, pb + db));
  SPCHK(stage_in(c, 0, ptrs.data(), 8 * ntabs));
  Fq* dout = (Fq*)((uint8_t*)c->dstage + pb);
  {
    ProfScope ps(c, PF_MISC, 64.0 * (double)(ntabs * count));
    hipLaunchKernelGGL(k_tables_pack, dim3((unsigned)grid_for(ntabs * count)), dim3(256), 0, c->stream, (const Fq* const*)c->dstage, ntabs, count, dout);
  }
  SPCHK(fetch_out(c, dout, out, db));
  return hipGetLastError() == hipSuccess ? SP_OK : SP_EHIP;
}
int32_t sp_tables_unpack_residues(sp_ctx* c, sp_table* const* tabs, size_t ntabs, size_t W, size_t sub, const uint64_t* in) {
  if (!c || !tabs || !in || ntabs == 0 || W == 0 || sub == 0) return SP_EINVAL;
  HIPCHK(hipSetDevice(c->dev));
  std::vector<Fq*> ptrs(ntabs);
  for (size_t k = 0; k < ntabs; k++) {
    if (!tabs[k] || W * sub > tabs[k]->cap) return SP_EINVAL;
    for (size_t j = 0; j < k; j++) if (tabs[j] == tabs[k]) return SP_EINVAL;  // every table once
    ptrs[k] = tabs[k]->d;
  }
  const size_t pb = (8 * ntabs + 255) & ~(size_t)255, db = 32 * W * ntabs * sub;
  SPCHK(ensure_dstage(c, pb + db));
  SPCHK(stage_in(c, 0, ptrs.data(), 8 * ntabs));
  SPCHK(stage_in(c, pb, in, db));
  {
    ProfScope ps(c, PF_MISC, 64.0 * (double)(W * ntabs * sub));
    hipLaunchKernelGGL(k_tables_unpack_residues, dim3((unsigned)grid_for(W * ntabs * sub)), dim3(256), 0, c->stream, (Fq* const*)c->dstage, ntabs, W, sub,
                       (const Fq*)((uint8_t*)c->dstage + pb));
  }
  for (size_t k = 0; k < ntabs; k++) tabs[k]->len = W * sub;
  SPCHK(sync_spin(c));  // the staging buffers are reused by the next call
  return hipGetLastError() == hipSuccess ? SP_OK : SP_EHIP;
}

}  // extern "C"
