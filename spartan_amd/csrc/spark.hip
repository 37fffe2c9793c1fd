// spartan_amd: device building blocks of the SPARK sparse-polynomial evaluation proof
// (src/sparse_mlpoly.rs, src/product_tree.rs): gathers, hash layers, product trees, batched cubic sum-check.
// All of it is F_q streaming work (no group operations), HBM/ALU-bound.
#include "internal.hpp"

__global__ void __launch_bounds__(256) k_from_index(const uint32_t* __restrict__ ix, size_t n, Fq* __restrict__ dst) { SP_FG_PRIO();
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) st_fq(dst + i, fq_from_u64(ix[i]));
}
__global__ void __launch_bounds__(256) k_gather(const Fq* __restrict__ mem, const uint32_t* __restrict__ addr, size_t n, Fq* __restrict__ dst) { SP_FG_PRIO();
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) st_fq(dst + i, ld_fq(mem + addr[i]));
}
__global__ void __launch_bounds__(256) k_hash_layer(const Fq* __restrict__ addr, const Fq* __restrict__ val, const Fq* __restrict__ ts, int ts_inc,
                                                    size_t n, Fq r_hash, Fq r_hash_sqr, Fq r_multiset, Fq* __restrict__ dst) { SP_FG_PRIO();
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    Fq a = addr ? ld_fq(addr + i) : fq_from_u64((uint64_t)i);
    Fq t = ts ? ld_fq(ts + i) : fq_zero();
    if (ts_inc) t = fq_add(t, fq_one());
    Fq h = fq_add(fq_add(fq_mul(t, r_hash_sqr), fq_mul(ld_fq(val + i), r_hash)), a);
    st_fq(dst + i, fq_sub(h, r_multiset));
  }
}
// The leaves AND the first multiplication layer of a product circuit in one pass (round 4): thread i hashes leaves i and i + n/2 and
// multiplies them (ProductCircuit::new, product_tree.rs:36-56: layer 1 [i] = leaf[i] * leaf[i + n/2], stored at offset n of the circuit's
// store) — the first layer no longer re-reads the 32 n bytes of leaves the hash layer has just written. PAIR: the read and the write set of
// one matrix differ by ts + 1 only (sparse_mlpoly.rs:572-598), i.e. by r_hash^2 per leaf: both circuits from one pass over addr, val, ts.
template <bool PAIR>
__global__ void __launch_bounds__(256) k_hash_layer_first(const Fq* __restrict__ addr, const Fq* __restrict__ val, const Fq* __restrict__ ts, int ts_inc,
                                                          size_t n, Fq r_hash, Fq r_hash_sqr, Fq r_multiset, Fq* __restrict__ dst_a, Fq* __restrict__ dst_b) { SP_FG_PRIO();
  const size_t half = n / 2;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < half; i += (size_t)gridDim.x * blockDim.x) {
    Fq h[2];
#pragma unroll
    for (int k = 0; k < 2; k++) {
      const size_t j = i + (size_t)k * half;
      Fq a = addr ? ld_fq(addr + j) : fq_from_u64((uint64_t)j);
      Fq t = ts ? ld_fq(ts + j) : fq_zero();
      if (ts_inc) t = fq_add(t, fq_one());
      h[k] = fq_sub(fq_add(fq_add(fq_mul(t, r_hash_sqr), fq_mul(ld_fq(val + j), r_hash)), a), r_multiset);
      st_fq(dst_a + j, h[k]);
    }
    st_fq(dst_a + n + i, fq_mul(h[0], h[1]));
    if (PAIR) {
      Fq w0 = fq_add(h[0], r_hash_sqr), w1 = fq_add(h[1], r_hash_sqr);  // (ts + 1) r^2 + val r + addr - gamma
      st_fq(dst_b + i, w0);
      st_fq(dst_b + half + i, w1);
      st_fq(dst_b + n + i, fq_mul(w0, w1));
    }
  }
}
__global__ void __launch_bounds__(256) k_prod_layer(const Fq* __restrict__ in, size_t half, Fq* __restrict__ out) { SP_FG_PRIO();
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < half; i += (size_t)gridDim.x * blockDim.x)
    st_fq(out + i, fq_mul(ld_fq(in + i), ld_fq(in + half + i)));
}

// the same layer of several product circuits of equal size in one launch (grid.y = circuit): a 2^20-leaf tree has 19
// layers of which 12 are shorter than a launch (a few us of fixed cost each)
struct Stores16 {
  Fq* p[16];
};
__global__ void __launch_bounds__(256) k_prod_layer_many(Stores16 st, size_t off, size_t half, size_t noff) { SP_FG_PRIO();
  const Fq* in = st.p[blockIdx.y] + off;
  Fq* out = st.p[blockIdx.y] + noff;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < half; i += (size_t)gridDim.x * blockDim.x)
    st_fq(out + i, fq_mul(ld_fq(in + i), ld_fq(in + half + i)));
}

// TWO layers per launch for the middle of the tree (round 4): the layers between the streaming ones and the one-launch tail are a few
// microseconds of work behind a launch each. Thread i reads the four entries i, i + q, i + 2q, i + 3q (q = len / 4) of layer k, writes
// layer k+1 [i] = x0 x2 and [i + q] = x1 x3 (its pairs are (i, i + len/2)), and layer k+2 [i] = their product.
__global__ void __launch_bounds__(256) k_prod_layer2_many(Stores16 st, size_t off, size_t len) { SP_FG_PRIO();
  const size_t q = len / 4, off1 = off + len, off2 = off1 + len / 2;
  Fq* base = st.p[blockIdx.y];
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < q; i += (size_t)gridDim.x * blockDim.x) {
    const Fq x0 = ld_fq(base + off + i), x1 = ld_fq(base + off + q + i), x2 = ld_fq(base + off + 2 * q + i), x3 = ld_fq(base + off + 3 * q + i);
    const Fq lo = fq_mul(x0, x2), hi = fq_mul(x1, x3);
    st_fq(base + off1 + i, lo);
    st_fq(base + off1 + q + i, hi);
    st_fq(base + off2 + i, fq_mul(lo, hi));
  }
}
// The short layers of the same circuits in ONE launch: from a layer of `len` <= 2048 elements down to the two roots, one
// workgroup per circuit, each layer kept in LDS for the next (a 2^20-leaf tree has 10 such layers: ten launches of a few
// microseconds each on the path to the first product_circuits_evaluate).
__global__ void __launch_bounds__(256) k_prod_layer_tail(Stores16 st, size_t off, size_t len) { SP_FG_PRIO();
  __shared__ Fq cur[1024];
  Fq* base = st.p[blockIdx.x];
  size_t half = len / 2, noff = off + len;
  for (size_t i = threadIdx.x; i < half; i += 256) {
    Fq v = fq_mul(ld_fq(base + off + i), ld_fq(base + off + half + i));
    cur[i] = v;
    st_fq(base + noff + i, v);
  }
  __syncthreads();
  len = half;
  while (len > 2) {
    half = len / 2;
    noff += len;
    Fq v[2];
    int n = 0;
    for (size_t i = threadIdx.x; i < half; i += 256) v[n++] = fq_mul(cur[i], cur[half + i]);   // half <= 512: at most two per thread
    __syncthreads();
    n = 0;
    for (size_t i = threadIdx.x; i < half; i += 256) { cur[i] = v[n]; st_fq(base + noff + i, v[n]); n++; }
    __syncthreads();
    len = half;
  }
}

struct Triple {
  Fq *a, *b, *c;
  Fq* c_out;  // where this instance writes the bound C (non-null for exactly one instance per distinct C table)
};
struct TripleInline { Triple t[24]; };  // up to 24 instances in the kernel arguments (T == null) instead of the host-mapped page: see Bind2Inline
__device__ __forceinline__ void cubic_point(const Fq& a0, const Fq& a1, const Fq& b0, const Fq& b1, const Fq& c0, const Fq& c1, Fq (&e)[3]) {
  // the line through (0, x0), (1, x1) at 2 and 3: x1 + d, x1 + 2d with d = x1 - x0 (three modular additions per table)
  Fq da = fq_sub(a1, a0), db = fq_sub(b1, b0), dc = fq_sub(c1, c0);
  Fq a2 = fq_add(a1, da), b2 = fq_add(b1, db), c2 = fq_add(c1, dc);
  Fq a3 = fq_add(a2, da), b3 = fq_add(b2, db), c3 = fq_add(c2, dc);
  e[0] = fq_add(e[0], fq_mul(fq_mul(a0, b0), c0));
  e[1] = fq_add(e[1], fq_mul(fq_mul(a2, b2), c2));
  e[2] = fq_add(e[2], fq_mul(fq_mul(a3, b3), c3));
}
// grid (nblk, ninst): partials[(inst*nblk + blk)*3 + {0,1,2}]
__global__ void __launch_bounds__(256) k_cubic_eval_batched(const Triple* __restrict__ T, TripleInline IN, size_t half, Fq* __restrict__ partials) { SP_FG_PRIO();
  __shared__ Fq sm[256];
  Triple t = T ? T[blockIdx.y] : IN.t[blockIdx.y];
  Fq e[3] = {fq_zero(), fq_zero(), fq_zero()};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < half; i += (size_t)gridDim.x * blockDim.x)
    cubic_point(ld_fq(t.a + i), ld_fq(t.a + half + i), ld_fq(t.b + i), ld_fq(t.b + half + i), ld_fq(t.c + i), ld_fq(t.c + half + i), e);
  block_sum_fq<3>(e, sm);
  if (threadIdx.x == 0) {
    Fq* p = partials + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 3;
    st_fq(p, e[0]); st_fq(p + 1, e[1]); st_fq(p + 2, e[2]);
  }
}
__global__ void __launch_bounds__(256) k_cubic_bind_eval_batched(const Triple* __restrict__ T, TripleInline IN, size_t quarter, Fq r, Fq* __restrict__ partials) { SP_FG_PRIO();
  __shared__ Fq sm[256];
  Triple t = T ? T[blockIdx.y] : IN.t[blockIdx.y];
  Fq e[3] = {fq_zero(), fq_zero(), fq_zero()};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < quarter; i += (size_t)gridDim.x * blockDim.x) {
    Fq lo[3], hi[3];
    Fq* ptr[3] = {t.a, t.b, t.c};
#pragma unroll
    for (int k = 0; k < 3; k++) {
      Fq x0 = ld_fq(ptr[k] + i), x1 = ld_fq(ptr[k] + quarter + i), x2 = ld_fq(ptr[k] + 2 * quarter + i), x3 = ld_fq(ptr[k] + 3 * quarter + i);
      lo[k] = fq_add(x0, fq_mul(r, fq_sub(x2, x0)));
      hi[k] = fq_add(x1, fq_mul(r, fq_sub(x3, x1)));
      if (k < 2) {
        st_fq(ptr[k] + i, lo[k]);
        st_fq(ptr[k] + quarter + i, hi[k]);
      } else if (t.c_out) {  // C may be shared between instances: bound out of place, once
        st_fq(t.c_out + i, lo[k]);
        st_fq(t.c_out + quarter + i, hi[k]);
      }
    }
    cubic_point(lo[0], hi[0], lo[1], hi[1], lo[2], hi[2], e);
  }
  block_sum_fq<3>(e, sm);
  if (threadIdx.x == 0) {
    Fq* p = partials + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 3;
    st_fq(p, e[0]); st_fq(p + 1, e[1]); st_fq(p + 2, e[2]);
  }
}
// ---- the eq table as a FACTOR (round 4): the throughput-sized rounds of the product-circuit instances -------------------------------
// In prove_cubic_batched under ProductCircuitEvalProofBatched::prove (product_tree.rs:259-383) the third table of every product-circuit
// instance is poly_C_par = EqPolynomial::new(rand).evals() (:279): C[x] = prod_k eq(x_k, rho_k). After the rounds' binds at r_1..r_{j-1} the
// bound table is  C_j[t, x''] = [prod_{k<j} eq(r_k, rho_k)] * eq(t, rho_j) * eq(x''; rho_{j+1}..),  and the ORIGINAL table's leading
// entries hold the last factor already:  C[x''] = [prod_{k<=j} (1 - rho_k)] * eq(x''; rho_{j+1}..)  (top j index bits zero). So
//     sum_x'' A(t,x'') B(t,x'') C_j[t,x'']  =  kappa_j(t) * q(t),   q(t) = sum_x'' A(t,x'') B(t,x'') C[x''],
// with a scalar kappa_j(t) the host knows (spark.inc): the device never binds the eq table, reads ONE entry of it per index instead of
// four, and q is only QUADRATIC in t: q(0) and q(2) come from here, q(1) from the round's claim, q(3) by extrapolation — two evaluation
// points instead of three. Per index and instance: 8 multiplications instead of 12 (4 instead of 6 in the first evaluation), 9 loads and
// 4 stores instead of 12 and 6. Exact field identities: the round polynomials, hence the proof bytes, are those of the generic kernels
// (tests/test_gpu_large.py). The generic instances (the dot-product circuits of the widest layer, whose third table is a real
// table) take the generic arithmetic and return FOUR evaluations, t = 0, 1, 2, 3 (their e(1) is needed to split the claim).
// partials[(inst * nblk + blk) * 4 + {0,1,2,3}]: factored {q(0), q(2), 0, 0}, generic {e(0), e(1), e(2), e(3)}.
__device__ __forceinline__ void quad_point_eq(const Fq& a0, const Fq& a1, const Fq& b0, const Fq& b1, const Fq& c, Fq (&e)[4]) {
  Fq a2 = fq_add(a1, fq_sub(a1, a0)), b2 = fq_add(b1, fq_sub(b1, b0));
  e[0] = fq_add(e[0], fq_mul(fq_mul(a0, b0), c));
  e[1] = fq_add(e[1], fq_mul(fq_mul(a2, b2), c));
}
__device__ __forceinline__ void cubic_point4(const Fq& a0, const Fq& a1, const Fq& b0, const Fq& b1, const Fq& c0, const Fq& c1, Fq (&e)[4]) {
  Fq da = fq_sub(a1, a0), db = fq_sub(b1, b0), dc = fq_sub(c1, c0);
  Fq a2 = fq_add(a1, da), b2 = fq_add(b1, db), c2 = fq_add(c1, dc);
  Fq a3 = fq_add(a2, da), b3 = fq_add(b2, db), c3 = fq_add(c2, dc);
  e[0] = fq_add(e[0], fq_mul(fq_mul(a0, b0), c0));
  e[1] = fq_add(e[1], fq_mul(fq_mul(a1, b1), c1));
  e[2] = fq_add(e[2], fq_mul(fq_mul(a2, b2), c2));
  e[3] = fq_add(e[3], fq_mul(fq_mul(a3, b3), c3));
}
// GEN selects the arithmetic at compile time (one launch for the product-circuit instances, one for the generic ones when there are any):
// a kernel holding both bodies is allocated for the larger one (205 registers, two waves per SIMD instead of three)
template <bool GEN>
__global__ void __launch_bounds__(256) k_cubic_eval_batched_eq(TripleInline IN, unsigned inst0, size_t half, Fq* __restrict__ partials) { SP_FG_PRIO();
  __shared__ Fq sm[256];
  const unsigned inst = inst0 + blockIdx.y;
  Triple t = IN.t[inst];
  Fq e[4] = {fq_zero(), fq_zero(), fq_zero(), fq_zero()};
  if (!GEN) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < half; i += (size_t)gridDim.x * blockDim.x)
      quad_point_eq(ld_fq(t.a + i), ld_fq(t.a + half + i), ld_fq(t.b + i), ld_fq(t.b + half + i), ld_fq(t.c + i), e);
  } else {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < half; i += (size_t)gridDim.x * blockDim.x)
      cubic_point4(ld_fq(t.a + i), ld_fq(t.a + half + i), ld_fq(t.b + i), ld_fq(t.b + half + i), ld_fq(t.c + i), ld_fq(t.c + half + i), e);
  }
  if (GEN) block_sum_fq<4>(e, sm);
  else {
    Fq q[2] = {e[0], e[1]};
    block_sum_fq<2>(q, sm);
    e[0] = q[0]; e[1] = q[1];
  }
  if (threadIdx.x == 0) {
    Fq* p = partials + ((size_t)inst * gridDim.x + blockIdx.x) * 4;
    st_fq(p, e[0]); st_fq(p + 1, e[1]); st_fq(p + 2, e[2]); st_fq(p + 3, e[3]);
  }
}
template <bool GEN>
__global__ void __launch_bounds__(256) k_cubic_bind_eval_batched_eq(TripleInline IN, unsigned inst0, size_t quarter, Fq r, Fq* __restrict__ partials) { SP_FG_PRIO();
  __shared__ Fq sm[256];
  const unsigned inst = inst0 + blockIdx.y;
  Triple t = IN.t[inst];
  Fq e[4] = {fq_zero(), fq_zero(), fq_zero(), fq_zero()};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < quarter; i += (size_t)gridDim.x * blockDim.x) {
    Fq lo[3], hi[3];
    Fq* ptr[3] = {t.a, t.b, t.c};
#pragma unroll
    for (int k = 0; k < 2; k++) {
      Fq x0 = ld_fq(ptr[k] + i), x1 = ld_fq(ptr[k] + quarter + i), x2 = ld_fq(ptr[k] + 2 * quarter + i), x3 = ld_fq(ptr[k] + 3 * quarter + i);
      lo[k] = fq_add(x0, fq_mul(r, fq_sub(x2, x0)));
      hi[k] = fq_add(x1, fq_mul(r, fq_sub(x3, x1)));
      st_fq(ptr[k] + i, lo[k]);
      st_fq(ptr[k] + quarter + i, hi[k]);
    }
    if (!GEN) {
      quad_point_eq(lo[0], hi[0], lo[1], hi[1], ld_fq(t.c + i), e);  // the eq table is read, never bound
    } else {
      Fq x0 = ld_fq(ptr[2] + i), x1 = ld_fq(ptr[2] + quarter + i), x2 = ld_fq(ptr[2] + 2 * quarter + i), x3 = ld_fq(ptr[2] + 3 * quarter + i);
      lo[2] = fq_add(x0, fq_mul(r, fq_sub(x2, x0)));
      hi[2] = fq_add(x1, fq_mul(r, fq_sub(x3, x1)));
      if (t.c_out) {
        st_fq(t.c_out + i, lo[2]);
        st_fq(t.c_out + quarter + i, hi[2]);
      }
      cubic_point4(lo[0], hi[0], lo[1], hi[1], lo[2], hi[2], e);
    }
  }
  if (GEN) block_sum_fq<4>(e, sm);
  else {
    Fq q[2] = {e[0], e[1]};
    block_sum_fq<2>(q, sm);
    e[0] = q[0]; e[1] = q[1];
  }
  if (threadIdx.x == 0) {
    Fq* p = partials + ((size_t)inst * gridDim.x + blockIdx.x) * 4;
    st_fq(p, e[0]); st_fq(p + 1, e[1]); st_fq(p + 2, e[2]); st_fq(p + 3, e[3]);
  }
}
// t[i] *= k for i < n (the hand-over from the factored rounds: the bound eq table the generic kernels continue with)
__global__ void __launch_bounds__(256) k_scale_prefix(Fq* __restrict__ t, size_t n, Fq k) { SP_FG_PRIO();
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) st_fq(t + i, fq_mul(ld_fq(t + i), k));
}
// Latency form of the fused round for short tables (quarter <= a few hundred): the ~12 dependent field
// multiplications of one index are spread over 8 lanes (six do one bind each, then three do one evaluation point
// each), so a round costs ~3 multiplications of latency instead of 12. Block = 32 indices x 8 roles; grid (nblk, ninst).
__global__ void __launch_bounds__(256) k_cubic_bind_eval_tiny(const Triple* __restrict__ T, TripleInline IN, size_t quarter, Fq r, Fq* __restrict__ partials, DoneSig sig) { SP_FG_PRIO();
  __shared__ Fq bound[32][6];  // [index][table*2 + half]
  __shared__ Fq lv[32][3][3];  // [index][table][t = 0, 2, 3]: the bound pair's line
  __shared__ Fq red[3][32];
  Triple t = T ? T[blockIdx.y] : IN.t[blockIdx.y];
  int li = threadIdx.x >> 3, role = threadIdx.x & 7;
  size_t i = (size_t)blockIdx.x * 32 + li;
  bool live = i < quarter;
  if (role < 6 && live) {
    int k = role >> 1, half = role & 1;
    Fq* ptr = k == 0 ? t.a : (k == 1 ? t.b : t.c);
    Fq x0 = ld_fq(ptr + (size_t)half * quarter + i), x2 = ld_fq(ptr + (size_t)(2 + half) * quarter + i);
    Fq v = fq_add(x0, fq_mul(r, fq_sub(x2, x0)));
    bound[li][role] = v;
    if (k < 2) st_fq(ptr + (size_t)half * quarter + i, v);
    else if (t.c_out) st_fq(t.c_out + (size_t)half * quarter + i, v);  // C may be shared between instances: bound out of place, once
  }
  __syncthreads();
  // lane `role` < 3 first extends the line of table `role` to t = 2, 3 (three modular additions), then — as evaluation point
  // `role` — multiplies the three tables' values: 3 additions + 2 multiplications deep instead of 12 + 2 per lane
  if (role < 3 && live) {
    const Fq lo = bound[li][2 * role], hi = bound[li][2 * role + 1];
    const Fq d = fq_sub(hi, lo), e2 = fq_add(hi, d), e3 = fq_add(e2, d);
    lv[li][role][0] = lo; lv[li][role][1] = e2; lv[li][role][2] = e3;
  }
  __syncthreads();
  if (role < 3) {
    Fq e = fq_zero();
    if (live) e = fq_mul(fq_mul(lv[li][0][role], lv[li][1][role]), lv[li][2][role]);
    red[role][li] = e;
  }
  __syncthreads();
  for (int s = 16; s > 0; s >>= 1) {
    if (role < 3 && li < s) red[role][li] = fq_add(red[role][li], red[role][li + s]);
    __syncthreads();
  }
  if (threadIdx.x < 3) st_fq(partials + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 3 + threadIdx.x, red[threadIdx.x][0]);
  signal_done(sig);
}
// Latency form of k_cubic_eval_batched (the first round of a layer's sum-check, half <= 8192): four lanes per index, lanes
// 0..2 evaluate t = 0, 2, 3 with one instruction stream; block = 64 indices; grid (nblk, ninst).
__global__ void __launch_bounds__(256) k_cubic_eval_tiny(const Triple* __restrict__ T, TripleInline IN, size_t half, Fq* __restrict__ partials, DoneSig sig) { SP_FG_PRIO();
  __shared__ Fq red[3][64];
  Triple t = T ? T[blockIdx.y] : IN.t[blockIdx.y];
  int li = threadIdx.x >> 2, role = threadIdx.x & 3;
  size_t i = (size_t)blockIdx.x * 64 + li;
  Fq e = fq_zero();
  if (role < 3 && i < half) {
    Fq pt[3];
    const Fq* ptr[3] = {t.a, t.b, t.c};
#pragma unroll
    for (int k = 0; k < 3; k++) {
      Fq x0 = ld_fq(ptr[k] + i), x1 = ld_fq(ptr[k] + half + i);
      Fq x2 = fq_sub(fq_dbl(x1), x0), x3 = fq_sub(fq_add(x2, x1), x0);
#pragma unroll
      for (int w = 0; w < 4; w++) pt[k].l[w] = role == 0 ? x0.l[w] : (role == 1 ? x2.l[w] : x3.l[w]);
    }
    e = fq_mul(fq_mul(pt[0], pt[1]), pt[2]);
  }
  if (role < 3) red[role][li] = e;
  __syncthreads();
  for (int s = 32; s > 0; s >>= 1) {
    if (role < 3 && li < s) red[role][li] = fq_add(red[role][li], red[role][li + s]);
    __syncthreads();
  }
  if (threadIdx.x < 3) st_fq(partials + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 3 + threadIdx.x, red[threadIdx.x][0]);
  signal_done(sig);
}

// ---- two rounds per launch (the latency-bound tail of prove_cubic_batched, sumcheck.rs:287-393) -----------------------
// A round trip (launch, completion flag, host wake-up) costs ~20 us, the arithmetic of a short round ~4 us. Two facts let
// one trip advance TWO rounds. (1) With both challenges in hand, the doubly bound table is a four-term combination of the
// current one, T''[z] = (1-r1)((1-r0) T[z] + r0 T[z+L/2]) + r1((1-r0) T[z+L/4] + r0 T[z+3L/4]), computed per entry with no
// dependency between workgroups. (2) The evaluations of the round AFTER a bind are a cubic in that bind's challenge:
// with x0..x3 the entries (i, i+q, i+2q, i+3q) of a table of length 4q, P(t) the line through (x0, x1) and U(t) the line
// through (x2, x3), the entry pair of the bound table is ((1-r) x0 + r x2, (1-r) x1 + r x3) and its line at t is
// (1-r) P(t) + r U(t); so sum_i A(t)B(t)C(t) = (1-r)^3 M0 + (1-r)^2 r M1 + (1-r) r^2 M2 + r^3 M3 with M0 = sum P_A P_B P_C,
// M3 = sum U_A U_B U_C and M1, M2 from T1 = sum (P+U)(P+U)(P+U) = M0+M1+M2+M3, T2 = sum (P-U)(P-U)(P-U) = M0-M1+M2-M3. The kernel
// returns (M0, M3, T1, T2) for t = 0, 2, 3 next to the evaluations of its own round: the caller derives the next challenge,
// evaluates the cubic for the round after it, derives that challenge too, and only then comes back. Exact field
// arithmetic: the values are those of the one-round-per-launch path.
// Block = 8 groups x 32 lanes; group g holds the (up to) four entries z = g + p*ng, p = 0..3, of each bound table.
// partials[(inst*nblk + blk)*18 + {0..2: evaluations at t = 0, 2, 3 | 3 + 4 t' + {0,1,2,3}: M0, M3, T1, T2 | 15..17: the entries
// of A, B, C when the bound tables have length 1}]
struct Triple2 {
  Fq *a, *b, *c;
  Fq* c_out;
};
__device__ __forceinline__ Fq line_at(const Fq& u, const Fq& v, int t) {  // the line through (0, u), (1, v) at t = 0, 2, 3
  Fq x2 = fq_sub(fq_dbl(v), u), x3 = fq_sub(fq_add(x2, v), u), r;
#pragma unroll
  for (int w = 0; w < 4; w++) r.l[w] = t == 0 ? u.l[w] : (t == 1 ? x2.l[w] : x3.l[w]);
  return r;
}
// dump != nullptr (only when the outputs describe tables of at most 8 entries): the tables themselves go out as well,
// dump[(instance * 3 + table) * 8 + z] — the caller finishes the last <= 3 rounds of the sum-check on its own core.
// Up to 24 instances travel in the kernel arguments (T == null): the table pointers and the weights are then scalar loads from the
// kernarg segment instead of a read of the host-mapped page over PCIe at the head of every workgroup.
struct Bind2Inline {
  Triple2 t[24];
  Fq w[24];
};
// ah.bell != nullptr: the launch was enqueued ahead of its challenges (AheadArm, internal.hpp): it waits for the bell, takes r0 and r1 from there, or
// gives up without touching anything.
__global__ void __launch_bounds__(256) k_cubic_bind2_eval(const Triple2* __restrict__ T, Bind2Inline IN, const Fq* __restrict__ weights, size_t len, int nbind, Fq r0, Fq r1,
                                                          Fq* __restrict__ partials, Fq* __restrict__ dump, DoneSig sig, AheadArgs ah) { SP_FG_PRIO();
  if (ah.bell) {  // the wait keeps the latency kernels' issue priority: at the default priority the wavefront that watches the bell queues behind the
                  // co-resident MSM's older wavefronts and the gain of the early launch is gone (measured: profiles/r6_ab_launch_ahead.txt)
    __shared__ uint32_t go;
    if (threadIdx.x < 32) {
      const uint32_t d = ahead_wait(ah);
      if (threadIdx.x == 0) go = d;
    }
    __syncthreads();
    if (go != 1) return;
    r0 = ahead_challenge(ah.chal);
    r1 = ahead_challenge(ah.chal + 8);
  }
  __shared__ Fq first[8][12][2];  // [group][table*4 + position][half]: the entries bound at r0
  __shared__ Fq bnd[8][12];       // [group][table*4 + slot]: the entries bound at r0 and r1 (slots 0..3 = x0..x3)
  __shared__ Fq red[18][8];
  __shared__ Fq lv[8][3][6][3];   // [group][table][line V, W, P, U, P + U, P - U][t = 0, 2, 3]
  DoneSig kt = sig; if (blockIdx.x != 0 || blockIdx.y != 0) kt.kt = nullptr;
  SP_KT(kt, 0);
  const Triple2 t = T ? T[blockIdx.y] : IN.t[blockIdx.y];
  const Fq wv = !weights ? fq_zero() : (T ? ld_fq(weights + blockIdx.y) : IN.w[blockIdx.y]);  // from the host-mapped page (T != null): requested now, needed last
  const int grp = threadIdx.x >> 5, lane = threadIdx.x & 31;
  const size_t n2 = len >> nbind;                  // length of the tables the outputs describe (nbind = 0, 1 or 2 binds first)
  const size_t np = n2 < 4 ? n2 : 4, ng = n2 / np; // entries per group, groups
  const size_t g = (size_t)blockIdx.x * 8 + grp;
  const bool live = g < ng;
  Fq* const ptr[3] = {t.a, t.b, t.c};
  if (nbind) {
    if (nbind == 2) {
      if (live && lane < 24) {
        const int k = lane >> 3, p = (lane >> 1) & 3, h = lane & 1;
        if ((size_t)p < np) {
          const size_t y = g + (size_t)p * ng + (size_t)h * (len / 4);
          Fq lo = ld_fq(ptr[k] + y), hi = ld_fq(ptr[k] + y + len / 2);
          SP_KT(kt, 1);
          first[grp][k * 4 + p][h] = fq_add(lo, fq_mul(r0, fq_sub(hi, lo)));
        }
      }
      SP_KT(kt, 2);
      __syncthreads();
    }
    if (live && lane < 12) {
      const int k = lane >> 2, p = lane & 3;
      if ((size_t)p < np) {
        Fq lo, hi;
        if (nbind == 2) { lo = first[grp][lane][0]; hi = first[grp][lane][1]; }
        else { lo = ld_fq(ptr[k] + g + (size_t)p * ng); hi = ld_fq(ptr[k] + g + (size_t)p * ng + len / 2); }
        Fq v = fq_add(lo, fq_mul(nbind == 2 ? r1 : r0, fq_sub(hi, lo)));
        const size_t z = g + (size_t)p * ng;
        if (k < 2) st_fq(ptr[k] + z, v);
        else if (t.c_out) st_fq(t.c_out + z, v);  // C may be shared between instances: bound out of place, once
        if (dump) st_fq(dump + ((size_t)blockIdx.y * 3 + k) * 8 + z, v);
        bnd[grp][k * 4 + (np == 2 && p == 1 ? 2 : p)] = v;  // a two-entry table is the pair (x0, x2)
      }
    }
  } else if (live && lane < 12) {
    const int k = lane >> 2, p = lane & 3;
    if ((size_t)p < np) {
      const Fq v = ld_fq(ptr[k] + g + (size_t)p * ng);
      if (dump) st_fq(dump + ((size_t)blockIdx.y * 3 + k) * 8 + g + (size_t)p * ng, v);
      bnd[grp][k * 4 + (np == 2 && p == 1 ? 2 : p)] = v;
    }
  }
  SP_KT(kt, 3);
  __syncthreads();
  // The 18 triple products per group (lane = 6 t' + kind; kind 0/1 = the two entry pairs of this round, 2..5 = M0, M3, T1, T2)
  // take the values of four lines per table — V through (x0, x2), W through (x1, x3), P through (x0, x1), U through (x2, x3) —
  // at t = 0, 2, 3, and of P + U, P - U. Each is a handful of modular additions; computed inside the product lanes they ran
  // as three divergent paths, ~55 dependent additions deep (9 us of a 17 us workgroup: bench/ktime_probe.py). Here twelve
  // lanes extend one line each (3 additions), nine lanes form sum and difference (2), and the product lanes only multiply.
  if (live && lane < 12 && np >= 2) {
    const int k = lane >> 2, line = lane & 3;
    if (line == 0 || np == 4) {
      const int iu = line == 0 ? 0 : (line == 1 ? 1 : (line == 2 ? 0 : 2)), iv = line == 0 ? 2 : (line == 1 ? 3 : (line == 2 ? 1 : 3));
      const Fq u = bnd[grp][k * 4 + iu], v = bnd[grp][k * 4 + iv];
      const Fq d = fq_sub(v, u), e2 = fq_add(v, d), e3 = fq_add(e2, d);  // the line through (0, u), (1, v) at 2 and 3
      lv[grp][k][line][0] = u; lv[grp][k][line][1] = e2; lv[grp][k][line][2] = e3;
    }
  }
  __syncthreads();
  if (live && lane < 9 && np == 4) {
    const int k = lane / 3, tt = lane % 3;
    const Fq P = lv[grp][k][2][tt], U = lv[grp][k][3][tt];
    lv[grp][k][4][tt] = fq_add(P, U);
    lv[grp][k][5][tt] = fq_sub(P, U);
  }
  __syncthreads();
  Fq e = fq_zero();
  if (live && lane < 18 && np >= 2) {
    const int tt = lane / 6, kind = lane % 6;
    if (kind == 0 || np == 4) {
      e = fq_mul(fq_mul(lv[grp][0][kind][tt], lv[grp][1][kind][tt]), lv[grp][2][kind][tt]);
      SP_KT(kt, 4);
      if (weights) e = fq_mul(e, wv);  // coeffs[i] of sumcheck.rs:359-369: the caller only adds the instances up
    }
  }
  SP_KT(kt, 5);
  if (lane < 18) red[lane][grp] = e;
  __syncthreads();
  if (threadIdx.x < 18) {
    Fq acc = red[threadIdx.x][0];
#pragma unroll
    for (int k = 1; k < 8; k++) acc = fq_add(acc, red[threadIdx.x][k]);
    red[threadIdx.x][0] = acc;
  }
  __syncthreads();
  if (threadIdx.x < 18) {
    Fq* o = partials + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 18;
    const int x = threadIdx.x;
    Fq v;
    if (x < 3) v = fq_add(red[6 * x][0], red[6 * x + 1][0]);               // evaluations at t = 0, 2, 3
    else if (x < 15) v = red[6 * ((x - 3) / 4) + 2 + (x - 3) % 4][0];       // M0, M3, T1, T2 per t
    else v = (n2 == 1 && blockIdx.x == 0) ? bnd[0][(x - 15) * 4] : fq_zero();  // final claims
    st_fq(o + x, v);
  }
  SP_KT(kt, 6);
  signal_done(sig);
  SP_KT(kt, 7);
}
// partials[ninst][nblk][K] -> out[ninst][K]; one block per instance
__global__ void __launch_bounds__(256) k_reduce_partials_batched(const Fq* __restrict__ partials, size_t nblk, int K, Fq* __restrict__ out, DoneSig sig) { SP_FG_PRIO();
  __shared__ Fq sm[256];
  const Fq* p = partials + (size_t)blockIdx.x * nblk * K;
  for (int k = 0; k < K; k++) {
    Fq acc[1] = {fq_zero()};
    for (size_t b = threadIdx.x; b < nblk; b += 256) acc[0] = fq_add(acc[0], ld_fq(p + b * K + k));
    block_sum_fq<1>(acc, sm);
    if (threadIdx.x == 0) st_fq(out + (size_t)blockIdx.x * K + k, acc[0]);
  }
  signal_done(sig);
}
// grid (nblk, nt): partials[t*nblk + blk] = partial <chi, T_t>
__global__ void __launch_bounds__(256) k_dot_many(const Fq* __restrict__ chi, Fq* const* __restrict__ tabs, size_t n, Fq* __restrict__ partials) { SP_FG_PRIO();
  __shared__ Fq sm[256];
  const Fq* t = tabs[blockIdx.y];
  Fq acc[1] = {fq_zero()};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    acc[0] = fq_add(acc[0], fq_mul(ld_fq(chi + i), ld_fq(t + i)));
  block_sum_fq<1>(acc, sm);
  if (threadIdx.x == 0) st_fq(partials + (size_t)blockIdx.y * gridDim.x + blockIdx.x, acc[0]);
}
__global__ void __launch_bounds__(256) k_dot3(const Fq* __restrict__ l, const Fq* __restrict__ r, const Fq* __restrict__ w, size_t n,
                                              Fq* __restrict__ partials) { SP_FG_PRIO();
  __shared__ Fq sm[256];
  Fq acc[1] = {fq_zero()};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    acc[0] = fq_add(acc[0], fq_mul(fq_mul(ld_fq(l + i), ld_fq(r + i)), ld_fq(w + i)));
  block_sum_fq<1>(acc, sm);
  if (threadIdx.x == 0) st_fq(partials + blockIdx.x, acc[0]);
}

// grid (nblk, nt): partials[t*nblk + blk] = partial sum_i l_t[i] r_t[i] w_t[i]; ptrs = [l_0..l_{nt-1} | r_0.. | w_0..]
__global__ void __launch_bounds__(256) k_dot3_many(const Fq* const* __restrict__ ptrs, size_t nt, size_t n, Fq* __restrict__ partials) { SP_FG_PRIO();
  __shared__ Fq sm[256];
  const Fq *l = ptrs[blockIdx.y], *r = ptrs[nt + blockIdx.y], *w = ptrs[2 * nt + blockIdx.y];
  Fq acc[1] = {fq_zero()};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    acc[0] = fq_add(acc[0], fq_mul(fq_mul(ld_fq(l + i), ld_fq(r + i)), ld_fq(w + i)));
  block_sum_fq<1>(acc, sm);
  if (threadIdx.x == 0) st_fq(partials + (size_t)blockIdx.y * gridDim.x + blockIdx.x, acc[0]);
}

static Fq limbs(const uint64_t* p) {
  Fq x;
  memcpy(x.l, p, 32);
  return x;
}

extern "C" {

int32_t sp_index_upload(sp_ctx* c, const uint64_t* idx, size_t n, sp_index** out) {
  if (!c || !idx || !out || n == 0) return SP_EINVAL;
  std::vector<uint32_t> v(n);
  for (size_t i = 0; i < n; i++) {
    if (idx[i] > 0xffffffffULL) return SP_EINVAL;
    v[i] = (uint32_t)idx[i];
  }
  HIPCHK(hipSetDevice(c->dev));
  sp_index* ix = new (std::nothrow) sp_index();
  if (!ix) return SP_ENOMEM;
  ix->ctx = c; ix->n = n; ix->d = nullptr;
  hipError_t e = hipMalloc((void**)&ix->d, 4 * n);
  if (e == hipSuccess) e = hipMemcpy(ix->d, v.data(), 4 * n, hipMemcpyHostToDevice);
  if (e != hipSuccess) { if (ix->d) (void)hipFree(ix->d); delete ix; return e == hipErrorOutOfMemory ? SP_ENOMEM : SP_EHIP; }
  *out = ix;
  return SP_OK;
}
void sp_index_free(sp_index* ix) {
  if (!ix) return;
  (void)hipSetDevice(ix->ctx->dev);
  (void)hipStreamSynchronize(ix->ctx->stream);
  (void)hipFree(ix->d);
  delete ix;
}
int32_t sp_table_from_index(sp_ctx* c, const sp_index* ix, sp_table* dst, size_t dst_off) {
  if (!c || !ix || !dst || dst_off + ix->n > dst->cap) return SP_EINVAL;
  HIPCHK(hipSetDevice(c->dev));
  ProfScope ps(c, PF_SPARK, 36.0 * (double)ix->n);
  hipLaunchKernelGGL(k_from_index, dim3((unsigned)grid_for(ix->n)), dim3(256), 0, c->stream, (const uint32_t*)ix->d, ix->n, dst->d + dst_off);
  return hipGetLastError() == hipSuccess ? SP_OK : SP_EHIP;
}
int32_t sp_table_view(sp_ctx* c, const sp_table* parent, size_t off, size_t len, sp_table** out) {
  if (!c || !parent || !out || len == 0 || off + len > parent->cap) return SP_EINVAL;
  sp_table* t = new (std::nothrow) sp_table();
  if (!t) return SP_ENOMEM;
  t->ctx = c; t->d = parent->d + off; t->cap = t->len = len; t->owner = 0; t->d_bytes = 0; t->alt = nullptr; t->alt_bytes = 0;
  *out = t;
  return SP_OK;
}
int32_t sp_gather(sp_ctx* c, const sp_table* mem, const sp_index* addr, sp_table* dst, size_t dst_off) {
  if (!c || !mem || !addr || !dst || dst_off + addr->n > dst->cap) return SP_EINVAL;
  HIPCHK(hipSetDevice(c->dev));
  ProfScope ps(c, PF_SPARK, 68.0 * (double)addr->n);
  hipLaunchKernelGGL(k_gather, dim3((unsigned)grid_for(addr->n)), dim3(256), 0, c->stream, (const Fq*)mem->d, (const uint32_t*)addr->d, addr->n,
                     dst->d + dst_off);
  return hipGetLastError() == hipSuccess ? SP_OK : SP_EHIP;
}
int32_t sp_hash_layer(sp_ctx* c, const sp_table* addr, const sp_table* val, const sp_table* ts, int ts_inc, size_t n, const uint64_t r_hash[4],
                      const uint64_t r_multiset[4], sp_table* dst, size_t dst_off) {
  if (!c || !val || !dst || !r_hash || !r_multiset || n == 0 || val->cap < n || (addr && addr->cap < n) || (ts && ts->cap < n) ||
      dst_off + n > dst->cap)
    return SP_EINVAL;
  HIPCHK(hipSetDevice(c->dev));
  Fq rh = limbs(r_hash), rm = limbs(r_multiset);
  Fq rh2 = fq_mul(rh, rh);
  ProfScope ps(c, PF_SPARK, 32.0 * (double)n * (2 + (addr ? 1 : 0) + (ts ? 1 : 0)));
  hipLaunchKernelGGL(k_hash_layer, dim3((unsigned)grid_for(n)), dim3(256), 0, c->stream, addr ? (const Fq*)addr->d : (const Fq*)nullptr,
                     (const Fq*)val->d, ts ? (const Fq*)ts->d : (const Fq*)nullptr, ts_inc, n, rh, rh2, rm, dst->d + dst_off);
  return hipGetLastError() == hipSuccess ? SP_OK : SP_EHIP;
}
// leaves [0, n) and layer 1 [n, n + n/2) of dst (and, dst_write != NULL, of the write-set circuit of the same matrix: ts + 1) in one pass;
// the rest of the tree: sp_product_tree_many_from(.., 1)
int32_t sp_hash_layer_first(sp_ctx* c, const sp_table* addr, const sp_table* val, const sp_table* ts, int ts_inc, size_t n, const uint64_t r_hash[4],
                            const uint64_t r_multiset[4], sp_table* dst, sp_table* dst_write) {
  if (!c || !val || !dst || !r_hash || !r_multiset || n < 4 || !is_pow2(n) || val->cap < n || (addr && addr->cap < n) || (ts && ts->cap < n) ||
      dst->cap < 2 * n || (dst_write && (dst_write->cap < 2 * n || dst_write == dst || ts_inc != 0)))
    return SP_EINVAL;
  HIPCHK(hipSetDevice(c->dev));
  Fq rh = limbs(r_hash), rm = limbs(r_multiset);
  Fq rh2 = fq_mul(rh, rh);
  ProfScope ps(c, PF_SPARK, 32.0 * (double)n * ((dst_write ? 3.0 : 1.5) + (addr ? 1 : 0) + (ts ? 1 : 0) + 1));
  const Fq* pa = addr ? (const Fq*)addr->d : (const Fq*)nullptr;
  const Fq* pt = ts ? (const Fq*)ts->d : (const Fq*)nullptr;
  if (dst_write)
    hipLaunchKernelGGL(k_hash_layer_first<true>, dim3((unsigned)grid_for(n / 2)), dim3(256), 0, c->stream, pa, (const Fq*)val->d, pt, 0, n, rh, rh2, rm, dst->d, dst_write->d);
  else
    hipLaunchKernelGGL(k_hash_layer_first<false>, dim3((unsigned)grid_for(n / 2)), dim3(256), 0, c->stream, pa, (const Fq*)val->d, pt, ts_inc, n, rh, rh2, rm, dst->d,
                       (Fq*)nullptr);
  return hipGetLastError() == hipSuccess ? SP_OK : SP_EHIP;
}
int32_t sp_product_tree(sp_ctx* c, sp_table* store, size_t n) {
  if (!c || !store || !is_pow2(n) || n < 2 || store->cap < 2 * n) return SP_EINVAL;
  HIPCHK(hipSetDevice(c->dev));
  size_t off = 0, len = n;
  while (len > 2) {
    size_t half = len / 2, noff = off + len;
    ProfScope ps(c, PF_SPARK, 48.0 * (double)len);
    hipLaunchKernelGGL(k_prod_layer, dim3((unsigned)grid_for(half)), dim3(256), 0, c->stream, (const Fq*)(store->d + off), half, store->d + noff);
    off = noff;
    len = half;
  }
  return hipGetLastError() == hipSuccess ? SP_OK : SP_EHIP;
}

int32_t sp_product_tree_many(sp_ctx* c, sp_table* const* stores, size_t count, size_t n) { return sp_product_tree_many_from(c, stores, count, n, 0); }
// layers_done: 0, or 1 when layer 1 is already in the stores (sp_hash_layer_first)
int32_t sp_product_tree_many_from(sp_ctx* c, sp_table* const* stores, size_t count, size_t n, size_t layers_done) {
  if (!c || !stores || count == 0 || !is_pow2(n) || n < 2 || layers_done > 1 || (layers_done == 1 && n < 4)) return SP_EINVAL;
  for (size_t k = 0; k < count; k++)
    if (!stores[k] || stores[k]->cap < 2 * n) return SP_EINVAL;
  HIPCHK(hipSetDevice(c->dev));
  for (size_t k0 = 0; k0 < count; k0 += 16) {
    size_t nk = count - k0 < 16 ? count - k0 : 16;
    Stores16 st;
    for (size_t k = 0; k < 16; k++) st.p[k] = k < nk ? stores[k0 + k]->d : nullptr;
    size_t off = 0, len = n;
    if (layers_done == 1) { off = n; len = n / 2; }
    while (len > 2) {
      size_t half = len / 2, noff = off + len;
      ProfScope ps(c, PF_SPARK, 48.0 * (double)len * (double)nk);
      if (len <= 2048) {  // the remaining layers in one launch
        hipLaunchKernelGGL(k_prod_layer_tail, dim3((unsigned)nk), dim3(256), 0, c->stream, st, off, len);
        break;
      }
      const bool two_layers = c->opt.v[OPT_SPARK_PROD_LAYER2] != 0;  // A/B switch
      const size_t l2max = (size_t)1 << (c->opt.v[OPT_SPARK_PROD_LAYER2_MAX_LOG2] < 13 ? 13 : c->opt.v[OPT_SPARK_PROD_LAYER2_MAX_LOG2]);
      if (two_layers && len <= l2max && len >= 8192) {  // launch-sized layers: two per launch (len / 4 >= 2048: the tail takes over below)
        hipLaunchKernelGGL(k_prod_layer2_many, dim3((unsigned)grid_for(len / 4, 1024), (unsigned)nk), dim3(256), 0, c->stream, st, off, len);
        off = off + len + len / 2;
        len = len / 4;
        continue;
      }
      hipLaunchKernelGGL(k_prod_layer_many, dim3((unsigned)grid_for(half, 1024), (unsigned)nk), dim3(256), 0, c->stream, st, off, half, noff);
      off = noff;
      len = half;
    }
  }
  return hipGetLastError() == hipSuccess ? SP_OK : SP_EHIP;
}

static int32_t batched_setup(sp_ctx* c, sp_table* const* A, sp_table* const* B, sp_table* const* C, size_t ninst, size_t* len_out, bool bind_c, TripleInline* IN,
                             const Triple** Tdev) {
  if (!c || !A || !B || !C || ninst == 0 || ninst > 64) return SP_EINVAL;
  size_t len = A[0] ? A[0]->len : 0;
  std::vector<Triple> T(ninst);
  for (size_t k = 0; k < ninst; k++) {
    if (!A[k] || !B[k] || !C[k] || A[k]->len != len || B[k]->len != len || C[k]->len != len) return SP_EINVAL;
    T[k] = Triple{A[k]->d, B[k]->d, C[k]->d, nullptr};
  }
  if (len < 2 || !is_pow2(len)) return SP_EINVAL;
  if (bind_c) {
    for (size_t k = 0; k < ninst; k++) {
      bool first = true;
      for (size_t m = 0; m < k; m++) first = first && C[m] != C[k];
      if (first) {
        SPCHK(table_ensure_alt(C[k], len / 2));
        T[k].c_out = C[k]->alt;
      }
    }
  }
  const bool inline_args = c->opt.v[OPT_SUMCHECK_INLINE_ARGS] != 0;  // A/B switch
  if (inline_args && ninst <= 24) {
    memcpy(IN->t, T.data(), sizeof(Triple) * ninst);
    *Tdev = nullptr;
  } else {
    stage_small(c, 0, T.data(), sizeof(Triple) * ninst);
    *Tdev = (const Triple*)c->hmap;
  }
  *len_out = len;
  return SP_OK;
}
// Few partials per instance (the late rounds of every layer: most of the ~400 rounds of a proof): the blocks write them
// straight into the host-mapped result page and the calling thread adds them — F_q additions are nanoseconds there,
// while a second launch to add them costs ~10 us on the critical path.
static bool host_sums(size_t nblk, size_t ninst) { return nblk == 1 || 96 * nblk * ninst <= HOST_SUM_BYTES; }
// sig: the completion signal of the trip — already raised by the evaluation kernel when it wrote its partials into the
// host page (on_host), raised by the reduction kernel launched here otherwise
static int32_t batched_finish(sp_ctx* c, Fq* partials, size_t nblk, size_t ninst, uint64_t* out, const DoneSig& sig) {
  if (partials != (Fq*)hres(c)) {
    {
      ProfScope ps(c, PF_REDUCE, 96.0 * (double)(nblk * ninst));
      hipLaunchKernelGGL(k_reduce_partials_batched, dim3((unsigned)ninst), dim3(256), 0, c->stream, (const Fq*)partials, nblk, 3, (Fq*)hres(c), sig);
    }
    SPCHK(sig_wait(c, sig));
    memcpy(out, hres(c), 96 * ninst);
    return hipGetLastError() == hipSuccess ? SP_OK : SP_EHIP;
  }
  SPCHK(sig_wait(c, sig));
  if (hipGetLastError() != hipSuccess) return SP_EHIP;
  const Fq* p = (const Fq*)hres(c);  // [inst][blk][3]
  Fq* o = (Fq*)out;
  for (size_t i = 0; i < ninst; i++)
    for (int k = 0; k < 3; k++) {
      Fq acc = p[(i * nblk) * 3 + k];
      for (size_t b = 1; b < nblk; b++) acc = fq_add(acc, p[(i * nblk + b) * 3 + k]);
      o[3 * i + k] = acc;
    }
  return SP_OK;
}
int32_t sp_sumcheck_eval_batched(sp_ctx* c, sp_table* const* A, sp_table* const* B, sp_table* const* C, size_t ninst, uint64_t* out) {
  if (!out) return SP_EINVAL;
  size_t len;
  HIPCHK(hipSetDevice(c ? c->dev : 0));
  TripleInline IN;
  const Triple* Tdev = nullptr;
  SPCHK(batched_setup(c, A, B, C, ninst, &len, false, &IN, &Tdev));
  size_t half = len / 2;
  bool tiny = half <= 8192;
  size_t nblk = tiny ? (half + 63) / 64 : grid_for(half, 256);
  SPCHK(ensure(&c->scratch, &c->scratch_cap, 32 * 3 * (nblk + 1) * ninst));
  bool on_host = host_sums(nblk, ninst) && tiny;
  Fq* partials = on_host ? (Fq*)hres(c) : (Fq*)c->scratch;
  DoneSig sig = sig_make(c, on_host ? nblk * ninst : ninst);
  {
    ProfScope ps(c, PF_SC_EVAL, 96.0 * (double)len * (double)ninst, nullptr, 6.0 * (double)half * (double)ninst);
    if (tiny) hipLaunchKernelGGL(k_cubic_eval_tiny, dim3((unsigned)nblk, (unsigned)ninst), dim3(256), 0, c->stream, Tdev, IN, half, partials, on_host ? sig : sig_none());
    else hipLaunchKernelGGL(k_cubic_eval_batched, dim3((unsigned)nblk, (unsigned)ninst), dim3(256), 0, c->stream, Tdev, IN, half, partials);
  }
  return batched_finish(c, partials, nblk, ninst, out, sig);
}
int32_t sp_sumcheck_bind_eval_batched(sp_ctx* c, sp_table* const* A, sp_table* const* B, sp_table* const* C, size_t ninst, const uint64_t r[4],
                                      uint64_t* out) {
  if (!out || !r) return SP_EINVAL;
  size_t len;
  HIPCHK(hipSetDevice(c ? c->dev : 0));
  if (!c) return SP_EINVAL;
  for (size_t k = 0; k < ninst && C; k++)
    if (C[k] && C[k]->len < 4) return SP_EINVAL;
  TripleInline IN;
  const Triple* Tdev = nullptr;
  SPCHK(batched_setup(c, A, B, C, ninst, &len, true, &IN, &Tdev));
  if (len < 4) return SP_EINVAL;
  size_t quarter = len / 4;
  bool tiny = quarter <= 8192;  // one index per 8 lanes while the round is latency-bound (512 / 2048 / 8192 / 32768 measured: 45.3 / 45.0 / 44.8 / 45.6 ms per proof)
  size_t nblk = tiny ? (quarter + 31) / 32 : grid_for(quarter, 256);
  SPCHK(ensure(&c->scratch, &c->scratch_cap, 32 * 3 * (nblk + 1) * ninst));
  bool on_host = host_sums(nblk, ninst) && tiny;
  Fq* partials = on_host ? (Fq*)hres(c) : (Fq*)c->scratch;
  DoneSig sig = sig_make(c, on_host ? nblk * ninst : ninst);
  {
    ProfScope ps(c, PF_SC_BIND_EVAL, (96.0 + 32.0) * (double)len * (double)ninst, nullptr, 12.0 * (double)quarter * (double)ninst);
    if (tiny)
      hipLaunchKernelGGL(k_cubic_bind_eval_tiny, dim3((unsigned)nblk, (unsigned)ninst), dim3(256), 0, c->stream, Tdev, IN, quarter,
                         limbs(r), partials, on_host ? sig : sig_none());
    else
      hipLaunchKernelGGL(k_cubic_bind_eval_batched, dim3((unsigned)nblk, (unsigned)ninst), dim3(256), 0, c->stream, Tdev, IN, quarter,
                         limbs(r), partials);
  }
  // A_k and B_k are now bound in place (distinct tables assumed for A and B); each distinct C was bound into its
  // alternate buffer by the first instance that uses it: make that buffer current
  for (size_t k = 0; k < ninst; k++) { A[k]->len = len / 2; B[k]->len = len / 2; }
  for (size_t k = 0; k < ninst; k++)
    if (C[k]->len == len) table_swap_to_alt(C[k], len / 2);
  return batched_finish(c, partials, nblk, ninst, out, sig);
}

// ---- the factored forms (kernels above): instances [0, neq) are product-circuit instances sharing the eq table C[0] (never bound, its
// length stays what it was: the rounds read its leading entries), instances [neq, ninst) are generic. out[4 * ninst]: {q(0), q(2), 0, 0} for
// the former, {e(0), e(1), e(2), e(3)} for the latter. Throughput-sized tables only (the latency forms have no factored twin): SP_EINVAL
// below 65536 entries, the caller hands over to the generic rounds there (sp_table_scale_prefix).
static int32_t batched_finish4(sp_ctx* c, Fq* partials, size_t nblk, size_t ninst, uint64_t* out, const DoneSig& sig) {
  {
    ProfScope ps(c, PF_REDUCE, 128.0 * (double)(nblk * ninst));
    hipLaunchKernelGGL(k_reduce_partials_batched, dim3((unsigned)ninst), dim3(256), 0, c->stream, (const Fq*)partials, nblk, 4, (Fq*)hres(c), sig);
  }
  SPCHK(sig_wait(c, sig));
  memcpy(out, hres(c), 128 * ninst);
  return hipGetLastError() == hipSuccess ? SP_OK : SP_EHIP;
}
static int32_t eq_form_check(sp_table* const* C, size_t ninst, size_t neq, size_t len) {
  if (neq == 0 || neq > ninst || ninst > 24 || 128 * ninst > HOST_SUM_BYTES) return SP_EINVAL;
  for (size_t k = 0; k < neq; k++)
    if (C[k] != C[0]) return SP_EINVAL;
  if (!C[0] || C[0]->len < len) return SP_EINVAL;  // its leading `len` entries are read
  for (size_t k = neq; k < ninst; k++)
    if (!C[k] || C[k]->len != len || C[k] == C[0]) return SP_EINVAL;
  return SP_OK;
}
// batched_setup wants every C at the tables' length: the shared eq table keeps its original length, so the triples are built here
static int32_t eq_form_setup(sp_ctx* c, sp_table* const* A, sp_table* const* B, sp_table* const* C, size_t ninst, size_t neq, bool bind_c, size_t* len_out,
                             TripleInline* IN) {
  if (!c || !A || !B || !C || ninst == 0) return SP_EINVAL;
  const size_t len = A[0] ? A[0]->len : 0;
  if (len < 65536 || !is_pow2(len)) return SP_EINVAL;
  SPCHK(eq_form_check(C, ninst, neq, len));
  for (size_t k = 0; k < ninst; k++) {
    if (!A[k] || !B[k] || A[k]->len != len || B[k]->len != len) return SP_EINVAL;
    IN->t[k] = Triple{A[k]->d, B[k]->d, C[k]->d, nullptr};
    if (bind_c && k >= neq) {
      bool first = true;
      for (size_t m = neq; m < k; m++) first = first && C[m] != C[k];
      if (first) {
        SPCHK(table_ensure_alt(C[k], len / 2));
        IN->t[k].c_out = C[k]->alt;
      }
    }
  }
  *len_out = len;
  return SP_OK;
}
int32_t sp_sumcheck_eval_batched_eq(sp_ctx* c, sp_table* const* A, sp_table* const* B, sp_table* const* C, size_t ninst, size_t neq, uint64_t* out) {
  if (!c || !out) return SP_EINVAL;
  HIPCHK(hipSetDevice(c->dev));
  size_t len;
  TripleInline IN;
  SPCHK(eq_form_setup(c, A, B, C, ninst, neq, false, &len, &IN));
  const size_t half = len / 2, nblk = grid_for(half, 256);
  SPCHK(ensure(&c->scratch, &c->scratch_cap, 32 * 4 * (nblk + 1) * ninst));
  Fq* partials = (Fq*)c->scratch;
  DoneSig sig = sig_make(c, ninst);
  {
    ProfScope ps(c, PF_SC_EVAL, 32.0 * (double)len * (2.5 * (double)neq + 3.0 * (double)(ninst - neq)), nullptr,
                 (double)half * (4.0 * (double)neq + 8.0 * (double)(ninst - neq)));
    hipLaunchKernelGGL(k_cubic_eval_batched_eq<false>, dim3((unsigned)nblk, (unsigned)neq), dim3(256), 0, c->stream, IN, 0u, half, partials);
    if (ninst > neq)
      hipLaunchKernelGGL(k_cubic_eval_batched_eq<true>, dim3((unsigned)nblk, (unsigned)(ninst - neq)), dim3(256), 0, c->stream, IN, (unsigned)neq, half, partials);
  }
  return batched_finish4(c, partials, nblk, ninst, out, sig);
}
int32_t sp_sumcheck_bind_eval_batched_eq(sp_ctx* c, sp_table* const* A, sp_table* const* B, sp_table* const* C, size_t ninst, size_t neq, const uint64_t r[4],
                                         uint64_t* out) {
  if (!c || !out || !r) return SP_EINVAL;
  HIPCHK(hipSetDevice(c->dev));
  size_t len;
  TripleInline IN;
  SPCHK(eq_form_setup(c, A, B, C, ninst, neq, true, &len, &IN));
  const size_t quarter = len / 4, nblk = grid_for(quarter, 256);
  SPCHK(ensure(&c->scratch, &c->scratch_cap, 32 * 4 * (nblk + 1) * ninst));
  Fq* partials = (Fq*)c->scratch;
  DoneSig sig = sig_make(c, ninst);
  {
    ProfScope ps(c, PF_SC_BIND_EVAL, 32.0 * (double)len * (3.25 * (double)neq + 4.0 * (double)(ninst - neq)), nullptr,
                 (double)quarter * (8.0 * (double)neq + 14.0 * (double)(ninst - neq)));
    hipLaunchKernelGGL(k_cubic_bind_eval_batched_eq<false>, dim3((unsigned)nblk, (unsigned)neq), dim3(256), 0, c->stream, IN, 0u, quarter, limbs(r), partials);
    if (ninst > neq)
      hipLaunchKernelGGL(k_cubic_bind_eval_batched_eq<true>, dim3((unsigned)nblk, (unsigned)(ninst - neq)), dim3(256), 0, c->stream, IN, (unsigned)neq, quarter,
                         limbs(r), partials);
  }
  for (size_t k = 0; k < ninst; k++) { A[k]->len = len / 2; B[k]->len = len / 2; }
  for (size_t k = neq; k < ninst; k++)
    if (C[k]->len == len) table_swap_to_alt(C[k], len / 2);
  return batched_finish4(c, partials, nblk, ninst, out, sig);
}
// t[i] *= k for i < n, and n becomes the table's length: the hand-over from the factored rounds to the generic ones (queued, not waited for)
int32_t sp_table_scale_prefix(sp_ctx* c, sp_table* t, size_t n, const uint64_t k[4]) {
  if (!c || !t || !k || n == 0 || n > t->len) return SP_EINVAL;
  HIPCHK(hipSetDevice(c->dev));
  {
    ProfScope ps(c, PF_MISC, 64.0 * (double)n);
    hipLaunchKernelGGL(k_scale_prefix, dim3((unsigned)grid_for(n)), dim3(256), 0, c->stream, t->d, n, limbs(k));
  }
  t->len = n;
  return hipGetLastError() == hipSuccess ? SP_OK : SP_EHIP;
}

// partials[ninst][nblk][18] -> out[ninst][18], one block per instance: thread = (component k < 18 of 32, slice of blocks)
__global__ void __launch_bounds__(256) k_reduce_partials18(const Fq* __restrict__ partials, size_t nblk, Fq* __restrict__ out, DoneSig sig, AheadArgs ah) { SP_FG_PRIO();
  if (ah.bell && (__hip_atomic_load(ah.decision, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != ((ah.seq << 2) | 1u))) return;  // behind a launch that gave up
  __shared__ Fq sm[8][18];
  const int k = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const Fq* p = partials + (size_t)blockIdx.x * nblk * 18;
  if (k < 18) {
    Fq acc = fq_zero();
    for (size_t b = sl; b < nblk; b += 8) acc = fq_add(acc, ld_fq(p + b * 18 + k));
    sm[sl][k] = acc;
  }
  __syncthreads();
  if (threadIdx.x < 18) {
    Fq acc = sm[0][threadIdx.x];
#pragma unroll
    for (int j = 1; j < 8; j++) acc = fq_add(acc, sm[j][threadIdx.x]);
    st_fq(out + (size_t)blockIdx.x * 18 + threadIdx.x, acc);
  }
  signal_done(sig);
}
constexpr size_t TAIL_OFF = 12288, TAIL_MAX_INST = 21;  // the <= 8-entry tables of <= 21 instances behind the 18 sums per instance in the result page
// The geometry of one k_cubic_bind2_eval trip over tables of length `len` (before its binds)
struct Bind2Shape {
  size_t n2, nblk;
  bool host, tail;
};
static Bind2Shape bind2_shape(size_t len, int nbind, size_t ninst, bool want_tables) {
  Bind2Shape s;
  s.n2 = len >> nbind;
  const size_t np = s.n2 < 4 ? s.n2 : 4, ng = s.n2 / np;
  s.nblk = (ng + 7) / 8;
  s.host = 32 * 18 * s.nblk * ninst <= HOST_SUM_BYTES;
  s.tail = want_tables && s.n2 >= 2 && s.n2 <= 8 && ninst <= TAIL_MAX_INST && s.host && s.nblk == 1;
  return s;
}
// enqueue the kernel(s) of one trip on the main stream; `ah.bell` set: ahead of its challenges (r0, r1 unused)
static void bind2_enqueue(sp_ctx* c, const Bind2Inline& IN, bool inl, const Fq* dweights, size_t len, int nbind, const Fq& r0, const Fq& r1, size_t ninst,
                          const Bind2Shape& sh, const DoneSig& sig, const AheadArgs& ah) {
  const size_t ng = sh.n2 / (sh.n2 < 4 ? sh.n2 : 4);
  Fq* partials = sh.host ? (Fq*)hres(c) : (Fq*)c->scratch;
  Fq* dump = sh.tail ? (Fq*)(hres(c) + TAIL_OFF) : nullptr;
  {
    ProfScope ps(c, nbind ? PF_SC_BIND_EVAL : PF_SC_EVAL, 96.0 * (double)len * (double)ninst, nullptr, (nbind ? 36.0 + 36.0 : 36.0) * (double)ng * (double)ninst);
    hipLaunchKernelGGL(k_cubic_bind2_eval, dim3((unsigned)sh.nblk, (unsigned)ninst), dim3(256), 0, c->stream, inl ? (const Triple2*)nullptr : (const Triple2*)c->hmap, IN, dweights, len,
                       nbind, r0, r1, partials, dump, sh.host ? sig : sig_none(), ah);
  }
  if (!sh.host) {
    ProfScope ps(c, PF_REDUCE, 32.0 * 18 * (double)(sh.nblk * ninst));
    hipLaunchKernelGGL(k_reduce_partials18, dim3((unsigned)ninst), dim3(256), 0, c->stream, (const Fq*)partials, sh.nblk, (Fq*)hres(c), sig, ah);
  }
}
// shared by the entry points below: one trip of k_cubic_bind2_eval, the 18 sums per instance brought to the host. Since round 6 the kernel of the
// NEXT trip is enqueued before this one's results are waited for whenever the next trip is known to be a two-bind trip over the same tables
// (the bound tables still have >= 4 entries): see AheadArm in internal.hpp. The next call rings it if it is that trip; anything else cancels it.
static int32_t bind2_launch(sp_ctx* c, sp_table* const* A, sp_table* const* B, sp_table* const* C, size_t ninst, int nbind, const uint64_t* r0, const uint64_t* r1,
                            const uint64_t* weights, uint64_t* out_evals, uint64_t* out_coeffs, uint64_t* out_heads, uint64_t* out_tables = nullptr) {
  if (!c || !A || !B || !C || ninst == 0 || ninst > 64) return SP_EINVAL;
  HIPCHK(hipSetDevice(c->dev));
  size_t len = A[0] ? A[0]->len : 0;
  if (len < ((size_t)1 << nbind) || len < 2 || !is_pow2(len)) return SP_EINVAL;
  const int do_bind = nbind;
  const size_t n2 = len >> nbind;
  if ((n2 >= 2 && !out_evals) || (n2 >= 4 && !out_coeffs) || (n2 == 1 && !out_heads)) return SP_EINVAL;
  for (size_t k = 0; k < ninst; k++)
    if (!A[k] || !B[k] || !C[k] || A[k]->len != len || B[k]->len != len || C[k]->len != len) return SP_EINVAL;
  const bool inline_args = c->opt.v[OPT_SUMCHECK_INLINE_ARGS] != 0;  // A/B switch
  const bool inl = inline_args && ninst <= 24;
  AheadArm& arm = c->ahead;
  // the instances' tables; a C shared between instances is bound once, out of place
  std::vector<Triple2> T(ninst);
  std::vector<sp_table*> distinctC;
  auto triples = [&](size_t out_len) -> int32_t {
    distinctC.clear();
    for (size_t k = 0; k < ninst; k++) {
      T[k] = Triple2{A[k]->d, B[k]->d, C[k]->d, nullptr};
      bool first = true;
      for (size_t m = 0; m < k; m++) first = first && C[m] != C[k];
      if (first) {
        distinctC.push_back(C[k]);
        if (out_len) {
          SPCHK(table_ensure_alt(C[k], out_len));
          T[k].c_out = C[k]->alt;
        }
      }
    }
    return SP_OK;
  };
  SPCHK(triples(do_bind ? n2 : 0));
  // is this the trip an enqueued kernel is waiting for? The same handles, the same device buffers behind them, the same length, weights and form.
  bool rung = false;
  if (arm.on) {
    bool same = inl && nbind == 2 && r0 && r1 && arm.ninst == ninst && arm.len == len && arm.weighted == (weights != nullptr) &&
                (!weights || memcmp(arm.w, weights, 32 * ninst) == 0);
    for (size_t k = 0; same && k < ninst; k++)
      same = arm.A[k] == A[k] && arm.B[k] == B[k] && arm.C[k] == C[k] && arm.buf[k][0] == T[k].a && arm.buf[k][1] == T[k].b && arm.buf[k][2] == T[k].c &&
             arm.buf[k][3] == T[k].c_out;
    if (same) rung = true;
    else ahead_cancel(c);
  }
  Bind2Inline IN;
  const Fq* dweights = nullptr;
  auto stage = [&]() {
    if (inl) {
      memcpy(IN.t, T.data(), sizeof(Triple2) * ninst);
      if (weights) { memcpy(IN.w, weights, 32 * ninst); dweights = (const Fq*)c->hmap; }  // non-null: "weighted"; the values come from IN.w
    } else {
      stage_small(c, 0, T.data(), sizeof(Triple2) * ninst);
      dweights = weights ? (const Fq*)stage_small(c, sizeof(Triple2) * 64, weights, 32 * ninst) : nullptr;
    }
  };
  const Bind2Shape sh = bind2_shape(len, nbind, ninst, out_tables != nullptr);
  const Fq z = fq_zero();
  DoneSig sig;
  uint32_t rung_seq = 0;
  if (rung) {
    // the kernel is in the stream already: hand it the challenges
    sig = DoneSig{c->done_flag, c->done_counter, arm.sig_seq, arm.sig_total, nullptr};
    rung_seq = arm.seq;
    arm.on = false;
    arm.n_rung++;
    uint32_t words[16], fold = rung_seq;
    memcpy(words, r0, 32);
    memcpy(words + 8, r1, 32);
    for (int k = 0; k < 16; k++) { c->bell->r[k] = words[k]; fold ^= words[k]; }
    c->bell->check = fold;
    if (c->opt.v[OPT_SUMCHECK_LAUNCH_AHEAD] != 2) __atomic_store_n(&c->bell->bell, rung_seq << 1, __ATOMIC_RELEASE);  // 2: the test hook never rings
  } else {
    stage();
    SPCHK(ensure(&c->scratch, &c->scratch_cap, 32 * 18 * (sh.nblk + 1) * ninst));
    sig = sig_make(c, sh.host ? sh.nblk * ninst : ninst);  // raised by the last kernel of the trip
    bind2_enqueue(c, IN, inl, dweights, len, nbind, r0 ? limbs(r0) : z, r1 ? limbs(r1) : z, ninst, sh, sig, ahead_none());
  }
  const bool tail = rung ? (arm.tail && out_tables) : sh.tail;
  // the tables as this trip leaves them (host bookkeeping: the kernels are stream-ordered)
  if (do_bind) {
    for (size_t k = 0; k < ninst; k++) { A[k]->len = n2; B[k]->len = n2; }
    for (sp_table* t : distinctC) table_swap_to_alt(t, n2);
  }
  // the next trip, ahead of its challenges: a two-bind trip over the same tables follows whenever they still have >= 4 entries (and were not handed
  // over for the last rounds on the host core)
  const bool prof_here = c->prof_on != 0 && (((c->prof_mask >> PF_SC_BIND_EVAL) & 1) || ((c->prof_mask >> PF_REDUCE) & 1));
  if (c->opt.v[OPT_SUMCHECK_LAUNCH_AHEAD] != 0 && c->bell && c->done_counter && !c->ktime && !prof_here && inl && n2 >= 4 && !tail) {
    const Bind2Shape nx = bind2_shape(n2, 2, ninst, true);
    if (ensure(&c->scratch, &c->scratch_cap, 32 * 18 * (nx.nblk + 1) * ninst) == SP_OK && c->scratch_cap >= 32 * 18 * (sh.nblk + 1) * ninst) {
      std::vector<Triple2> Tkeep = T;
      if (triples(nx.n2) == SP_OK) {
        stage();
        const DoneSig nsig = sig_make(c, nx.host ? nx.nblk * ninst : ninst);
        const AheadArgs ah{c->bell, c->done_counter + 64, c->done_counter + 80, c->done_flag + 16, ++c->ahead_seq};
        bind2_enqueue(c, IN, inl, dweights, n2, 2, z, z, ninst, nx, nsig, ah);
        arm.on = true;
        arm.n_armed++;
        arm.seq = ah.seq;
        arm.ninst = ninst; arm.len = n2; arm.nblk = nx.nblk; arm.host = nx.host; arm.tail = nx.tail; arm.weighted = weights != nullptr;
        for (size_t k = 0; k < ninst; k++) {
          arm.A[k] = A[k]; arm.B[k] = B[k]; arm.C[k] = C[k];
          arm.buf[k][0] = T[k].a; arm.buf[k][1] = T[k].b; arm.buf[k][2] = T[k].c; arm.buf[k][3] = T[k].c_out;
        }
        if (weights) memcpy(arm.w, weights, 32 * ninst);
        arm.sig_seq = nsig.seq; arm.sig_total = nsig.total;
      }
      T = Tkeep;
    }
  }
  // wait for this trip
  if (rung) {
    bool gave_up = false;
    for (uint64_t spins = 1;; spins++) {
      if (*c->done_flag == sig.seq) { c->sync_epoch++; break; }
      if (c->done_flag[16] == rung_seq) { gave_up = true; break; }
      if ((spins & 0xFFFFFF) == 0 && hipStreamQuery(c->stream.s) != hipErrorNotReady) {
        if (*c->done_flag == sig.seq) { c->sync_epoch++; break; }
        if (c->done_flag[16] == rung_seq) { gave_up = true; break; }
        return SP_EHIP;
      }
    }
    if (gave_up) {
      arm.n_gave_up++;
      // nothing was touched: the trip again, the ordinary way (behind whatever was enqueued for the trip after it, which is told to give up too)
      ahead_cancel(c);
      memcpy(IN.t, T.data(), sizeof(Triple2) * ninst);
      if (weights) { memcpy(IN.w, weights, 32 * ninst); dweights = (const Fq*)c->hmap; }
      sig = sig_make(c, sh.host ? sh.nblk * ninst : ninst);
      bind2_enqueue(c, IN, true, dweights, len, nbind, limbs(r0), limbs(r1), ninst, sh, sig, ahead_none());
      SPCHK(sig_wait(c, sig));
    }
  } else {
    SPCHK(sig_wait(c, sig));
  }
  const Bind2Shape& got = sh;
  std::vector<Fq> sums(18 * ninst);
  if (!got.host) {
    memcpy(sums.data(), hres(c), 32 * 18 * ninst);
  } else {
    const Fq* p = (const Fq*)hres(c);
    for (size_t i = 0; i < ninst; i++)
      for (int k = 0; k < 18; k++) {
        Fq acc = p[(i * got.nblk) * 18 + k];
        for (size_t b = 1; b < got.nblk; b++) acc = fq_add(acc, p[(i * got.nblk + b) * 18 + k]);
        sums[18 * i + k] = acc;
      }
  }
  if (hipGetLastError() != hipSuccess) return SP_EHIP;
  if (out_tables) {  // [ninst][3][n2], or nothing (first word all ones) when the tables are longer than 8 or there are too many instances
    if (tail) {
      const Fq* d = (const Fq*)(hres(c) + TAIL_OFF);
      for (size_t i = 0; i < ninst; i++)
        for (int k = 0; k < 3; k++) memcpy(out_tables + 4 * ((i * 3 + k) * n2), d + (i * 3 + k) * 8, 32 * n2);
    } else {
      out_tables[0] = ~0ULL;
    }
  }
  if (weights) {  // the instances' weighted sums added up: 3 evaluations and 12 coefficients in all
    Fq tot[15];
    for (int k = 0; k < 15; k++) {
      Fq acc = sums[k];
      for (size_t i = 1; i < ninst; i++) acc = fq_add(acc, sums[18 * i + k]);
      tot[k] = acc;
    }
    if (n2 >= 2) memcpy(out_evals, tot, 96);
    if (n2 >= 4) memcpy(out_coeffs, tot + 3, 384);
  }
  for (size_t i = 0; i < ninst; i++) {
    if (!weights && n2 >= 2) memcpy(out_evals + 12 * i, &sums[18 * i], 96);
    if (!weights && n2 >= 4) memcpy(out_coeffs + 48 * i, &sums[18 * i + 3], 384);
    if (n2 == 1) { memcpy(out_heads + 8 * i, &sums[18 * i + 15], 64); }
  }
  if (n2 == 1)
    for (size_t k = 0; k < distinctC.size(); k++) {
      size_t owner = 0;
      while (C[owner] != distinctC[k]) owner++;
      memcpy(out_heads + 8 * ninst + 4 * k, &sums[18 * owner + 17], 32);
    }
  return SP_OK;
}
int32_t sp_sumcheck_bind2_eval_batched(sp_ctx* c, sp_table* const* A, sp_table* const* B, sp_table* const* C, size_t ninst, const uint64_t r0[4],
                                       const uint64_t* r1, const uint64_t* weights, uint64_t* out_evals, uint64_t* out_coeffs, uint64_t* out_heads) {
  if (!r0) return SP_EINVAL;
  return bind2_launch(c, A, B, C, ninst, r1 ? 2 : 1, r0, r1, weights, out_evals, out_coeffs, out_heads);
}
int32_t sp_sumcheck_eval_coeffs_batched(sp_ctx* c, sp_table* const* A, sp_table* const* B, sp_table* const* C, size_t ninst, const uint64_t* weights,
                                        uint64_t* out_evals, uint64_t* out_coeffs) {
  return bind2_launch(c, A, B, C, ninst, 0, nullptr, nullptr, weights, out_evals, out_coeffs, nullptr);
}
int32_t sp_sumcheck_bind2_eval_tables_batched(sp_ctx* c, sp_table* const* A, sp_table* const* B, sp_table* const* C, size_t ninst, const uint64_t* r0,
                                              const uint64_t* r1, const uint64_t* weights, uint64_t* out_evals, uint64_t* out_coeffs, uint64_t* out_heads,
                                              uint64_t* out_tables) {
  if (!out_tables || (r1 && !r0)) return SP_EINVAL;
  return bind2_launch(c, A, B, C, ninst, r0 ? (r1 ? 2 : 1) : 0, r0, r1, weights, out_evals, out_coeffs, out_heads, out_tables);
}
int32_t sp_dot_many(sp_ctx* c, const sp_table* chi, sp_table* const* tabs, size_t nt, uint64_t* out) {
  if (!c || !chi || !tabs || !out || nt == 0 || nt > 64) return SP_EINVAL;
  size_t n = chi->len;
  std::vector<Fq*> ptrs(nt);
  for (size_t k = 0; k < nt; k++) {
    if (!tabs[k] || tabs[k]->cap < n) return SP_EINVAL;
    ptrs[k] = tabs[k]->d;
  }
  HIPCHK(hipSetDevice(c->dev));
  stage_small(c, 0, ptrs.data(), 8 * nt);
  size_t nblk = grid_for(n, 256);
  SPCHK(ensure(&c->scratch, &c->scratch_cap, 32 * (nblk + 1) * nt));
  Fq* partials = (Fq*)c->scratch;
  {
    ProfScope ps(c, PF_DOT, 32.0 * (double)n * (double)(nt + 1), nullptr, (double)n * (double)nt);
    hipLaunchKernelGGL(k_dot_many, dim3((unsigned)nblk, (unsigned)nt), dim3(256), 0, c->stream, (const Fq*)chi->d, (Fq* const*)c->hmap, n, partials);
  }
  {
    ProfScope ps(c, PF_REDUCE, 32.0 * (double)(nblk * nt));
    hipLaunchKernelGGL(k_reduce_partials_batched, dim3((unsigned)nt), dim3(256), 0, c->stream, (const Fq*)partials, nblk, 1, (Fq*)hres(c), sig_none());
  }
  SPCHK(fetch_small(c, out, 32 * nt));
  return hipGetLastError() == hipSuccess ? SP_OK : SP_EHIP;
}
int32_t sp_dot3_many(sp_ctx* c, const sp_table* const* l, const sp_table* const* r, const sp_table* const* w, size_t nt, size_t n, uint64_t* out) {
  if (!c || !l || !r || !w || !out || nt == 0 || nt > 64 || n == 0 || 24 * nt > HMAP_GEN) return SP_EINVAL;
  std::vector<const Fq*> ptrs(3 * nt);
  for (size_t k = 0; k < nt; k++) {
    if (!l[k] || !r[k] || !w[k] || l[k]->cap < n || r[k]->cap < n || w[k]->cap < n) return SP_EINVAL;
    ptrs[k] = l[k]->d; ptrs[nt + k] = r[k]->d; ptrs[2 * nt + k] = w[k]->d;
  }
  HIPCHK(hipSetDevice(c->dev));
  stage_small(c, 0, ptrs.data(), 8 * ptrs.size());
  size_t nblk = grid_for(n, 1024);
  SPCHK(ensure(&c->scratch, &c->scratch_cap, 32 * (nblk + 1) * nt));
  Fq* partials = (Fq*)c->scratch;
  DoneSig sig = sig_make(c, nt);
  {
    ProfScope ps(c, PF_DOT, 96.0 * (double)n * (double)nt, nullptr, 2.0 * (double)n * (double)nt);
    hipLaunchKernelGGL(k_dot3_many, dim3((unsigned)nblk, (unsigned)nt), dim3(256), 0, c->stream, (const Fq* const*)c->hmap, nt, n, partials);
  }
  {
    ProfScope ps(c, PF_REDUCE, 32.0 * (double)(nblk * nt));
    hipLaunchKernelGGL(k_reduce_partials_batched, dim3((unsigned)nt), dim3(256), 0, c->stream, (const Fq*)partials, nblk, 1, (Fq*)hres(c), sig);
  }
  SPCHK(sig_wait(c, sig));
  memcpy(out, hres(c), 32 * nt);
  return hipGetLastError() == hipSuccess ? SP_OK : SP_EHIP;
}
int32_t sp_dot3(sp_ctx* c, const sp_table* l, const sp_table* r, const sp_table* w, size_t off, size_t n, uint64_t out[4]) {
  if (!c || !l || !r || !w || !out || n == 0 || off + n > l->cap || off + n > r->cap || off + n > w->cap) return SP_EINVAL;
  HIPCHK(hipSetDevice(c->dev));
  size_t nblk = grid_for(n, 1024);
  SPCHK(ensure(&c->scratch, &c->scratch_cap, 32 * (nblk + 1)));
  Fq* partials = partials_dst(c, nblk, 1);
  {
    ProfScope ps(c, PF_DOT, 96.0 * (double)n, nullptr, 2.0 * (double)n);
    hipLaunchKernelGGL(k_dot3, dim3((unsigned)nblk), dim3(256), 0, c->stream, (const Fq*)(l->d + off), (const Fq*)(r->d + off), (const Fq*)(w->d + off), n,
                       partials);
  }
  return reduce_and_fetch(c, partials, nblk, 1, out);
}

}  // extern "C"
