// EXPERIMENT RECORD (not compiled into the product; needs spartan_amd/csrc/spark.hip's Triple2 / block helpers around it): the
// three-rounds-per-trip "grid" form of the batched cubic sum-check (round 3). 329 instead of 356 trips per 2^20 proof, 27.99 against
// 27.69 ms: 64 products per entry group cost more than the trip they save. Removed from the library in round 5
// with its entry point sp_sumcheck_grid_batched; the host half is cubic_grid_host.inc.
// ---- up to THREE rounds per launch: the sums of a whole grid -----------------------------------------------------------
// Generalisation of the two-round kernel above. For the tables as bound by this call (length n2), let
//   F(y_0, .., y_{kd-1}) = sum_z prod_{T in (A, B, C)} T~(y_0, .., y_{kd-1}, z)
// where T~ is multilinear in the top kd index bits (entry index = y_0 n2/2 + y_1 n2/4 + .. + z). F has degree 3 in every y, so its
// values on the grid {0, 1, 2, 3}^kd determine it, and the next kd rounds of prove_cubic_batched (sumcheck.rs:287-393) follow
// from them on the host: round j sends s_j(t) = sum_{b in {0,1}^(kd-1)} F(t, b); with its challenge r_j the grid is contracted
// along the first axis (Lagrange basis on {0, 1, 2, 3}); round j+1 sends sum_b F(r_j, t, b); and so on. Exact field arithmetic:
// the evaluations at t = 0, 2, 3 are the reference's, and the value at t = 1 equals e - s(0) as the sum-check invariant says.
// One trip advances kd <= 3 rounds (64 sums per call, weighted by the batching coefficients and summed over the instances on
// the device), and the same launch first binds the tables at the <= 3 challenges of the previous trip.
// Block = 2 groups x 128 lanes; group <-> one z, i.e. 2^kd entries per table. Stages (each a few lane-tasks, see k_cubic_bind2_eval
// for why lines are extended by separate lanes): binds (1 multiplication each, nbind levels), then axis kd-1 .. 0 extended from
// {0, 1} to {0, 1, 2, 3} (3 additions per pair), then 4^kd triple products.
struct GridArgs {
  const Triple2* T;
  const Fq* weights;
  size_t len;
  int nbind, kd;
  Fq r[3];
  Fq* part;            // device: [ninst][nblk][64] block sums, then [ninst][64] instance sums behind them
  uint32_t* tickets;   // device: [ninst] + 1, zero between launches
  Fq* out;             // host page: 4^kd sums
  Fq* dump;            // host page or null: [ninst][3][8] the bound tables when they have <= 8 entries
  unsigned nblk, ninst;
};
__global__ void __launch_bounds__(256) k_cubic_grid(GridArgs A, DoneSig sig) {
  __shared__ Fq wb[2][3][8][8];    // [group][table][p][c]: entries being bound (c = the not yet bound challenge bits)
  __shared__ Fq va[2][3][64], vb[2][3][64];  // grid values per table, ping-pong between the axis extensions
  __shared__ Fq red[2][64];
  __shared__ unsigned ticket;
  const Triple2 t = A.T[blockIdx.y];
  const Fq wv = ld_fq(A.weights + blockIdx.y);
  const int grp = threadIdx.x >> 7, gl = threadIdx.x & 127;
  const int nbind = A.nbind, kd = A.kd;
  const size_t len = A.len, n2 = len >> nbind;
  const int np = 1 << kd;                       // entries per table per group
  const size_t ng = n2 >> kd;                   // groups
  const size_t z = (size_t)blockIdx.x * 2 + grp;
  const bool live = z < ng;
  Fq* const ptr[3] = {t.a, t.b, t.c};
  // ---- binds: level l folds challenge bit c_l (the top remaining one) with r[l]
  if (nbind == 0) {
    if (live && gl < 3 * np) {
      const int k = gl / np, p = gl % np;
      wb[grp][k][p][0] = ld_fq(ptr[k] + (size_t)p * ng + z);
    }
  } else {
    for (int l = 0; l < nbind; l++) {
      const int nc = 1 << (nbind - 1 - l);      // combinations that remain after this level
      if (live && gl < 3 * np * nc) {
        const int k = gl / (np * nc), p = (gl / nc) % np, c = gl % nc;
        Fq lo, hi;
        if (l == 0) {
          // original index of (c_0 = 0/1, remaining bits c): c_0 len/2 + c_1 len/4 + .. + position in the bound table
          size_t idx = (size_t)p * ng + z;
          for (int b = 1; b < nbind; b++) idx += (size_t)((c >> (nbind - 1 - b)) & 1) * (len >> (b + 1));
          lo = ld_fq(ptr[k] + idx); hi = ld_fq(ptr[k] + idx + len / 2);
        } else {
          lo = wb[grp][k][p][c]; hi = wb[grp][k][p][nc + c];
        }
        wb[grp][k][p][c] = fq_add(lo, fq_mul(A.r[l], fq_sub(hi, lo)));
      }
      __syncthreads();
    }
    if (live && gl < 3 * np) {  // the bound tables back to device memory (C may be shared between instances: bound out of place, once)
      const int k = gl / np, p = gl % np;
      const Fq v = wb[grp][k][p][0];
      const size_t y = (size_t)p * ng + z;
      if (k < 2) st_fq(ptr[k] + y, v);
      else if (t.c_out) st_fq(t.c_out + y, v);
    }
  }
  if (A.dump && live && gl < 3 * np) {
    const int k = gl / np, p = gl % np;
    st_fq(A.dump + ((size_t)blockIdx.y * 3 + k) * 8 + (size_t)p * ng + z, wb[grp][k][p][0]);
  }
  if (kd == 0) { signal_done(sig); return; }
  if (gl < 3 * np) { const int k = gl / np, p = gl % np; va[grp][k][p] = live ? wb[grp][k][p][0] : fq_zero(); }
  __syncthreads();
  // ---- axis s = kd-1 .. 0: (prefix bits b_0..b_{s-1}, bit b_s, suffix y_{s+1}..) -> (prefix, y_s in 0..3, suffix)
  Fq (*vin)[3][64] = va, (*vout)[3][64] = vb;
  for (int s2 = kd - 1; s2 >= 0; s2--) {
    const int S = 1 << (2 * (kd - 1 - s2)), npre = 1 << s2, ntask = npre * S;   // per table
    if (gl < 3 * ntask) {
      const int k = gl / ntask, q = gl % ntask, pre = q / S, suf = q % S;
      const Fq u = vin[grp][k][(pre * 2) * S + suf], v = vin[grp][k][(pre * 2 + 1) * S + suf];
      const Fq d = fq_sub(v, u), e2 = fq_add(v, d), e3 = fq_add(e2, d);
      Fq* o = &vout[grp][k][(pre * 4) * S + suf];
      o[0] = u; o[S] = v; o[2 * S] = e2; o[3 * S] = e3;
    }
    __syncthreads();
    Fq (*tmp)[3][64] = vin; vin = vout; vout = tmp;
  }
  const int G = 1 << (2 * kd);
  if (gl < G) red[grp][gl] = live ? fq_mul(fq_mul(fq_mul(vin[grp][0][gl], vin[grp][1][gl]), vin[grp][2][gl]), wv) : fq_zero();
  __syncthreads();
  const size_t inst_sums = (size_t)A.ninst * A.nblk * 64;
  if (threadIdx.x < G) {
    Fq v = fq_add(red[0][threadIdx.x], red[1][threadIdx.x]);
    st_fq((A.nblk == 1 ? A.part + inst_sums + (size_t)blockIdx.y * 64 : A.part + ((size_t)blockIdx.y * A.nblk + blockIdx.x) * 64) + threadIdx.x, v);
  }
  // ---- the last workgroup of an instance adds the instance's block sums; the last instance adds the instances up
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    ticket = A.nblk == 1 ? 0u : atomicAdd(A.tickets + blockIdx.y, 1u);
  }
  __syncthreads();
  if (A.nblk > 1) {
    if (ticket != A.nblk - 1) { signal_done(sig); return; }
    __threadfence();
    const int g = threadIdx.x & 63, sl = threadIdx.x >> 6;
    Fq acc = fq_zero();
    if (g < G)
      for (size_t b = sl; b < A.nblk; b += 4) acc = fq_add(acc, ld_fq(A.part + ((size_t)blockIdx.y * A.nblk + b) * 64 + g));
    Fq* const r4 = &va[0][0][0];  // [4][64]
    r4[sl * 64 + g] = acc;
    __syncthreads();
    if (threadIdx.x < G) st_fq(A.part + inst_sums + (size_t)blockIdx.y * 64 + threadIdx.x, fq_add(fq_add(r4[threadIdx.x], r4[64 + threadIdx.x]), fq_add(r4[128 + threadIdx.x], r4[192 + threadIdx.x])));
    __syncthreads();
    if (threadIdx.x == 0) A.tickets[blockIdx.y] = 0;
  }
  if (threadIdx.x == 0) {
    __threadfence();
    ticket = atomicAdd(A.tickets + A.ninst, 1u);
  }
  __syncthreads();
  if (ticket == A.ninst - 1) {
    __threadfence();
    const int g = threadIdx.x & 63, sl = threadIdx.x >> 6;
    Fq acc = fq_zero();
    if (g < G)
      for (unsigned i = sl; i < A.ninst; i += 4) acc = fq_add(acc, ld_fq(A.part + inst_sums + (size_t)i * 64 + g));
    Fq* const r4 = &vb[0][0][0];
    r4[sl * 64 + g] = acc;
    __syncthreads();
    if (threadIdx.x < G) st_fq(A.out + threadIdx.x, fq_add(fq_add(r4[threadIdx.x], r4[64 + threadIdx.x]), fq_add(r4[128 + threadIdx.x], r4[192 + threadIdx.x])));
    if (threadIdx.x == 0) A.tickets[A.ninst] = 0;
  }
  signal_done(sig);
}

// ---- the entry point it had
int32_t sp_sumcheck_grid_batched(sp_ctx* c, sp_table* const* A, sp_table* const* B, sp_table* const* C, size_t ninst, const uint64_t* r, size_t nbind,
                                 const uint64_t* weights, size_t kd, uint64_t* out_grid, uint64_t* out_tables) {
  if (!c || !A || !B || !C || ninst == 0 || ninst > 64 || nbind > 3 || kd > 3 || !weights || (nbind && !r) || (kd && !out_grid) || (nbind == 0 && kd == 0 && !out_tables))
    return SP_EINVAL;
  HIPCHK(hipSetDevice(c->dev));
  size_t len = A[0] ? A[0]->len : 0;
  if (len < 2 || !is_pow2(len) || (len >> nbind) < ((size_t)1 << kd) || (len >> nbind) == 0) return SP_EINVAL;
  const size_t n2 = len >> nbind;
  std::vector<Triple2> T(ninst);
  std::vector<sp_table*> distinctC;
  for (size_t k = 0; k < ninst; k++) {
    if (!A[k] || !B[k] || !C[k] || A[k]->len != len || B[k]->len != len || C[k]->len != len) return SP_EINVAL;
    T[k] = Triple2{A[k]->d, B[k]->d, C[k]->d, nullptr};
    bool first = true;
    for (size_t m = 0; m < k; m++) first = first && C[m] != C[k];
    if (first) {
      distinctC.push_back(C[k]);
      if (nbind) {
        SPCHK(table_ensure_alt(C[k], n2));
        T[k].c_out = C[k]->alt;
      }
    }
  }
  GridArgs G;
  G.T = (const Triple2*)stage_small(c, 0, T.data(), sizeof(Triple2) * ninst);
  G.weights = (const Fq*)stage_small(c, sizeof(Triple2) * 64, weights, 32 * ninst);
  G.len = len; G.nbind = (int)nbind; G.kd = (int)kd;
  for (size_t l = 0; l < 3; l++) G.r[l] = l < nbind ? limbs(r + 4 * l) : fq_zero();
  const size_t ng = n2 >> kd;
  G.nblk = (unsigned)((ng + 1) / 2); G.ninst = (unsigned)ninst;
  SPCHK(ensure(&c->scratch, &c->scratch_cap, 32 * 64 * ((size_t)G.nblk + 1) * ninst + 256));
  G.part = (Fq*)c->scratch;
  if (!c->grid_tickets) {
    HIPCHK(hipMalloc((void**)&c->grid_tickets, 4 * 80));
    HIPCHK(hipMemsetAsync(c->grid_tickets, 0, 4 * 80, c->stream));
  }
  G.tickets = c->grid_tickets;
  G.out = (Fq*)hres(c);
  const bool tail = out_tables && n2 <= 8 && ninst <= TAIL_MAX_INST;
  G.dump = tail ? (Fq*)(hres(c) + TAIL_OFF) : nullptr;
  DoneSig sig = sig_make(c, (size_t)G.nblk * ninst);
  {
    const double prods = (double)ng * (double)((size_t)1 << (2 * kd)) * (double)ninst;
    ProfScope ps(c, nbind ? PF_SC_BIND_EVAL : PF_SC_EVAL, 96.0 * (double)len * (double)ninst, nullptr, 3.0 * prods + (double)(len - n2) * 3.0 * (double)ninst);
    hipLaunchKernelGGL(k_cubic_grid, dim3(G.nblk, (unsigned)ninst), dim3(256), 0, c->stream, G, sig);
  }
  SPCHK(sig_wait(c, sig));
  if (hipGetLastError() != hipSuccess) return SP_EHIP;
  if (kd) memcpy(out_grid, hres(c), 32 * ((size_t)1 << (2 * kd)));
  if (out_tables) {
    if (tail) {
      const Fq* d = (const Fq*)(hres(c) + TAIL_OFF);
      for (size_t i = 0; i < ninst; i++)
        for (int k = 0; k < 3; k++) memcpy(out_tables + 4 * ((i * 3 + k) * n2), d + (i * 3 + k) * 8, 32 * n2);
    } else {
      out_tables[0] = ~0ULL; out_tables[1] = out_tables[2] = out_tables[3] = 0;
    }
  }
  if (nbind) {
    for (size_t k = 0; k < ninst; k++) { A[k]->len = n2; B[k]->len = n2; }
    for (sp_table* t : distinctC) table_swap_to_alt(t, n2);
  }
  return SP_OK;
}
