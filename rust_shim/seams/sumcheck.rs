// src/sumcheck.rs — the provers under `--features gpu`. Every polynomial is a device table (DensePolynomial.dev, see
// seams/dense_mlpoly.rs); the round bodies are C-ABI calls; every transcript and tape operation stays here, in the
// reference's order. The executable rendering of exactly these bodies is spartan_amd/host/prover.cc (zk_sumcheck_prove) and
// spartan_amd/host/spark.inc (prove_cubic_batched_v2, prove_cubic_batched): same calls, same order, byte-identical proofs.
use super::commitments::commit_small;
use super::gpu::{self, sp_host_point, sp_table};

// ------------------------------------------------------------------------------------------------------------------
// the tables of a ZK sum-check sharded by index residue (C++ rendering: ResidueShards, spartan_amd/host/prover.cc).
// Shard g of W holds T_g[k] = T[k W + g] of every table; the top-variable pair (i, i + len/2) stays on one shard while len/2 is a
// multiple of W, so every shard runs the ordinary round kernels on its sub-tables and a round exchanges its 2..3 partial sums,
// added here in F_q. When the sub-tables are down to two entries they are bound to one and the W survivors of every table go
// back into the owner's tables; the last log2(W) rounds run unsharded. Same field values, same proof bytes.
#[cfg(feature = "gpu")]
struct ResidueShards {
  ctxs: Vec<*mut gpu::sp_ctx>,
  sub: Vec<Vec<gpu::Table>>, // [shard][table]
  active: bool,
}
#[cfg(feature = "gpu")]
impl ResidueShards {
  fn split(tabs: &[*mut sp_table]) -> Self {
    let ctxs = gpu::shard_ctxs();
    let (w, len) = (ctxs.len(), unsafe { gpu::sp_table_len(tabs[0]) });
    if w < 2 || len < 4 * w { return ResidueShards { ctxs: Vec::new(), sub: Vec::new(), active: false }; }
    gpu::ok(unsafe { gpu::sp_ctx_sync(gpu::ctx()) }); // the tables as produced by everything queued on the owning context
    let sub = (0..w)
      .map(|g| tabs.iter().map(|&t| { let mut o = std::ptr::null_mut(); gpu::ok(unsafe { gpu::sp_table_residue_split(ctxs[g], t, w, g, &mut o) }); gpu::Table(o) }).collect())
      .collect();
    ResidueShards { ctxs, sub, active: true }
  }
  fn sub_len(&self) -> usize { unsafe { gpu::sp_table_len(self.sub[0][0].0) } }
  fn handles(&self, g: usize) -> Vec<*mut sp_table> { self.sub[g].iter().map(|t| t.0).collect() }
  fn add_partials(parts: &[[Scalar; 3]], n: usize, ev: &mut [Scalar]) { for k in 0..n { ev[k] = parts.iter().map(|p| p[k]).sum(); } }
  fn eval(&self, kind: i32, ev: &mut [Scalar]) {
    let mut parts = vec![[Scalar::zero(); 3]; self.ctxs.len()];
    for g in 0..self.ctxs.len() {
      let h = self.handles(g);
      gpu::ok(unsafe { gpu::sp_sumcheck_eval(self.ctxs[g], kind, h.as_ptr(), h.len(), gpu::limbs_mut(&mut parts[g])) });
    }
    Self::add_partials(&parts, if kind == 0 { 2 } else { 3 }, ev);
  }
  fn bind_eval_start(&self, kind: i32, r: &Scalar) { // every shard's bind + next evaluation in flight together (own streams)
    for g in 0..self.ctxs.len() {
      let h = self.handles(g);
      gpu::ok(unsafe { gpu::sp_sumcheck_bind_eval_start(self.ctxs[g], kind, h.as_ptr(), h.len(), gpu::limbs1(r)) });
    }
  }
  fn bind_eval_collect(&self, kind: i32, ev: &mut [Scalar]) {
    let mut parts = vec![[Scalar::zero(); 3]; self.ctxs.len()];
    for g in 0..self.ctxs.len() { gpu::ok(unsafe { gpu::sp_sumcheck_bind_eval_collect(self.ctxs[g], gpu::limbs_mut(&mut parts[g])) }); }
    Self::add_partials(&parts, if kind == 0 { 2 } else { 3 }, ev);
  }
  /// sub-tables of two entries: bind them to one and hand the W survivors of every table back to the owner's tables
  fn bind_last_and_gather(&mut self, r: &Scalar, tabs: &[*mut sp_table]) {
    let w = self.ctxs.len();
    let heads: Vec<Vec<Scalar>> = (0..w)
      .map(|g| {
        let h = self.handles(g);
        let mut v = vec![Scalar::zero(); h.len()];
        gpu::ok(unsafe { gpu::sp_table_bind_top_heads(self.ctxs[g], h.as_ptr(), h.len(), gpu::limbs1(r), gpu::limbs_mut(&mut v)) });
        v
      })
      .collect();
    for (t, &tab) in tabs.iter().enumerate() {
      let v: Vec<Scalar> = (0..w).map(|g| heads[g][t]).collect();
      gpu::ok(unsafe { gpu::sp_table_write(gpu::ctx(), tab, 0, gpu::limbs(&v), w) });
      gpu::ok(unsafe { gpu::sp_table_set_len(tab, w) });
    }
    self.sub.clear();
    self.active = false;
  }
}

// ------------------------------------------------------------------------------------------------------------------
// ZKSumcheckInstanceProof::prove_quad (:428-586, kind 0: A*B over gens_3) and ::prove_cubic_with_additive_term (:588-776,
// kind 2: A*(B*C - D) over gens_4) share everything but `kind`.
// Round j: the device binds the tables at r_j and evaluates round j+1 in the same pass (sp_sumcheck_bind_eval_start ..
// _collect) while this core commits: comm_eval, the DotProductProof's delta, Cy, beta and the next comm_poly are 2..5-term
// commitments under (gens_n.G.., gens_n.h, gens_1.G, gens_1.h) — ONE index list serves them all. Everything in them that
// depends on the tape alone (delta entirely; the blind terms of comm_eval, beta, comm_poly) is computed AHEAD of the rounds by
// the library's helper thread (sp_host_zk_ahead_*): inside the round loop nothing but DotProductProof::prove draws from the
// tape (d_vec, r_delta, r_beta: nizk/mod.rs:330-334), so the rounds' draws are taken up front, in the reference's order.
#[cfg(feature = "gpu")]
impl ZKSumcheckInstanceProof {
  pub fn prove_zk_gpu(
    kind: i32, // 0: prove_quad, 2: prove_cubic_with_additive_term
    claim: &Scalar,
    blind_claim: &Scalar,
    num_rounds: usize,
    polys: &mut [&mut DensePolynomial], // (A, B) or (A, B, C, D), device-resident
    gens_1: &MultiCommitGens,
    gens_n: &MultiCommitGens,
    transcript: &mut Transcript,
    random_tape: &mut RandomTape,
  ) -> (Self, Vec<Scalar>, Vec<Scalar>, Scalar) {
    let blinds_poly = random_tape.random_vector(b"blinds_poly", num_rounds);
    let blinds_evals = random_tape.random_vector(b"blinds_evals", num_rounds);
    let tabs: Vec<*mut sp_table> = polys.iter().map(|p| p.dev.as_ref().expect("device-resident polynomial").0).collect();
    let (gn, g1) = (&gens_n.dev, &gens_1.dev);
    assert!(gn.g == g1.g); // both are prefixes of the "gens_r1cs_sat" stream
    let nn = gens_n.n;
    let mut idx_u: Vec<u32> = gn.G.clone();
    idx_u.extend_from_slice(&[gn.h, g1.G[0], g1.h]);
    let W = idx_u.len();
    let on_host = gpu::small_msm_on_host();

    // the rounds' tape draws, up front (same stream of draws as the reference's loop: see the header)
    let mut d_all: Vec<Scalar> = Vec::new();
    let (mut r_delta_all, mut r_beta_all) = (Vec::new(), Vec::new());
    let ahead = if on_host {
      for _ in 0..num_rounds {
        d_all.extend(random_tape.random_vector(b"d_vec", nn));
        r_delta_all.push(random_tape.random_scalar(b"r_delta"));
        r_beta_all.push(random_tape.random_scalar(b"r_beta"));
      }
      let mut h = std::ptr::null_mut();
      gpu::ok(unsafe {
        gpu::sp_host_zk_ahead_begin(gn.g, idx_u.as_ptr(), W, nn, num_rounds, gpu::limbs(&blinds_poly), gpu::limbs(&blinds_evals), gpu::limbs(&d_all),
                                    gpu::limbs(&r_delta_all), gpu::limbs(&r_beta_all), &mut h)
      });
      Some(gpu::ZkAhead::new(h, num_rounds)) // sp_host_zk_ahead_wait per round, sp_host_zk_ahead_free on drop
    } else {
      None
    };
    // rows of scalars over idx_u -> encoded commitments (+ a point computed ahead, host mode)
    let commit_rows = |rows: &[Scalar], nrows: usize, addend: Option<&[*const sp_host_point]>| commit_small(gn.g, &idx_u, rows, nrows, addend);
    let make_poly = |ev: &[Scalar], cl: &Scalar| {
      if kind == 0 { UniPoly::from_evals(&[ev[0], cl - ev[0], ev[1]]) } else { UniPoly::from_evals(&[ev[0], cl - ev[0], ev[1], ev[2]]) }
    };

    let mut ev = vec![Scalar::zero(); 3];
    let mut rs = if on_host { ResidueShards::split(&tabs) } else { ResidueShards { ctxs: Vec::new(), sub: Vec::new(), active: false } };
    if rs.active { rs.eval(kind, &mut ev); } else { gpu::ok(unsafe { gpu::sp_sumcheck_eval(gpu::ctx(), kind, tabs.as_ptr(), tabs.len(), gpu::limbs_mut(&mut ev)) }); }
    let mut claim_per_round = *claim;
    let mut poly = make_poly(&ev, &claim_per_round);
    assert_eq!(poly.as_vec().len(), nn);
    // comm_claim_per_round (:448 / :611) and the first comm_poly (:473 / :661) in one call
    let (mut comm_claim_per_round, mut comm_poly) = {
      let mut rows = vec![Scalar::zero(); 2 * W];
      rows[nn + 1] = claim_per_round;
      rows[nn + 2] = *blind_claim;
      rows[W..W + nn].copy_from_slice(&poly.as_vec());
      if !on_host { rows[W + nn] = blinds_poly[0]; }
      let add = ahead.as_ref().map(|a| [std::ptr::null(), a.wait(0).bp_hn()]);
      let cm = commit_rows(&rows, 2, add.as_ref().map(|a| &a[..]));
      (cm[0], cm[1])
    };

    let (mut r, mut comm_polys, mut comm_evals, mut proofs) = (Vec::new(), Vec::new(), Vec::new(), Vec::new());
    for j in 0..num_rounds {
      comm_poly.append_to_transcript(b"comm_poly", transcript);
      comm_polys.push(comm_poly);
      let r_j = transcript.challenge_scalar(b"challenge_nextround");
      let more = j + 1 < num_rounds;
      let eval = poly.evaluate(&r_j);
      let (d, r_delta, r_beta) = if on_host {
        (d_all[j * nn..(j + 1) * nn].to_vec(), r_delta_all[j], r_beta_all[j])
      } else {
        (random_tape.random_vector(b"d_vec", nn), random_tape.random_scalar(b"r_delta"), random_tape.random_scalar(b"r_beta"))
      };
      // bind every table at r_j (:485-486 / :673-676), fused with the next round's evaluations (:460-469 / :624-652)
      let len = unsafe { gpu::sp_table_len(tabs[0]) };
      let (comm_eval, delta, mut pending);
      let mut resharded = false; // the shards have just handed their last entries back: the next evaluation is a call of its own
      if on_host {
        if rs.active && rs.sub_len() >= 4 {
          rs.bind_eval_start(kind, &r_j);
          pending = true;
        } else if rs.active {
          rs.bind_last_and_gather(&r_j, &tabs);
          resharded = true;
          pending = false;
        } else if len >= 4 {
          gpu::ok(unsafe { gpu::sp_sumcheck_bind_eval_start(gpu::ctx(), kind, tabs.as_ptr(), tabs.len(), gpu::limbs1(&r_j)) });
          pending = true;
        } else {
          gpu::ok(unsafe { gpu::sp_table_bind_top(gpu::ctx(), tabs.as_ptr(), tabs.len(), gpu::limbs1(&r_j)) });
          pending = false;
        }
        let aj = ahead.as_ref().unwrap().wait(j);
        let mut row = vec![Scalar::zero(); W];
        row[nn + 1] = eval; // comm_eval = eval * G1 + blinds_evals[j] * h: the blind term comes from `ahead`
        comm_eval = commit_rows(&row, 1, Some(&[aj.be_h()]))[0];
        delta = aj.delta(); // commit(d_j, r_delta_j) under gens_n: complete
      } else {
        pending = false;
        let mut rows1 = vec![Scalar::zero(); 2 * W];
        rows1[nn + 1] = eval;
        rows1[nn + 2] = blinds_evals[j];
        rows1[W..W + nn].copy_from_slice(&d);
        rows1[W + nn] = r_delta;
        if len >= 4 {
          // all-device variant: the round's r-dependent commitments next to the bind, one completion wait
          let mut pts = [0u8; 64];
          gpu::ok(unsafe {
            gpu::sp_sumcheck_bind_eval_commit(gpu::ctx(), kind, tabs.as_ptr(), tabs.len(), gpu::limbs1(&r_j), gpu::limbs_mut(&mut ev), gn.g, idx_u.as_ptr(), W,
                                              gpu::limbs(&rows1), 2, pts.as_mut_ptr())
          });
          comm_eval = CompressedGroup::from_slice(&pts[..32]);
          delta = CompressedGroup::from_slice(&pts[32..]);
        } else {
          gpu::ok(unsafe { gpu::sp_table_bind_top(gpu::ctx(), tabs.as_ptr(), tabs.len(), gpu::limbs1(&r_j)) });
          let cm = commit_rows(&rows1, 2, None);
          comm_eval = cm[0];
          delta = cm[1];
        }
      }
      comm_claim_per_round.append_to_transcript(b"comm_claim_per_round", transcript);
      comm_eval.append_to_transcript(b"comm_eval", transcript);
      let w = transcript.challenge_vector(b"combine_two_claims_to_one", 2);
      let target = w[0] * claim_per_round + w[1] * eval;
      let blind_sc = if j == 0 { *blind_claim } else { blinds_evals[j - 1] };
      let blind = w[0] * blind_sc + w[1] * blinds_evals[j];
      // a = w[0] * a_sc + w[1] * a_eval (:509-533 / :699-723)
      let mut a = Vec::with_capacity(nn);
      let mut pw = Scalar::one();
      for i in 0..nn {
        let a_sc = if i == 0 { Scalar::one() + Scalar::one() } else { Scalar::one() };
        a.push(w[0] * a_sc + w[1] * pw);
        pw *= r_j;
      }
      // DotProductProof::prove (nizk/mod.rs:311-370): x = poly.coeffs, blind_x = blinds_poly[j], y = target, blind_y = blind
      transcript.append_protocol_name(b"dot product proof");
      comm_poly.append_to_transcript(b"Cx", transcript); // Cx = commit(x, blind_x): same inputs and generators as comm_poly
      let dp: Scalar = (0..nn).map(|i| a[i] * d[i]).sum();
      // Cy = target * G1 + blind * h ; beta = dp * G1 + r_beta * h ; and the next round's comm_poly
      let coeffs = poly.as_vec();
      let (Cy, beta, next);
      if on_host {
        let mut rows2 = vec![Scalar::zero(); 2 * W];
        rows2[nn + 1] = target;
        rows2[nn + 2] = blind;
        rows2[W + nn + 1] = dp; // + r_beta * h from `ahead`
        let cm2 = commit_rows(&rows2, 2, Some(&[std::ptr::null(), ahead.as_ref().unwrap().wait(j).rb_h()]));
        Cy = cm2[0];
        beta = cm2[1];
        if pending && rs.active { rs.bind_eval_collect(kind, &mut ev); }
        else if pending { gpu::ok(unsafe { gpu::sp_sumcheck_bind_eval_collect(gpu::ctx(), gpu::limbs_mut(&mut ev)) }); }
        if resharded && more {
          gpu::ok(unsafe { gpu::sp_sumcheck_eval(gpu::ctx(), kind, tabs.as_ptr(), tabs.len(), gpu::limbs_mut(&mut ev)) });
          pending = true;
        }
        next = if more {
          assert!(pending);
          let np = make_poly(&ev, &eval);
          let mut row3 = vec![Scalar::zero(); W];
          row3[..nn].copy_from_slice(&np.as_vec());
          let cp = commit_rows(&row3, 1, Some(&[ahead.as_ref().unwrap().wait(j + 1).bp_hn()]))[0];
          Some((np, cp))
        } else { None };
      } else {
        let nrows2 = if more { 3 } else { 2 };
        let mut rows2 = vec![Scalar::zero(); nrows2 * W];
        rows2[nn + 1] = target;
        rows2[nn + 2] = blind;
        rows2[W + nn + 1] = dp;
        rows2[W + nn + 2] = r_beta;
        let np = if more { Some(make_poly(&ev, &eval)) } else { None };
        if let Some(p) = &np {
          rows2[2 * W..2 * W + nn].copy_from_slice(&p.as_vec());
          rows2[2 * W + nn] = blinds_poly[j + 1];
        }
        let cm2 = commit_rows(&rows2, nrows2, None);
        Cy = cm2[0];
        beta = cm2[1];
        next = np.map(|p| (p, cm2[2]));
      }
      Cy.append_to_transcript(b"Cy", transcript);
      a.append_to_transcript(b"a", transcript);
      delta.append_to_transcript(b"delta", transcript);
      beta.append_to_transcript(b"beta", transcript);
      let c = transcript.challenge_scalar(b"c");
      let z = (0..nn).map(|i| c * coeffs[i] + d[i]).collect::<Vec<Scalar>>();
      proofs.push(DotProductProof { delta, beta, z, z_delta: c * blinds_poly[j] + r_delta, z_beta: c * blind + r_beta });
      claim_per_round = eval;
      comm_claim_per_round = comm_eval;
      comm_evals.push(comm_eval);
      r.push(r_j);
      if let Some((np, cp)) = next { poly = np; comm_poly = cp; }
    }
    let mut final_claims = vec![Scalar::zero(); tabs.len()];
    gpu::ok(unsafe { gpu::sp_table_heads(gpu::ctx(), tabs.as_ptr(), tabs.len(), gpu::limbs_mut(&mut final_claims)) });
    for p in polys.iter_mut() { p.num_vars -= num_rounds; p.len >>= num_rounds; }
    (ZKSumcheckInstanceProof::new(comm_polys, comm_evals, proofs), r, final_claims, blinds_evals[num_rounds - 1])
  }
}

// ------------------------------------------------------------------------------------------------------------------
// SURVEY 8e for prove_cubic_batched (C++ rendering: CubicShards, spartan_amd/host/spark.inc): every table of the batch split by index
// residue over the W shards, the throughput-sized rounds run per shard with the per-instance partial evaluations added here, then
// every shard packs its sub-tables (sp_tables_pack), the buffers are concatenated in shard order and the owner scatters them back into the
// full tables (sp_tables_unpack_residues); the latency-sized rounds continue unsharded. LOGIC a maintainer must review (not a mechanical
// call sequence): the hand-over length, the instance -> sub-table maps, the order of the gathered buffer.
#[cfg(feature = "gpu")]
struct CubicShards {
  ctxs: Vec<*mut gpu::sp_ctx>,
  full: Vec<*mut sp_table>,                 // every table once: A_i, B_i interleaved, then the distinct C tables
  sub: Vec<Vec<gpu::Table>>,                // [shard][table of `full`]
  ia: Vec<usize>, ib: Vec<usize>, ic: Vec<usize>, // per instance: position of its A, B, C in `full`
  active: bool,
}
#[cfg(feature = "gpu")]
impl CubicShards {
  fn min_len() -> usize { (2 * gpu::double_round_max_len()).max(64) }
  fn split(all: &[*mut sp_table], A: &[*mut sp_table], B: &[*mut sp_table], C: &[*mut sp_table]) -> Self {
    let none = CubicShards { ctxs: Vec::new(), full: Vec::new(), sub: Vec::new(), ia: Vec::new(), ib: Vec::new(), ic: Vec::new(), active: false };
    let ctxs = gpu::shard_ctxs();
    let (w, len) = (ctxs.len(), unsafe { gpu::sp_table_len(all[0]) });
    if w < 2 || (w & (w - 1)) != 0 || len < 4 * Self::min_len() || Self::min_len() / w < 4 { return none; }
    gpu::ok(unsafe { gpu::sp_ctx_sync(gpu::ctx()) });
    let pos = |t: *mut sp_table| all.iter().position(|&x| x == t).expect("table of the batch");
    let sub = (0..w)
      .map(|g| all.iter().map(|&t| { let mut o = std::ptr::null_mut(); gpu::ok(unsafe { gpu::sp_table_residue_split(ctxs[g], t, w, g, &mut o) }); gpu::Table(o) }).collect())
      .collect();
    CubicShards { ctxs, full: all.to_vec(), sub, ia: A.iter().map(|&t| pos(t)).collect(), ib: B.iter().map(|&t| pos(t)).collect(), ic: C.iter().map(|&t| pos(t)).collect(), active: true }
  }
  fn sub_len(&self) -> usize { unsafe { gpu::sp_table_len(self.sub[0][0].0) } }
  fn keep_going(&self) -> bool { self.active && self.sub_len() * self.ctxs.len() / 2 >= Self::min_len() }
  fn lists(&self, g: usize) -> (Vec<*mut sp_table>, Vec<*mut sp_table>, Vec<*mut sp_table>) {
    let h = |ix: &Vec<usize>| ix.iter().map(|&k| self.sub[g][k].0).collect::<Vec<_>>();
    (h(&self.ia), h(&self.ib), h(&self.ic))
  }
  fn sum(parts: &[Vec<Scalar>], ev: &mut [Scalar]) { for k in 0..ev.len() { ev[k] = parts.iter().map(|p| p[k]).sum(); } }
  fn eval(&self, ev: &mut [Scalar]) {
    let ni = self.ia.len();
    let mut parts = vec![vec![Scalar::zero(); 3 * ni]; self.ctxs.len()];
    for g in 0..self.ctxs.len() {
      let (a, b, c) = self.lists(g);
      gpu::ok(unsafe { gpu::sp_sumcheck_eval_batched(self.ctxs[g], a.as_ptr(), b.as_ptr(), c.as_ptr(), ni, gpu::limbs_mut(&mut parts[g])) });
    }
    Self::sum(&parts, ev);
  }
  fn bind_eval(&self, r: &Scalar, ev: &mut [Scalar]) {
    let ni = self.ia.len();
    let mut parts = vec![vec![Scalar::zero(); 3 * ni]; self.ctxs.len()];
    for g in 0..self.ctxs.len() {
      let (a, b, c) = self.lists(g);
      gpu::ok(unsafe { gpu::sp_sumcheck_bind_eval_batched(self.ctxs[g], a.as_ptr(), b.as_ptr(), c.as_ptr(), ni, gpu::limbs1(r), gpu::limbs_mut(&mut parts[g])) });
    }
    Self::sum(&parts, ev);
  }
  /// the sub-tables go back into the owner's full tables (current length sub_len * W): in[((g * ntabs + t) * sub + k)]
  fn hand_back(&mut self) {
    let (w, sl, nt) = (self.ctxs.len(), self.sub_len(), self.full.len());
    let mut all = vec![Scalar::zero(); w * nt * sl];
    for g in 0..w {
      let h: Vec<*mut sp_table> = self.sub[g].iter().map(|t| t.0).collect();
      gpu::ok(unsafe { gpu::sp_tables_pack(self.ctxs[g], h.as_ptr(), nt, sl, gpu::limbs_mut(&mut all[g * nt * sl..(g + 1) * nt * sl])) });
    }
    gpu::ok(unsafe { gpu::sp_tables_unpack_residues(gpu::ctx(), self.full.as_ptr(), nt, w, sl, gpu::limbs(&all)) });
    self.sub.clear();
    self.active = false;
  }
}

// SumcheckInstanceProof::prove_cubic_batched (:254-424). comb_func is the cubic product on this path
// (product_tree.rs:316-318). Long tables: one round per call, the bind at r_j fused with round j+1's evaluations
// (sp_sumcheck_eval_batched, then sp_sumcheck_bind_eval_batched). Short tables (<= 512 entries: ~290 of the 361 rounds of a 2^20
// proof, each a latency-bound trip): TWO rounds per call — the evaluations of the round after a bind are a cubic in that bind's
// challenge,  E(t; r) = (1-r)^3 M0 + (1-r)^2 r M1 + (1-r) r^2 M2 + r^3 M3,  M1 = (T1-T2)/2 - M3,  M2 = (T1+T2)/2 - M0,
// and the device returns (M0, M3, T1, T2) for t = 0, 2, 3 next to a round's evaluations (sp_sumcheck_eval_coeffs_batched /
// sp_sumcheck_bind2_eval_batched), already weighted by `coeffs` and summed over the instances. Once the tables have at most 8
// entries the same call hands them over (.._tables_batched) and the last <= 3 rounds run here with the reference's own loop
// body. Exact field arithmetic throughout: same values, same transcript, same bytes as the one-round form.
#[cfg(feature = "gpu")]
fn evals_from_coeffs(S: &[Scalar], r: &Scalar) -> [Scalar; 3] {
  let half = (2_usize).to_scalar().invert().unwrap();
  let om = Scalar::one() - r;
  let (w0, w1, w2, w3) = (om * om * om, om * om * r, om * r * r, r * r * r);
  let mut ev = [Scalar::zero(); 3];
  for k in 0..3 {
    let (M0, M3, T1, T2) = (S[4 * k], S[4 * k + 1], S[4 * k + 2], S[4 * k + 3]);
    let (M1, M2) = ((T1 - T2) * half - M3, (T1 + T2) * half - M0);
    ev[k] = M0 * w0 + M1 * w1 + M2 * w2 + M3 * w3;
  }
  ev
}
/// The eq table as a FACTOR (spartan_hip.h, sp_sumcheck_eval_batched_eq; spark.inc EqFactor is the C++ twin): the throughput-sized rounds of a
/// batch whose product-circuit instances share poly_C_par = EqPolynomial::new(rho).evals() (product_tree.rs:279). With the tables bound at
/// r_0..r_{j-1}, round j's combined evaluations are E(t) = kappa_j(t) Q(t) + D(t) with Q(t) = sum_i coeff_i q_i(t) QUADRATIC (the device
/// returns q_i(0), q_i(2) summed against the ORIGINAL eq table's leading entries, which it never binds), D(t) the generic instances' cubic
/// (four evaluations each), kappa_j(t) = K_j eq(t, rho_j) / (1 - rho_j), K_j = prod_{k<j} eq(r_k, rho_k) / (1 - rho_k). Q(1) follows from
/// the round's claim, Q(3) = Q(0) - 3 Q(1) + 3 Q(2). Exact identities in the field: the round polynomials are the reference's.
/// THIS BODY CARRIES LOGIC (not a mechanical call sequence): review it against spark.inc and tests/test_host_arith.py.
#[cfg(feature = "gpu")]
struct EqFactor { on: bool, rho: Vec<Scalar>, inv1m: Vec<Scalar>, inv_rho: Vec<Scalar>, K: Scalar, Kinv: Scalar }
#[cfg(feature = "gpu")]
impl EqFactor {
  const MIN_LEN: usize = 65536; // the factored kernels are the throughput forms only
  fn off() -> Self { EqFactor { on: false, rho: vec![], inv1m: vec![], inv_rho: vec![], K: Scalar::one(), Kinv: Scalar::one() } }
  /// usable when no coordinate of the point is 0 or 1 (1 / (1 - rho) and 1 / rho must exist)
  fn begin(point: Option<&[Scalar]>, num_rounds: usize, len: usize, np: usize, ni: usize) -> Self {
    let rho = match point { Some(p) if gpu::eq_factor_enabled() && p.len() == num_rounds && np > 0 && ni <= 24 && len >= Self::MIN_LEN => p.to_vec(), _ => return Self::off() };
    let one = Scalar::one();
    if rho.iter().any(|r| *r == Scalar::zero() || *r == one) { return Self::off(); }
    let inv1m = rho.iter().map(|r| (one - r).invert().unwrap()).collect();
    let inv_rho = rho.iter().map(|r| r.invert().unwrap()).collect();
    EqFactor { on: true, rho, inv1m, inv_rho, K: one, Kinv: one }
  }
  /// E(0), E(2), E(3) of round j from the device's 4 scalars per instance, the coefficients and the round's claim e
  fn combine(&self, j: usize, ev4: &[Scalar], coeffs: &[Scalar], np: usize, ni: usize, e: &Scalar) -> [Scalar; 3] {
    let (mut q0, mut q2) = (Scalar::zero(), Scalar::zero());
    let mut d = [Scalar::zero(); 4];
    for i in 0..np { q0 += ev4[4 * i] * coeffs[i]; q2 += ev4[4 * i + 1] * coeffs[i]; }
    for i in np..ni { for k in 0..4 { d[k] += ev4[4 * i + k] * coeffs[i]; } }
    if self.K == Scalar::zero() { return [d[0], d[2], d[3]]; } // a challenge hit a root of eq(., rho_k): the eq table is zero from there on
    let (one, two, three, five) = (Scalar::one(), (2_usize).to_scalar(), (3_usize).to_scalar(), (5_usize).to_scalar());
    let r = self.rho[j];
    let ks = self.K * self.inv1m[j];
    let (k2, k3) = (ks * (three * r - one), ks * (five * r - two));
    let e0 = self.K * q0 + d[0];
    let e1 = e - e0;
    let q1 = (e1 - d[1]) * self.Kinv * self.inv_rho[j] * (one - r); // / kappa(1), kappa(1) = K rho / (1 - rho)
    let q3 = q0 - three * q1 + three * q2;
    [e0, k2 * q2 + d[2], k3 * q3 + d[3]]
  }
  /// the tables have been bound at r (round j's challenge)
  fn bound(&mut self, j: usize, r: &Scalar) {
    let one = Scalar::one();
    let f = (one - r) * (one - self.rho[j]) + r * self.rho[j]; // eq(r, rho_j)
    self.K = self.K * f * self.inv1m[j];
    self.Kinv = if f == Scalar::zero() { Scalar::zero() } else { self.Kinv * f.invert().unwrap() * (one - self.rho[j]) };
  }
}
/// The last rounds on tables of m <= 8 entries, tab = [instance][A, B, C][m]: the reference's loop body (:290-393) verbatim in
/// structure; `message` appends the round's polynomial and returns its challenge.
#[cfg(feature = "gpu")]
fn cubic_tail_rounds(tab: &mut [Scalar], ni: usize, m_in: usize, coeffs: &[Scalar], message: &mut dyn FnMut([Scalar; 3]) -> Scalar) {
  let mut m = m_in;
  while m >= 2 {
    let h = m / 2;
    let mut evc = [Scalar::zero(); 3];
    for i in 0..ni {
      let at = |k: usize, z: usize| tab[(i * 3 + k) * m_in + z];
      let (mut e0, mut e2, mut e3) = (Scalar::zero(), Scalar::zero(), Scalar::zero());
      for z in 0..h {
        let (a0, a1, b0, b1, c0, c1) = (at(0, z), at(0, h + z), at(1, z), at(1, h + z), at(2, z), at(2, h + z));
        let (a2, b2, c2) = (a1 + a1 - a0, b1 + b1 - b0, c1 + c1 - c0);
        let (a3, b3, c3) = (a2 + a1 - a0, b2 + b1 - b0, c2 + c1 - c0);
        e0 += a0 * b0 * c0; e2 += a2 * b2 * c2; e3 += a3 * b3 * c3;
      }
      evc[0] += e0 * coeffs[i]; evc[1] += e2 * coeffs[i]; evc[2] += e3 * coeffs[i];
    }
    let r = message(evc);
    for i in 0..ni { for k in 0..3 { for z in 0..h {
      let (lo, hi) = (tab[(i * 3 + k) * m_in + z], tab[(i * 3 + k) * m_in + h + z]);
      tab[(i * 3 + k) * m_in + z] = lo + r * (hi - lo);
    } } }
    m = h;
  }
}

#[cfg(feature = "gpu")]
impl SumcheckInstanceProof {
  pub fn prove_cubic_batched_gpu(
    claim: &Scalar,
    num_rounds: usize,
    poly_vec_par: (&mut Vec<&mut DensePolynomial>, &mut Vec<&mut DensePolynomial>, &mut DensePolynomial),
    poly_vec_seq: (&mut Vec<&mut DensePolynomial>, &mut Vec<&mut DensePolynomial>, &mut Vec<&mut DensePolynomial>),
    coeffs: &[Scalar],
    transcript: &mut Transcript,
    eq_point: Option<&[Scalar]>, // Some(rand) when poly_C_par = EqPolynomial::new(rand).evals() (product_tree.rs:279): the factored rounds (EqFactor)
  ) -> (Self, Vec<Scalar>, (Vec<Scalar>, Vec<Scalar>, Scalar), (Vec<Scalar>, Vec<Scalar>, Vec<Scalar>)) {
    let (poly_A_vec_par, poly_B_vec_par, poly_C_par) = poly_vec_par;
    let (poly_A_vec_seq, poly_B_vec_seq, poly_C_vec_seq) = poly_vec_seq;
    let (np, ns) = (poly_A_vec_par.len(), poly_A_vec_seq.len());
    let ni = np + ns;
    let dev = |p: &DensePolynomial| -> *mut sp_table { p.dev.as_ref().expect("device-resident polynomial").0 };
    // instance k = (A_k, B_k, C_k); the `par` instances share poly_C_par (the library binds a shared table once)
    let mut A: Vec<*mut sp_table> = poly_A_vec_par.iter().map(|p| dev(p)).collect();
    let mut B: Vec<*mut sp_table> = poly_B_vec_par.iter().map(|p| dev(p)).collect();
    let mut C: Vec<*mut sp_table> = vec![dev(poly_C_par); np];
    A.extend(poly_A_vec_seq.iter().map(|p| dev(p)));
    B.extend(poly_B_vec_seq.iter().map(|p| dev(p)));
    C.extend(poly_C_vec_seq.iter().map(|p| dev(p)));
    // every table once: A_i, B_i interleaved, then the distinct C tables (the order of the final claims)
    let mut all: Vec<*mut sp_table> = Vec::with_capacity(2 * ni + 1 + ns);
    for i in 0..ni { all.push(A[i]); all.push(B[i]); }
    all.push(dev(poly_C_par));
    all.extend(poly_C_vec_seq.iter().map(|p| dev(p)));
    let (ap, bp, cp, c) = (A.as_ptr(), B.as_ptr(), C.as_ptr(), gpu::ctx());
    let null = std::ptr::null_mut::<u64>();

    let mut e = *claim;
    let mut r: Vec<Scalar> = Vec::new();
    let mut cubic_polys: Vec<CompressedUniPoly> = Vec::new();
    let mut ev = vec![Scalar::zero(); 3 * ni];
    let mut ev4 = vec![Scalar::zero(); 4 * ni]; // the factored form's 4 scalars per instance
    let mut eqf = EqFactor::off(); // per-instance evaluations (one round per launch)
    let mut evc = [Scalar::zero(); 3];         // combined with coeffs (:359-369)
    let mut S = vec![Scalar::zero(); 12];      // the cubic that gives the next round's evaluations
    let mut heads = vec![Scalar::zero(); all.len()];
    let mut tail = vec![Scalar::zero(); 3 * 8 * ni];
    let (mut have_heads, mut have_S) = (false, false);
    let tail_ok = gpu::opt("sumcheck.host_tail") != 0 && np >= 1 && ni <= 21;
    let dmax = gpu::double_round_max_len(); // 4096 (option sumcheck.double_round_max_len)
    let len_of = |t: *mut sp_table| unsafe { gpu::sp_table_len(t) };
    // :370-381 and :395-396: the round's cubic, its transcript message, the challenge
    let mut round_message = |evc: &[Scalar; 3], e: &mut Scalar, r: &mut Vec<Scalar>, polys: &mut Vec<CompressedUniPoly>, transcript: &mut Transcript| {
      let poly = UniPoly::from_evals(&[evc[0], *e - evc[0], evc[1], evc[2]]);
      poly.append_to_transcript(b"poly", transcript);
      let r_j = transcript.challenge_scalar(b"challenge_nextround");
      r.push(r_j);
      *e = poly.evaluate(&r_j);
      polys.push(poly.compress());
      r_j
    };
    let combine = |ev: &[Scalar]| -> [Scalar; 3] {
      let mut o = [Scalar::zero(); 3];
      for i in 0..ni { o[0] += ev[3 * i] * coeffs[i]; o[1] += ev[3 * i + 1] * coeffs[i]; o[2] += ev[3 * i + 2] * coeffs[i]; }
      o
    };
    let tail_given = |tail: &[Scalar]| gpu::limbs_of(&tail[0]) != [!0u64, 0, 0, 0];
    let mark = |tail: &mut [Scalar]| gpu::set_limbs(&mut tail[0], [!0u64, 0, 0, 0]);

    let mut j = 0usize;
    // the last rounds on this core: (m entries per table in `tail`) -> final claims in `heads`
    macro_rules! finish_on_host { ($m:expr) => {{
      let m = $m;
      cubic_tail_rounds(&mut tail, ni, m, coeffs, &mut |evs| { j += 1; round_message(&evs, &mut e, &mut r, &mut cubic_polys, transcript) });
      for i in 0..ni { heads[2 * i] = tail[(i * 3) * m]; heads[2 * i + 1] = tail[(i * 3 + 1) * m]; }
      heads[2 * ni] = tail[2 * m]; // poly_C_par is instance 0's C
      for k in 0..ns { heads[2 * ni + 1 + k] = tail[((np + k) * 3 + 2) * m]; }
      have_heads = true;
    }} }

    // ---- TWO rounds per trip on short tables (the default) ---------------------------------------------------------------
    // the throughput-sized rounds on W residue classes of every table (CubicShards above), until the hand-over length
    let mut cs = if num_rounds > 0 { CubicShards::split(&all, &A, &B, &C) } else { CubicShards::split(&all[..0], &A[..0], &B[..0], &C[..0]) };
    if cs.active {
      cs.eval(&mut ev);
      evc = combine(&ev);
      while cs.active {
        let r_j = round_message(&evc, &mut e, &mut r, &mut cubic_polys, transcript);
        j += 1;
        let last = !cs.keep_going();
        cs.bind_eval(&r_j, &mut ev);
        evc = combine(&ev);
        if last { cs.hand_back(); }
      }
    } else if num_rounds > 0 {
      let len0 = len_of(A[0]);
      if tail_ok && len0 >= 2 && len0 <= 8 {
        mark(&mut tail);
        gpu::ok(unsafe {
          gpu::sp_sumcheck_bind2_eval_tables_batched(c, ap, bp, cp, ni, std::ptr::null(), std::ptr::null(), gpu::limbs(coeffs), gpu::limbs_mut(&mut evc),
                                                     if len0 >= 4 { gpu::limbs_mut(&mut S) } else { null }, null, gpu::limbs_mut(&mut tail))
        });
        assert!(tail_given(&tail));
        finish_on_host!(len0);
      } else if len0 >= 4 && len0 <= dmax {
        gpu::ok(unsafe { gpu::sp_sumcheck_eval_coeffs_batched(c, ap, bp, cp, ni, gpu::limbs(coeffs), gpu::limbs_mut(&mut evc), gpu::limbs_mut(&mut S)) });
        have_S = true;
      } else {
        eqf = EqFactor::begin(eq_point, num_rounds, len0, np, ni);
        if eqf.on {
          gpu::ok(unsafe { gpu::sp_sumcheck_eval_batched_eq(c, ap, bp, cp, ni, np, gpu::limbs_mut(&mut ev4)) });
          evc = eqf.combine(0, &ev4, coeffs, np, ni, &e);
        } else {
          gpu::ok(unsafe { gpu::sp_sumcheck_eval_batched(c, ap, bp, cp, ni, gpu::limbs_mut(&mut ev)) });
          evc = combine(&ev);
        }
      }
    }
    while j < num_rounds {
      let len = len_of(A[0]);
      let r_j = round_message(&evc, &mut e, &mut r, &mut cubic_polys, transcript);
      if eqf.on && len >= EqFactor::MIN_LEN {
        // a throughput-sized round in the factored form: A and B are bound at r_j, the eq table is only read
        gpu::ok(unsafe { gpu::sp_sumcheck_bind_eval_batched_eq(c, ap, bp, cp, ni, np, gpu::limbs1(&r_j), gpu::limbs_mut(&mut ev4)) });
        eqf.bound(j, &r_j);
        j += 1;
        evc = eqf.combine(j, &ev4, coeffs, np, ni, &e);
        continue;
      }
      if eqf.on {
        // hand-over: K * C_original[0 .. len) is the eq table bound at r_0 .. r_{j-1}; the generic rounds continue with it
        gpu::ok(unsafe { gpu::sp_table_scale_prefix(c, dev(poly_C_par), len, gpu::limbs1(&eqf.K)) });
        eqf.on = false;
      }
      if have_S && len >= 4 {
        // two rounds in one trip: the next round's evaluations are the device's cubic at r_j
        evc = evals_from_coeffs(&S, &r_j);
        let r_j1 = round_message(&evc, &mut e, &mut r, &mut cubic_polys, transcript);
        let n2 = len / 4;
        let want_tail = tail_ok && n2 >= 2 && n2 <= 8;
        let (pe, ps) = (if n2 >= 2 { gpu::limbs_mut(&mut evc) } else { null }, if n2 >= 4 { gpu::limbs_mut(&mut S) } else { null });
        if want_tail {
          mark(&mut tail);
          gpu::ok(unsafe { gpu::sp_sumcheck_bind2_eval_tables_batched(c, ap, bp, cp, ni, gpu::limbs1(&r_j), gpu::limbs1(&r_j1), gpu::limbs(coeffs), pe, ps, null, gpu::limbs_mut(&mut tail)) });
        } else {
          gpu::ok(unsafe {
            gpu::sp_sumcheck_bind2_eval_batched(c, ap, bp, cp, ni, gpu::limbs1(&r_j), gpu::limbs1(&r_j1), gpu::limbs(coeffs), pe, ps,
                                                if n2 == 1 { gpu::limbs_mut(&mut heads) } else { null })
          });
        }
        have_S = n2 >= 4;
        have_heads = n2 == 1;
        j += 2;
        if want_tail && tail_given(&tail) { finish_on_host!(n2); }
        continue;
      }
      if len >= 8 && len / 2 <= dmax {
        // the tables become short with this bind: it also returns the cubic, then two rounds per trip
        let n2 = len / 2;
        let want_tail = tail_ok && n2 <= 8;
        if want_tail {
          mark(&mut tail);
          gpu::ok(unsafe { gpu::sp_sumcheck_bind2_eval_tables_batched(c, ap, bp, cp, ni, gpu::limbs1(&r_j), std::ptr::null(), gpu::limbs(coeffs), gpu::limbs_mut(&mut evc), gpu::limbs_mut(&mut S), null, gpu::limbs_mut(&mut tail)) });
        } else {
          gpu::ok(unsafe { gpu::sp_sumcheck_bind2_eval_batched(c, ap, bp, cp, ni, gpu::limbs1(&r_j), std::ptr::null(), gpu::limbs(coeffs), gpu::limbs_mut(&mut evc), gpu::limbs_mut(&mut S), null) });
        }
        have_S = true;
        j += 1;
        if want_tail && tail_given(&tail) { finish_on_host!(n2); }
        continue;
      } else if len >= 4 {
        gpu::ok(unsafe { gpu::sp_sumcheck_bind_eval_batched(c, ap, bp, cp, ni, gpu::limbs1(&r_j), gpu::limbs_mut(&mut ev)) });
        evc = combine(&ev);
      } else {
        gpu::ok(unsafe { gpu::sp_table_bind_top_heads(c, all.as_ptr(), all.len(), gpu::limbs1(&r_j), gpu::limbs_mut(&mut heads)) });
        have_heads = true;
      }
      j += 1;
    }
    if !have_heads { gpu::ok(unsafe { gpu::sp_table_heads(c, all.as_ptr(), all.len(), gpu::limbs_mut(&mut heads)) }); }

    // host-side bookkeeping of the bound polynomials (their tables were halved num_rounds times on the device)
    for p in poly_A_vec_par.iter_mut().chain(poly_B_vec_par.iter_mut()).chain(poly_A_vec_seq.iter_mut())
      .chain(poly_B_vec_seq.iter_mut()).chain(poly_C_vec_seq.iter_mut()) {
      p.num_vars -= num_rounds;
      p.len >>= num_rounds;
    }
    poly_C_par.num_vars -= num_rounds;
    poly_C_par.len >>= num_rounds;
    let claims_prod = ((0..np).map(|k| heads[2 * k]).collect(), (0..np).map(|k| heads[2 * k + 1]).collect(), heads[2 * ni]);
    let claims_dotp = (
      (0..ns).map(|k| heads[2 * (np + k)]).collect(),
      (0..ns).map(|k| heads[2 * (np + k) + 1]).collect(),
      (0..ns).map(|k| heads[2 * ni + 1 + k]).collect(),
    );
    (SumcheckInstanceProof::new(cubic_polys), r, claims_prod, claims_dotp)
  }
}
