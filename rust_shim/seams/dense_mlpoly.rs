// src/dense_mlpoly.rs — bodies swapped under `--features gpu` (same pattern as the `multicore` pair at :148-177).
// DensePolynomial gains `#[cfg(feature = "gpu")] dev: Option<gpu::Table>`: tables produced on the device (eq tables,
// Az/Bz/Cz, bound polynomials, product-circuit layers, views into the SPARK dense representation) stay there; `Z` is only
// materialised by `download()` when host code indexes it. The C++ rendering is spartan_amd/host/prover.cc (poly_commit,
// polyeval_prove) — same calls, same order.
use super::gpu;

impl DensePolynomial {
  /// A polynomial that lives on the device: `len` elements of table `t` (an owned table or a view, gpu::Table::view).
  #[cfg(feature = "gpu")]
  pub fn from_dev(t: gpu::Table) -> Self {
    let len = t.len();
    DensePolynomial { num_vars: len.log_2(), len, Z: Vec::new(), dev: Some(t) }
  }
  #[cfg(feature = "gpu")]
  fn table(&self) -> std::borrow::Cow<'_, gpu::Table> {
    match &self.dev {
      Some(t) => std::borrow::Cow::Borrowed(t),
      None => std::borrow::Cow::Owned(gpu::Table::upload(&self.Z)), // sp_table_upload
    }
  }

  /// DensePolynomial::commit_inner (:164-177): L row commitments in one call.
  #[cfg(feature = "gpu")]
  fn commit_inner(&self, blinds: &[Scalar], gens: &MultiCommitGens) -> PolyCommitment {
    let L_size = blinds.len();
    let R_size = self.len() / L_size;
    assert_eq!(L_size * R_size, self.len());
    assert_eq!(gens.n, R_size);
    let d = &gens.dev;
    let t = self.table();
    let mut out = vec![0u8; 32 * L_size];
    let shards = gpu::shard_ctxs();
    let w = shards.len();
    if w >= 2 && L_size <= 8 && R_size % w == 0 {
      // fewer rows than a shard is worth -> sharded by COLUMNS (C++ rendering: sharded_commit_cols, spartan_amd/host/shard.cc).
      // Shard k sums the generators [k R/W, (k+1) R/W) of every row into one partial point per row; the points are added here
      // together with the blind terms and encoded.
      let per = R_size / w;
      let mut pts = vec![gpu::sp_host_point { w: [0u64; 16] }; (w + 1) * L_size];
      gpu::ok(unsafe { gpu::sp_ctx_sync(gpu::ctx()) });
      for k in 0..w {
        gpu::ok(unsafe { gpu::sp_commit_rows_partial(shards[k], d.g, d.G[0] as usize + k * per, t.0, k * per, R_size, L_size, per, pts[k * L_size..].as_mut_ptr()) });
      }
      let hh = [d.h];
      for r in 0..L_size {
        gpu::ok(unsafe { gpu::sp_host_commit_point(d.g, hh.as_ptr(), 1, gpu::limbs1(&blinds[r]), &mut pts[w * L_size + r]) });
      }
      gpu::ok(unsafe { gpu::sp_host_points_sum_encode(pts.as_ptr(), w + 1, L_size, out.as_mut_ptr()) });
      return PolyCommitment { C: out.chunks_exact(32).map(CompressedGroup::from_slice).collect() };
    }
    gpu::ok(unsafe {
      gpu::sp_commit_rows_dev(gpu::ctx(), d.g, d.G[0] as usize, d.h as usize, t.0, 0, L_size, R_size, gpu::limbs(blinds), out.as_mut_ptr())
    });
    PolyCommitment { C: out.chunks_exact(32).map(CompressedGroup::from_slice).collect() }
  }

  /// rows [row0, row0 + rows) of the L x R layout, no blinds, synchronous (a handful of rows left over next to a background job)
  #[cfg(feature = "gpu")]
  pub fn commit_rows_sync(&self, gens: &MultiCommitGens, row0: usize, rows: usize) -> Vec<CompressedGroup> {
    let d = &gens.dev;
    let t = self.dev.as_ref().expect("device-resident polynomial");
    let mut out = vec![0u8; 32 * rows];
    gpu::ok(unsafe { gpu::sp_commit_rows_dev(gpu::ctx(), d.g, d.G[0] as usize, d.h as usize, t.0, row0 * gens.n, rows, gens.n, std::ptr::null(), out.as_mut_ptr()) });
    out.chunks_exact(32).map(CompressedGroup::from_slice).collect()
  }

  /// The same commitment queued on the context's main stream without waiting (rows > 8): the caller does host work — R1CSProof::prove
  /// absorbs the transcript prefix while the witness commitment is computed — and collects it with `CommitJob::wait`.
  #[cfg(feature = "gpu")]
  pub fn commit_start(&self, blinds: Option<&[Scalar]>, gens: &MultiCommitGens, row0: usize, rows: usize) -> gpu::CommitJob {
    let R_size = gens.n;
    let d = &gens.dev;
    let t = self.dev.as_ref().expect("device-resident polynomial");
    let mut job = std::ptr::null_mut();
    gpu::ok(unsafe {
      gpu::sp_commit_rows_dev_start(gpu::ctx(), d.g, d.G[0] as usize, d.h as usize, t.0, row0 * R_size, rows, R_size,
                                    blinds.map_or(std::ptr::null(), |b| gpu::limbs(b)), &mut job)
    });
    gpu::CommitJob { job, rows } // wait(): sp_job_wait -> Vec<CompressedGroup>
  }
  /// ... and on the low-priority background stream (no blinds; persistent workgroups on half the CUs), for a commitment that
  /// can run under latency-bound work: the row half of `derefs`, started as soon as rx is known (seams/lib.rs).
  #[cfg(feature = "gpu")]
  pub fn commit_begin_background(&self, gens: &MultiCommitGens, row0: usize, rows: usize) -> gpu::CommitJob {
    let R_size = gens.n;
    let d = &gens.dev;
    let t = self.dev.as_ref().expect("device-resident polynomial");
    let mut job = std::ptr::null_mut();
    gpu::ok(unsafe { gpu::sp_commit_rows_dev_begin(gpu::ctx(), d.g, d.G[0] as usize, t.0, row0 * R_size, rows, R_size, &mut job) });
    gpu::CommitJob { job, rows }
  }

  /// DensePolynomial::bound (:206-213): LZ = L * Z, left on the device (PolyEvalProof::prove only commits to it and feeds it to
  /// the inner-product argument). Queued, not waited for: the caller computes the R vector meanwhile.
  #[cfg(feature = "gpu")]
  pub fn bound_dev(&self, L: &[Scalar]) -> gpu::Table {
    let (left_num_vars, _right_num_vars) = EqPolynomial::compute_factored_lens(self.get_num_vars());
    assert_eq!(L.len(), left_num_vars.pow2());
    let t = self.table();
    let shards = gpu::shard_ctxs();
    let w = shards.len();
    if w >= 2 && L.len() % w == 0 {
      // row blocks. Shard g multiplies rows [g L/W, (g+1) L/W) by its slice of L; the W partial vectors are added in F_q
      let (c, per, R_size) = (gpu::ctx(), L.len() / w, self.len() / L.len());
      gpu::ok(unsafe { gpu::sp_ctx_sync(c) });
      let mut parts: Vec<gpu::Table> = Vec::new();
      let mut views: Vec<gpu::Table> = Vec::new();
      for g in 0..w {
        views.push(gpu::Table::view_on(shards[g], &t, g * per * R_size, per * R_size)); // sp_table_view on the shard's context
        let mut pz = std::ptr::null_mut();
        gpu::ok(unsafe { gpu::sp_vecmat_dev(shards[g], gpu::limbs(&L[g * per..]), per, views[g].0, &mut pz) });
        parts.push(gpu::Table(pz));
      }
      for g in 1..w { gpu::ok(unsafe { gpu::sp_ctx_sync(shards[g]) }); }
      for g in 1..w { gpu::ok(unsafe { gpu::sp_table_add_into(c, parts[0].0, parts[g].0) }); }
      gpu::ok(unsafe { gpu::sp_ctx_sync(c) }); // the partial vectors go back to their contexts' pools when `parts` drops
      return parts.swap_remove(0);
    }
    let mut lz = std::ptr::null_mut();
    gpu::ok(unsafe { gpu::sp_vecmat_dev(gpu::ctx(), gpu::limbs(L), L.len(), t.0, &mut lz) });
    gpu::Table(lz)
  }

  /// DensePolynomial::bound_poly_var_top (:215-223).
  #[cfg(feature = "gpu")]
  pub fn bound_poly_var_top(&mut self, r: &Scalar) {
    let t = self.dev.as_ref().expect("device-resident polynomial");
    let tabs = [t.0];
    gpu::ok(unsafe { gpu::sp_table_bind_top(gpu::ctx(), tabs.as_ptr(), 1, gpu::limbs1(r)) });
    self.num_vars -= 1;
    self.len /= 2;
  }

  /// DensePolynomial::evaluate (:236-242): <Z, chi(r)> with chi generated on the device.
  #[cfg(feature = "gpu")]
  pub fn evaluate(&self, r: &[Scalar]) -> Scalar {
    assert_eq!(r.len(), self.get_num_vars());
    let t = self.table();
    let shards = gpu::shard_ctxs();
    let (w, lw) = (shards.len(), shards.len().max(1).trailing_zeros() as usize);
    if w >= 2 && r.len() > lw + 1 {
      // <Z, chi(r)> = sum_g chi_g(r[..lw]) <Z_g, chi(r[lw..])> over contiguous chunks, one scalar per shard
      gpu::ok(unsafe { gpu::sp_ctx_sync(gpu::ctx()) });
      let top = EqPolynomial::new(r[..lw].to_vec()).evals();
      let chunk = self.len() / w;
      let mut acc = Scalar::zero();
      for g in 0..w {
        let v = gpu::Table::view_on(shards[g], &t, g * chunk, chunk);
        let mut e = Scalar::zero();
        gpu::ok(unsafe { gpu::sp_evaluate(shards[g], v.0, gpu::limbs(&r[lw..]), r.len() - lw, &mut e as *mut Scalar as *mut u64) });
        acc += top[g] * e;
      }
      return acc;
    }
    let mut out = Scalar::zero();
    gpu::ok(unsafe { gpu::sp_evaluate(gpu::ctx(), t.0, gpu::limbs(r), r.len(), &mut out as *mut Scalar as *mut u64) });
    out
  }
}

impl EqPolynomial {
  /// EqPolynomial::evals (:68-84) as a device table (r[0] <-> most significant index bit, as in the reference).
  #[cfg(feature = "gpu")]
  pub fn evals_dev(&self) -> gpu::Table {
    gpu::Table::eq(&self.r) // sp_eq_expand
  }
}

impl PolyEvalProof {
  /// PolyEvalProof::prove (:312-365).
  #[cfg(feature = "gpu")]
  pub fn prove(
    poly: &DensePolynomial, blinds_opt: Option<&PolyCommitmentBlinds>, r: &[Scalar], Zr: &Scalar, blind_Zr_opt: Option<&Scalar>,
    gens: &PolyCommitmentGens, transcript: &mut Transcript, random_tape: &mut RandomTape,
  ) -> (PolyEvalProof, CompressedGroup) {
    let (proof, c, _zr) = PolyEvalProof::prove_eval(poly, blinds_opt, r, Some(Zr), blind_Zr_opt, gens, transcript, random_tape);
    (proof, c)
  }
  /// The same with the evaluation itself optional: `Zr = None` computes it here as <LZ, R> — the field element
  /// DensePolynomial::evaluate(r) = L^T Z R — from the vector-matrix product the opening needs anyway (sp_eq_expand + sp_dot: one
  /// launch-sized trip instead of a pass over the polynomial), and returns it. Used by R1CSProof::prove for `eval_vars_at_ry` (:299).
  #[cfg(feature = "gpu")]
  pub fn prove_eval(
    poly: &DensePolynomial, blinds_opt: Option<&PolyCommitmentBlinds>, r: &[Scalar], Zr: Option<&Scalar>, blind_Zr_opt: Option<&Scalar>,
    gens: &PolyCommitmentGens, transcript: &mut Transcript, random_tape: &mut RandomTape,
  ) -> (PolyEvalProof, CompressedGroup, Scalar) {
    transcript.append_protocol_name(PolyEvalProof::protocol_name());
    assert_eq!(poly.get_num_vars(), r.len());
    let (left_num_vars, right_num_vars) = EqPolynomial::compute_factored_lens(r.len());
    let (L_size, _R_size) = (left_num_vars.pow2(), right_num_vars.pow2());
    // the sqrt(N)-sized L and R vectors stay on the host (compute_factored_evals, :90-98); L first: the device multiplies
    // while R is computed
    // with blinds the host needs L too (LZ_blind); without, L is generated and consumed on the device (sp_eq_expand + sp_vecmat_tab)
    let (LZ, L) = match blinds_opt {
      Some(_) => { let L = EqPolynomial::new(r[..left_num_vars].to_vec()).evals(); (poly.bound_dev(&L), L) }
      None if gpu::shard_ctxs().len() >= 2 => { let L = EqPolynomial::new(r[..left_num_vars].to_vec()).evals(); (poly.bound_dev(&L), L) }
      None => {
        let Lt = gpu::Table::eq(&r[..left_num_vars]);
        let mut lz = std::ptr::null_mut();
        gpu::ok(unsafe { gpu::sp_vecmat_tab(gpu::ctx(), Lt.0, poly.table().0, &mut lz) });
        (gpu::Table(lz), Vec::new())
      }
    };
    let Rt = if Zr.is_none() { Some(gpu::Table::eq(&r[left_num_vars..])) } else { None };  // queued behind the product
    let R = EqPolynomial::new(r[left_num_vars..].to_vec()).evals();
    let Zr_val: Scalar = match (Zr, &Rt) {
      (Some(z), _) => *z,
      (None, Some(rt)) => {
        let mut out = Scalar::zero();
        gpu::ok(unsafe { gpu::sp_dot(gpu::ctx(), LZ.0, 0, rt.0, 0, R.len(), &mut out as *mut Scalar as *mut u64) });
        out
      }
      (None, None) => unreachable!(),
    };
    let Zr = &Zr_val;
    let LZ_blind: Scalar = match blinds_opt {
      Some(b) => { assert_eq!(b.blinds.len(), L_size); (0..L.len()).map(|i| b.blinds[i] * L[i]).sum() }
      None => Scalar::zero(),
    };
    let zero = Scalar::zero();
    let blind_Zr = blind_Zr_opt.map_or(&zero, |p| p);
    let (proof, _C_LR, C_Zr_prime) = DotProductProofLog::prove_dev(&gens.gens, transcript, random_tape, &LZ, &LZ_blind, &R, Zr, blind_Zr);
    (PolyEvalProof { proof }, C_Zr_prime, Zr_val)
  }
}
