// src/sparse_mlpoly.rs (+ the three R1CSShape methods of src/r1cs.rs:268-303 that call into it) under `--features gpu`.
// C++ rendering: spartan_amd/host/spark.inc (SNARK::encode, layers_new, product_layer_prove, hash_layer_prove,
// derefs_early_begin, sparse_eval_prove) and prover.cc (Instance: sp_sparse_upload). Same calls, same order.
use super::gpu::{self, sp_table};
use super::product_tree::product_circuits_evaluate;

// ---------------------------------------------------------------------------------- sparse matrices (:19-38, :429-481)
#[cfg(feature = "gpu")]
impl SparseMatPolynomial {
  /// the entries on the device, row- and column-sorted (CSR + CSC: both products become gathers; F_q has no atomic add).
  /// Uploaded once per matrix (SparseMatPolynomial::new, :344-350) and kept in `self.dev` (gpu::Sparse: sp_sparse_free on drop).
  pub fn upload(&mut self, num_rows: usize, num_cols: usize) {
    let rows: Vec<u64> = self.M.iter().map(|e| e.row as u64).collect();
    let cols: Vec<u64> = self.M.iter().map(|e| e.col as u64).collect();
    let vals: Vec<Scalar> = self.M.iter().map(|e| e.val).collect();
    let mut h = std::ptr::null_mut();
    gpu::ok(unsafe { gpu::sp_sparse_upload(gpu::ctx(), rows.as_ptr(), cols.as_ptr(), gpu::limbs(&vals), vals.len(), num_rows, num_cols, &mut h) });
    self.dev = Some(gpu::Sparse(h));
  }
  /// multiply_vec (:454-464): out[row] = sum val * z[col], left on the device
  pub fn multiply_vec_dev(&self, z: &DensePolynomial) -> DensePolynomial {
    let mut out = std::ptr::null_mut();
    gpu::ok(unsafe { gpu::sp_sparse_mulvec(gpu::ctx(), self.dev.as_ref().unwrap().0, z.dev.as_ref().unwrap().0, &mut out) });
    DensePolynomial::from_dev(gpu::Table(out))
  }
  /// evaluate_with_tables (:429-438)
  pub fn evaluate_with_tables_dev(&self, tx: &gpu::Table, ty: &gpu::Table) -> Scalar {
    let mut out = Scalar::zero();
    gpu::ok(unsafe { gpu::sp_sparse_evaluate(gpu::ctx(), self.dev.as_ref().unwrap().0, tx.0, ty.0, &mut out as *mut Scalar as *mut u64) });
    out
  }
}
#[cfg(feature = "gpu")]
impl R1CSShape {
  /// multiply_vec (r1cs.rs:268-282)
  pub fn multiply_vec_dev(&self, z: &DensePolynomial) -> (DensePolynomial, DensePolynomial, DensePolynomial) {
    (self.A.multiply_vec_dev(z), self.B.multiply_vec_dev(z), self.C.multiply_vec_dev(z))
  }
  /// compute_eval_table_sparse (r1cs.rs:284-298) x3 and the combination r_A A + r_B B + r_C C of r1csproof.rs:275-283 in one call
  pub fn compute_eval_table_sparse_dev(&self, evals_rx: &gpu::Table, w: &[Scalar; 3]) -> DensePolynomial {
    let ms = [self.A.dev.as_ref().unwrap().0 as *const _, self.B.dev.as_ref().unwrap().0 as *const _, self.C.dev.as_ref().unwrap().0 as *const _];
    let mut out = std::ptr::null_mut();
    gpu::ok(unsafe { gpu::sp_sparse_eval_table(gpu::ctx(), ms.as_ptr(), gpu::limbs(w), 3, evals_rx.0, &mut out) });
    DensePolynomial::from_dev(gpu::Table(out))
  }
  /// evaluate (r1cs.rs:300-303) -> multi_evaluate (:440-452)
  pub fn evaluate_dev(&self, rx: &[Scalar], ry: &[Scalar]) -> (Scalar, Scalar, Scalar) {
    let (tx, ty) = (gpu::Table::eq(rx), gpu::Table::eq(ry));
    (self.A.evaluate_with_tables_dev(&tx, &ty), self.B.evaluate_with_tables_dev(&tx, &ty), self.C.evaluate_with_tables_dev(&tx, &ty))
  }
}

// ---------------------------------------------------------------------------------- SNARK::encode: dense representation (:367-427)
#[cfg(feature = "gpu")]
impl SparseMatPolynomial {
  /// multi_sparse_to_dense_rep (:370-427) with AddrTimestamps::new (:221-254) on the device: comb_ops = merge(row.ops_addr x3,
  /// row.read_ts x3, col.ops_addr x3, col.read_ts x3, val x3) and comb_mem = row.audit_ts | col.audit_ts are ONE table each; the
  /// member polynomials are views into them. The sequential timestamp scan per address becomes a stable sort by address plus
  /// rank-in-run (sp_addr_timestamps).
  pub fn multi_sparse_to_dense_rep_dev(sparse_polys: &[&SparseMatPolynomial]) -> MultiSparseMatPolynomialAsDense {
    let c = gpu::ctx();
    let N = sparse_polys.iter().map(|p| p.get_num_nz_entries()).max().unwrap().next_power_of_two();
    let (nvx, nvy) = (sparse_polys[0].num_vars_x, sparse_polys[0].num_vars_y);
    let cells = if nvx > nvy { nvx.pow2() } else { nvy.pow2() };
    let comb_ops = gpu::Table::alloc_zeroed((15 * N).next_power_of_two());
    let comb_mem = gpu::Table::alloc_zeroed(2 * cells);
    let side = |is_row: bool, slot0: usize, mem_off: usize| -> AddrTimestamps {
      let (mut usize_lists, mut ops_addr, mut read_ts, mut handles, mut ts_off) = (Vec::new(), Vec::new(), Vec::new(), Vec::new(), [0usize; 3]);
      for (k, p) in sparse_polys.iter().enumerate() {
        // the entries' addresses from the entry-order copy sp_sparse keeps on the device (zero-padded to N): no host pass, no upload
        let mut ix = std::ptr::null_mut();
        gpu::ok(unsafe { gpu::sp_sparse_entry_index(c, p.dev.as_ref().unwrap().0, if is_row { 0 } else { 1 }, N, &mut ix) });
        gpu::ok(unsafe { gpu::sp_table_from_index(c, ix, comb_ops.0, (slot0 + k) * N) }); // DensePolynomial::from_usize (:246-248)
        ops_addr.push(DensePolynomial::from_dev(comb_ops.view((slot0 + k) * N, N)));
        read_ts.push(DensePolynomial::from_dev(comb_ops.view((slot0 + 3 + k) * N, N)));
        ts_off[k] = (slot0 + 3 + k) * N;
        handles.push(ix);
        usize_lists.push(gpu::Index(ix)); // sp_index_free on drop
      }
      gpu::ok(unsafe { gpu::sp_addr_timestamps(c, handles.as_ptr(), 3, cells, comb_ops.0, ts_off.as_ptr(), comb_mem.0, mem_off) });
      AddrTimestamps { ops_addr_usize: usize_lists, ops_addr, read_ts, audit_ts: DensePolynomial::from_dev(comb_mem.view(mem_off, cells)) }
    };
    let row = side(true, 0, 0);
    let col = side(false, 6, cells);
    let mut val = Vec::new();
    for (k, p) in sparse_polys.iter().enumerate() {
      gpu::ok(unsafe { gpu::sp_sparse_entry_values(c, p.dev.as_ref().unwrap().0, comb_ops.0, (12 + k) * N, N) });
      val.push(DensePolynomial::from_dev(comb_ops.view((12 + k) * N, N)));
    }
    MultiSparseMatPolynomialAsDense { batch_size: sparse_polys.len(), val, row, col, comb_ops: DensePolynomial::from_dev(comb_ops), comb_mem: DensePolynomial::from_dev(comb_mem) }
  }
  // multi_commit (:483-503) is unchanged reference code: dense.comb_ops.commit(..) / dense.comb_mem.commit(..) reach the device
  // through DensePolynomial::commit_inner (seams/dense_mlpoly.rs).
}

// ---------------------------------------------------------------------------------- layers (:529-678)
#[cfg(feature = "gpu")]
impl Layers {
  /// Layers::new (:606-655): build_hash_layer (:529-604) writes the hashed leaves straight into the circuit stores; the
  /// multiplication layers of all circuits of one size are built together.
  pub fn new_dev(eval_table: &gpu::Table, at: &AddrTimestamps, poly_ops_val: &[DensePolynomial], r_mem_check: &(Scalar, Scalar)) -> Self {
    let c = gpu::ctx();
    let (r_hash, r_multiset) = (&r_mem_check.0, &r_mem_check.1);
    let cells = eval_table.len();
    let dev = |p: &DensePolynomial| p.dev.as_ref().unwrap().0 as *const sp_table;
    let leaves = |addr: *const sp_table, val: *const sp_table, ts: *const sp_table, ts_inc: i32, n: usize| -> gpu::Table {
      let store = gpu::Table::alloc_uninit(2 * n); // leaves [0, n) and layers [n, 2n - 2) are all written before they are read
      gpu::ok(unsafe { gpu::sp_hash_layer(c, addr, val, ts, ts_inc, n, gpu::limbs1(r_hash), gpu::limbs1(r_multiset), store.0, 0) });
      store
    };
    let null = std::ptr::null::<sp_table>();
    let n_ops = at.ops_addr[0].len();
    // the hash layer writes the first multiplication layer too, and the read and write sets of a matrix (ts, ts + 1) come out of one pass
    // (sp_hash_layer_first); option spark.hash_fuse = 0: the separate launches. Mechanical call sequence (spark.inc layers_new is the twin).
    let fuse = gpu::hash_fuse_enabled() && cells >= 4 && n_ops >= 4;
    // leaves [0, n) and layer 1 [n, n + n/2) of `store` (and of `store_w`, the write set: ts + 1) in one pass
    let first = |addr: *const sp_table, val: *const sp_table, ts: *const sp_table, n: usize, with_write: bool| -> (gpu::Table, Option<gpu::Table>) {
      let store = gpu::Table::alloc_uninit(2 * n);
      let store_w = if with_write { Some(gpu::Table::alloc_uninit(2 * n)) } else { None };
      let pw = store_w.as_ref().map_or(std::ptr::null_mut(), |t| t.0);
      gpu::ok(unsafe { gpu::sp_hash_layer_first(c, addr, val, ts, 0, n, gpu::limbs1(r_hash), gpu::limbs1(r_multiset), store.0, pw) });
      (store, store_w)
    };
    let (mut reads, mut writes) = (Vec::new(), Vec::new());
    let (init, audit);
    if fuse {
      init = first(null, eval_table.0, null, cells, false).0;
      for k in 0..at.ops_addr.len() {
        let (rd, wr) = first(dev(&at.ops_addr[k]), dev(&poly_ops_val[k]), dev(&at.read_ts[k]), n_ops, true);
        reads.push(rd);
        writes.push(wr.unwrap());
      }
      audit = first(null, eval_table.0, dev(&at.audit_ts), cells, false).0;
    } else {
      init = leaves(null, eval_table.0, null, 0, cells);
      for k in 0..at.ops_addr.len() {
        reads.push(leaves(dev(&at.ops_addr[k]), dev(&poly_ops_val[k]), dev(&at.read_ts[k]), 0, n_ops));
        writes.push(leaves(dev(&at.ops_addr[k]), dev(&poly_ops_val[k]), dev(&at.read_ts[k]), 1, n_ops));
      }
      audit = leaves(null, eval_table.0, dev(&at.audit_ts), 0, cells);
    }
    let done = if fuse { 1 } else { 0 };
    let mut mem = ProductCircuit::new_many(vec![init, audit], cells, done); // sp_product_tree_many_from
    let nr = reads.len();
    let mut ops = ProductCircuit::new_many(reads.into_iter().chain(writes).collect(), n_ops, done);
    let write_vec = ops.split_off(nr);
    let audit = mem.pop().unwrap();
    Layers { prod_layer: ProductLayer { init: mem.pop().unwrap(), read_vec: ops, write_vec, audit } }
  }
}

// ---------------------------------------------------------------------------------- ProductLayerProof::prove (:1035-1214)
#[cfg(feature = "gpu")]
impl ProductLayerProof {
  pub fn prove_gpu(
    row: &mut ProductLayer, col: &mut ProductLayer, dense: &MultiSparseMatPolynomialAsDense, derefs: &Derefs, eval: &[Scalar], transcript: &mut Transcript,
  ) -> (Self, Vec<Scalar>, Vec<Scalar>) {
    transcript.append_protocol_name(ProductLayerProof::protocol_name());
    let side = |L: &ProductLayer, transcript: &mut Transcript, li: &'static [u8], lr: &'static [u8], lw: &'static [u8], la: &'static [u8]| {
      let mut ps: Vec<&ProductCircuit> = vec![&L.init, &L.audit];
      ps.extend(L.read_vec.iter());
      ps.extend(L.write_vec.iter());
      let ev = product_circuits_evaluate(&ps); // one round trip for all the roots
      let (init, audit) = (ev[0], ev[1]);
      let rd = ev[2..2 + L.read_vec.len()].to_vec();
      let wr = ev[2 + L.read_vec.len()..].to_vec();
      let ws: Scalar = wr.iter().product();
      let rs: Scalar = rd.iter().product();
      assert_eq!(init * ws, rs * audit); // subset check (:1057-1060)
      init.append_to_transcript(li, transcript);
      rd.append_to_transcript(lr, transcript);
      wr.append_to_transcript(lw, transcript);
      audit.append_to_transcript(la, transcript);
      (init, rd, wr, audit)
    };
    let (row_eval_init, row_eval_read, row_eval_write, row_eval_audit) =
      side(row, transcript, b"claim_row_eval_init", b"claim_row_eval_read", b"claim_row_eval_write", b"claim_row_eval_audit");
    let (col_eval_init, col_eval_read, col_eval_write, col_eval_audit) =
      side(col, transcript, b"claim_col_eval_init", b"claim_col_eval_read", b"claim_col_eval_write", b"claim_col_eval_audit");
    // dot-product circuits over clones (the sum-check binds them in place; the originals are evaluated again later)
    let nb = derefs.row_ops_val.len();
    assert_eq!(eval.len(), nb);
    let clone = |p: &DensePolynomial| -> gpu::Table { let mut t = std::ptr::null_mut(); gpu::ok(unsafe { gpu::sp_table_clone(gpu::ctx(), p.dev.as_ref().unwrap().0, &mut t) }); gpu::Table(t) };
    let (mut dl, mut dr, mut keep) = (Vec::new(), Vec::new(), Vec::new());
    for i in 0..nb {
      let (cr, cc, cv) = (clone(&derefs.row_ops_val[i]), clone(&derefs.col_ops_val[i]), clone(&dense.val[i]));
      let half = cr.len() / 2; // DotProductCircuit::split (:90-109)
      let h = |t: &gpu::Table, off: usize| DensePolynomial::from_dev(t.view(off, half));
      dl.push(DotProductCircuit { left: h(&cr, 0), right: h(&cc, 0), weight: h(&cv, 0) });
      dr.push(DotProductCircuit { left: h(&cr, half), right: h(&cc, half), weight: h(&cv, half) });
      keep.push((cr, cc, cv));
    }
    // the six DotProductCircuit::evaluate (:1084-1101) in one launch; appended in the reference's order
    let dev = |p: &DensePolynomial| p.dev.as_ref().unwrap().0 as *const sp_table;
    let (mut L6, mut R6, mut W6) = (Vec::new(), Vec::new(), Vec::new());
    for i in 0..nb {
      L6.push(dev(&dl[i].left)); R6.push(dev(&dl[i].right)); W6.push(dev(&dl[i].weight));
      L6.push(dev(&dr[i].left)); R6.push(dev(&dr[i].right)); W6.push(dev(&dr[i].weight));
    }
    let mut v = vec![Scalar::zero(); 2 * nb];
    gpu::ok(unsafe { gpu::sp_dot3_many(gpu::ctx(), L6.as_ptr(), R6.as_ptr(), W6.as_ptr(), 2 * nb, dl[0].left.len(), gpu::limbs_mut(&mut v)) });
    let (mut eval_dotp_left_vec, mut eval_dotp_right_vec) = (Vec::new(), Vec::new());
    for i in 0..nb {
      let (el, er) = (v[2 * i], v[2 * i + 1]);
      el.append_to_transcript(b"claim_eval_dotp_left", transcript);
      er.append_to_transcript(b"claim_eval_dotp_right", transcript);
      assert_eq!(el + er, eval[i]);
      eval_dotp_left_vec.push(el);
      eval_dotp_right_vec.push(er);
    }
    // The first batch: the ops-related product circuits with the dot-product circuits (:1108-1150)
    let (r0, r1, r2) = { let s = row.read_vec.as_mut_slice(); let (a, rest) = s.split_at_mut(1); let (b, c) = rest.split_at_mut(1); (&mut a[0], &mut b[0], &mut c[0]) };
    let (w0, w1, w2) = { let s = row.write_vec.as_mut_slice(); let (a, rest) = s.split_at_mut(1); let (b, c) = rest.split_at_mut(1); (&mut a[0], &mut b[0], &mut c[0]) };
    let (c0, c1, c2) = { let s = col.read_vec.as_mut_slice(); let (a, rest) = s.split_at_mut(1); let (b, c) = rest.split_at_mut(1); (&mut a[0], &mut b[0], &mut c[0]) };
    let (x0, x1, x2) = { let s = col.write_vec.as_mut_slice(); let (a, rest) = s.split_at_mut(1); let (b, c) = rest.split_at_mut(1); (&mut a[0], &mut b[0], &mut c[0]) };
    let mut dps: Vec<&mut DotProductCircuit> = Vec::new();
    for (l, r) in dl.iter_mut().zip(dr.iter_mut()) { dps.push(l); dps.push(r); }
    let dps_evals: Vec<Scalar> = (0..nb).flat_map(|i| [eval_dotp_left_vec[i], eval_dotp_right_vec[i]]).collect();
    // the circuits' roots were evaluated and absorbed above (the claims of :1043-1100): handed on, a device trip less per batch
    let ops_roots: Vec<Scalar> = row_eval_read.iter().chain(row_eval_write.iter()).chain(col_eval_read.iter()).chain(col_eval_write.iter()).cloned().collect();
    let (proof_ops, rand_ops) = ProductCircuitEvalProofBatched::prove_gpu(&mut [r0, r1, r2, w0, w1, w2, c0, c1, c2, x0, x1, x2], &mut dps, Some(&dps_evals), Some(&ops_roots), transcript);
    // The second batch: the memory-related product circuits
    let mem_roots = [row_eval_init, row_eval_audit, col_eval_init, col_eval_audit];
    let (proof_mem, rand_mem) = ProductCircuitEvalProofBatched::prove_gpu(&mut [&mut row.init, &mut row.audit, &mut col.init, &mut col.audit], &mut [], None, Some(&mem_roots), transcript);
    drop(keep);
    (
      ProductLayerProof {
        eval_row: (row_eval_init, row_eval_read, row_eval_write, row_eval_audit),
        eval_col: (col_eval_init, col_eval_read, col_eval_write, col_eval_audit),
        eval_val: (eval_dotp_left_vec, eval_dotp_right_vec),
        proof_mem, proof_ops,
      },
      rand_mem, rand_ops,
    )
  }
}

// ---------------------------------------------------------------------------------- HashLayerProof::prove (:722-835)
#[cfg(feature = "gpu")]
impl HashLayerProof {
  pub fn prove_gpu(
    rand: (&Vec<Scalar>, &Vec<Scalar>), dense: &MultiSparseMatPolynomialAsDense, derefs: &Derefs, gens: &SparseMatPolyCommitmentGens,
    transcript: &mut Transcript, random_tape: &mut RandomTape,
  ) -> Self {
    transcript.append_protocol_name(HashLayerProof::protocol_name());
    let (rand_mem, rand_ops) = rand;
    // all DensePolynomial::evaluate(rand_ops) calls share chi(rand_ops); the audit polynomials share chi(rand_mem): two launches
    let (chi_ops, chi_mem) = (gpu::Table::eq(rand_ops), gpu::Table::eq(rand_mem));
    let dev = |p: &DensePolynomial| p.dev.as_ref().unwrap().0;
    let ops_tabs: Vec<*mut sp_table> = derefs.row_ops_val.iter().chain(derefs.col_ops_val.iter()).chain(dense.row.ops_addr.iter()).chain(dense.row.read_ts.iter())
      .chain(dense.col.ops_addr.iter()).chain(dense.col.read_ts.iter()).chain(dense.val.iter()).map(dev).collect();
    // SURVEY 8e (C++ rendering: dot_many_sharded, spark.inc): with W shards, shard g evaluates <chi, T_k> over the contiguous chunk
    // [g n/W, (g+1) n/W) of every table — views, nothing copied — and the W partial vectors are added (exact F_q sums: same bytes)
    let dot_many = |chi: *mut sp_table, tabs: &[*mut sp_table], out: &mut [Scalar]| {
      let ctxs = gpu::shard_ctxs();
      let (w, n, nt) = (ctxs.len(), unsafe { gpu::sp_table_len(chi) }, tabs.len());
      if w < 2 || n % w != 0 || n / w < 1024 {
        gpu::ok(unsafe { gpu::sp_dot_many(gpu::ctx(), chi, tabs.as_ptr(), nt, gpu::limbs_mut(out)) });
        return;
      }
      gpu::ok(unsafe { gpu::sp_ctx_sync(gpu::ctx()) });
      let per = n / w;
      let mut all = vec![Scalar::zero(); w * nt];
      for g in 0..w {
        let view = |t: *mut sp_table| { let mut v = std::ptr::null_mut(); gpu::ok(unsafe { gpu::sp_table_view(ctxs[g], t, g * per, per, &mut v) }); gpu::Table(v) };
        let (chi_v, tv): (gpu::Table, Vec<gpu::Table>) = (view(chi), tabs.iter().map(|&t| view(t)).collect());
        let h: Vec<*mut sp_table> = tv.iter().map(|t| t.0).collect();
        gpu::ok(unsafe { gpu::sp_dot_many(ctxs[g], chi_v.0, h.as_ptr(), nt, gpu::limbs_mut(&mut all[g * nt..(g + 1) * nt])) });
      }
      for k in 0..nt { out[k] = (0..w).map(|g| all[g * nt + k]).sum(); }
    };
    let mut ev = vec![Scalar::zero(); ops_tabs.len()];
    dot_many(chi_ops.0, &ops_tabs, &mut ev);
    let mem_tabs = [dev(&dense.row.audit_ts), dev(&dense.col.audit_ts)];
    let mut evm = vec![Scalar::zero(); 2];
    dot_many(chi_mem.0, &mem_tabs, &mut evm);
    let (eval_row_ops_val, eval_col_ops_val) = (ev[0..3].to_vec(), ev[3..6].to_vec());
    // DerefsEvalProof::prove (:124-149) -> prove_single (:79-121): the n-to-1 reduction is O(8) scalars of reference code
    let proof_derefs = DerefsEvalProof::prove(derefs, &eval_row_ops_val, &eval_col_ops_val, rand_ops, &gens.gens_derefs, transcript, random_tape);
    let (eval_row_addr_vec, eval_row_read_ts_vec) = (ev[6..9].to_vec(), ev[9..12].to_vec());
    let (eval_col_addr_vec, eval_col_read_ts_vec) = (ev[12..15].to_vec(), ev[15..18].to_vec());
    let eval_val_vec = ev[18..21].to_vec();
    let (eval_row_audit_ts, eval_col_audit_ts) = (evm[0], evm[1]);
    // from here on the reference's code unchanged (:780-835): the two joint claims and PolyEvalProof::prove on comb_ops / comb_mem
    let mut evals_ops: Vec<Scalar> = Vec::new();
    evals_ops.extend(&eval_row_addr_vec); evals_ops.extend(&eval_row_read_ts_vec); evals_ops.extend(&eval_col_addr_vec);
    evals_ops.extend(&eval_col_read_ts_vec); evals_ops.extend(&eval_val_vec);
    evals_ops.resize(evals_ops.len().next_power_of_two(), Scalar::zero());
    evals_ops.append_to_transcript(b"claim_evals_ops", transcript);
    let challenges_ops = transcript.challenge_vector(b"challenge_combine_n_to_one", evals_ops.len().log_2());
    let mut poly_evals_ops = DensePolynomial::new(evals_ops);
    for i in (0..challenges_ops.len()).rev() { poly_evals_ops.bound_poly_var_bot(&challenges_ops[i]); }
    let joint_claim_eval_ops = poly_evals_ops[0];
    let mut r_joint_ops = challenges_ops;
    r_joint_ops.extend(rand_ops);
    joint_claim_eval_ops.append_to_transcript(b"joint_claim_eval_ops", transcript);
    let (proof_ops, _) = PolyEvalProof::prove(&dense.comb_ops, None, &r_joint_ops, &joint_claim_eval_ops, None, &gens.gens_ops, transcript, random_tape);
    let evals_mem: Vec<Scalar> = vec![eval_row_audit_ts, eval_col_audit_ts];
    evals_mem.append_to_transcript(b"claim_evals_mem", transcript);
    let challenges_mem = transcript.challenge_vector(b"challenge_combine_two_to_one", evals_mem.len().log_2());
    let mut poly_evals_mem = DensePolynomial::new(evals_mem);
    for i in (0..challenges_mem.len()).rev() { poly_evals_mem.bound_poly_var_bot(&challenges_mem[i]); }
    let joint_claim_eval_mem = poly_evals_mem[0];
    let mut r_joint_mem = challenges_mem;
    r_joint_mem.extend(rand_mem);
    joint_claim_eval_mem.append_to_transcript(b"joint_claim_eval_mem", transcript);
    let (proof_mem, _) = PolyEvalProof::prove(&dense.comb_mem, None, &r_joint_mem, &joint_claim_eval_mem, None, &gens.gens_mem, transcript, random_tape);
    HashLayerProof {
      eval_row: (eval_row_addr_vec, eval_row_read_ts_vec, eval_row_audit_ts),
      eval_col: (eval_col_addr_vec, eval_col_read_ts_vec, eval_col_audit_ts),
      eval_val: eval_val_vec, eval_derefs: (eval_row_ops_val, eval_col_ops_val), proof_ops, proof_mem, proof_derefs,
    }
  }
}

// ---------------------------------------------------------------------------------- SparseMatPolyEvalProof::prove (:1447-1514)
/// The row half of `derefs` (row_ops_val of A, B, C: :506-513) depends on rx only, and rx is fixed by the FIRST sum-check of
/// R1CSProof::prove. Its share of the derefs commitment — the largest MSM of the proof — is started on the context's
/// background stream at that point (R1CSProof::prove_gpu's on_rx hook, seams/lib.rs) and runs under the latency-bound second
/// sum-check and the witness opening.
#[cfg(feature = "gpu")]
pub struct DerefsEarly { pub rx_ext: Vec<Scalar>, pub mem_rx: gpu::Table, pub comb: gpu::Table, pub job: Option<gpu::CommitJob>, pub rows_bg: usize }

#[cfg(feature = "gpu")]
impl SparseMatPolyEvalProof {
  pub fn derefs_early_begin(dense: &MultiSparseMatPolynomialAsDense, gens_derefs: &PolyCommitmentGens, rx: &[Scalar], ry_len: usize) -> DerefsEarly {
    let N = dense.row.ops_addr[0].len();
    let mut rx_ext = vec![Scalar::zero(); ry_len.saturating_sub(rx.len())]; // equalize (:1429-1445)
    rx_ext.extend_from_slice(rx);
    let mem_rx = gpu::Table::eq(&rx_ext);
    let comb = gpu::Table::alloc_zeroed((6 * N).next_power_of_two());
    for k in 0..3 { gpu::ok(unsafe { gpu::sp_gather(gpu::ctx(), mem_rx.0, dense.row.ops_addr_usize[k].0, comb.0, k * N) }); } // deref_mem (:256-265)
    let R_size = gens_derefs.gens.gens_n.n;
    let rows_bg = 3 * N / R_size; // rows of the L x R layout that hold row-half values only
    let job = if rows_bg >= 64 { Some(DensePolynomial::from_dev(comb.view(0, comb.len())).commit_begin_background(&gens_derefs.gens.gens_n, 0, rows_bg)) } else { None };
    DerefsEarly { rx_ext, mem_rx, comb, job, rows_bg }
  }

  pub fn prove_gpu(
    dense: &MultiSparseMatPolynomialAsDense, rx: &[Scalar], ry: &[Scalar], evals: &[Scalar], gens: &SparseMatPolyCommitmentGens,
    transcript: &mut Transcript, random_tape: &mut RandomTape, early: Option<DerefsEarly>,
  ) -> SparseMatPolyEvalProof {
    transcript.append_protocol_name(SparseMatPolyEvalProof::protocol_name());
    assert_eq!(evals.len(), dense.batch_size);
    let (rx_ext, ry_ext) = SparseMatPolyEvalProof::equalize(rx, ry);
    let N = dense.row.ops_addr[0].len();
    let c = gpu::ctx();
    let (mem_rx, comb, mut job, rows_bg) = match early {
      Some(e) if e.rx_ext == rx_ext => (e.mem_rx, e.comb, e.job, e.rows_bg),
      _ => {
        let mem_rx = gpu::Table::eq(&rx_ext);
        let comb = gpu::Table::alloc_zeroed((6 * N).next_power_of_two());
        for k in 0..3 { gpu::ok(unsafe { gpu::sp_gather(c, mem_rx.0, dense.row.ops_addr_usize[k].0, comb.0, k * N) }); }
        (mem_rx, comb, None, 0)
      }
    };
    let mem_ry = gpu::Table::eq(&ry_ext);
    for k in 0..3 { gpu::ok(unsafe { gpu::sp_gather(c, mem_ry.0, dense.col.ops_addr_usize[k].0, comb.0, (3 + k) * N) }); }
    let derefs = Derefs {
      row_ops_val: (0..3).map(|k| DensePolynomial::from_dev(comb.view(k * N, N))).collect(),
      col_ops_val: (0..3).map(|k| DensePolynomial::from_dev(comb.view((3 + k) * N, N))).collect(),
      comb: DensePolynomial::from_dev(comb),
    };
    // commit to non-deterministic choices of the prover (:1473-1478)
    let comm_derefs = {
      let g = &gens.gens_derefs.gens.gens_n;
      let L_size = derefs.comb.len() / g.n;
      let comm = if let Some(bg) = job.take() {
        // rows [0, rows_bg) are in flight on the background stream: queue the rest on the main stream, then collect; the
        // background half finishes first and its shares are absorbed while the main stream still works
        let rest = if L_size - rows_bg > 8 { Some(derefs.comb.commit_start(None, g, rows_bg, L_size - rows_bg)) } else { None };
        let mut C = bg.wait(); // sp_job_wait
        transcript.append_message(b"derefs_commitment", b"begin_derefs_commitment"); // :204-210
        transcript.append_message(b"comm_poly_row_col_ops_val", b"poly_commitment_begin");
        for s in C.iter() { transcript.append_point(b"poly_commitment_share", s); }
        let tail = match rest { Some(j) => j.wait(), None => derefs.comb.commit_rows_sync(g, rows_bg, L_size - rows_bg) }; // sp_commit_rows_dev
        for s in tail.iter() { transcript.append_point(b"poly_commitment_share", s); }
        C.extend(tail);
        transcript.append_message(b"comm_poly_row_col_ops_val", b"poly_commitment_end");
        transcript.append_message(b"derefs_commitment", b"end_derefs_commitment");
        DerefsCommitment { comm_ops_val: PolyCommitment { C } }
      } else {
        let comm = derefs.commit(&gens.gens_derefs); // Derefs::commit (:64-67)
        comm.append_to_transcript(b"comm_poly_row_col_ops_val", transcript);
        comm
      };
      comm
    };
    let poly_eval_network_proof = {
      let r_mem_check = transcript.challenge_vector(b"challenge_r_hash", 2);
      let mut net = PolyEvalNetwork {
        row_layers: Layers::new_dev(&mem_rx, &dense.row, &derefs.row_ops_val, &(r_mem_check[0], r_mem_check[1])),
        col_layers: Layers::new_dev(&mem_ry, &dense.col, &derefs.col_ops_val, &(r_mem_check[0], r_mem_check[1])),
      };
      // PolyEvalNetworkProof::prove (:1318-1352)
      transcript.append_protocol_name(PolyEvalNetworkProof::protocol_name());
      let (proof_prod_layer, rand_mem, rand_ops) =
        ProductLayerProof::prove_gpu(&mut net.row_layers.prod_layer, &mut net.col_layers.prod_layer, dense, &derefs, evals, transcript);
      let proof_hash_layer = HashLayerProof::prove_gpu((&rand_mem, &rand_ops), dense, &derefs, gens, transcript, random_tape);
      PolyEvalNetworkProof { proof_prod_layer, proof_hash_layer }
    };
    SparseMatPolyEvalProof { comm_derefs, poly_eval_network_proof }
  }
}
