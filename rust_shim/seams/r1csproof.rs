// src/r1csproof.rs — R1CSProof::prove (:144-349) under `--features gpu`. The reference's control flow, transcript and tape
// order, with every polynomial a device table. C++ rendering: spartan_amd/host/prover.cc (r1cs_prove).
// Two things are re-ordered on the HOST (never on the transcript): the witness commitment is queued first and the transcript
// prefix is absorbed while the GPU computes it; z and Az, Bz, Cz are queued before the commitment is absorbed.
use super::gpu;

#[cfg(feature = "gpu")]
pub struct ProveHooks<'a> {
  /// executed while the witness commitment is in flight: everything the caller's transcript absorbs BEFORE "R1CS proof"
  /// (SNARK::prove: the protocol name and the computation commitment, lib.rs:354-360; NIZK::prove: the shape digest, :514)
  pub transcript_prefix: &'a mut dyn FnMut(&mut Transcript),
  /// called as soon as the first sum-check has fixed rx (SNARK::prove starts the row half of the derefs commitment there)
  pub on_rx: Option<&'a mut dyn FnMut(&[Scalar])>,
  pub on_ry: Option<&'a mut dyn FnMut(&[Scalar])>,
}

#[cfg(feature = "gpu")]
impl R1CSProof {
  pub fn prove_gpu(
    inst: &R1CSShape,
    vars: gpu::VarsSource<'_>, // Host(&[Scalar]) or Resident(&gpu::Table): the assignment, NOT yet padded (lib.rs:360-368 pads)
    input: &[Scalar],
    gens: &R1CSGens,
    transcript: &mut Transcript,
    random_tape: &mut RandomTape,
    hooks: ProveHooks<'_>,
  ) -> (R1CSProof, Vec<Scalar>, Vec<Scalar>) {
    let c = gpu::ctx();
    let num_vars = inst.get_num_vars(); // padded
    assert!(input.len() < num_vars);
    // polycommit (:160-171): the padded assignment as a zero-filled device table with the given prefix written into it
    let poly_vars = DensePolynomial::from_dev(gpu::Table::alloc_zeroed(num_vars)); // sp_table_alloc
    let ell = poly_vars.get_num_vars();
    let (left_num_vars, _) = EqPolynomial::compute_factored_lens(ell);
    let L_size = left_num_vars.pow2();
    // a host assignment of full length goes up and is committed in one call: the additions of a chunk of rows run while the next chunk
    // crosses PCIe (sp_commit_rows_upload_start); otherwise upload / copy first, then commit
    let upload_commit = matches!(&vars, gpu::VarsSource::Host(v) if v.len() == num_vars) && L_size > 8;
    match &vars {
      gpu::VarsSource::Resident(t) => gpu::ok(unsafe { gpu::sp_table_copy(c, poly_vars.dev.as_ref().unwrap().0, 0, t.0, 0, t.len()) }),
      gpu::VarsSource::Host(v) if !upload_commit => gpu::ok(unsafe { gpu::sp_table_write(c, poly_vars.dev.as_ref().unwrap().0, 0, gpu::limbs(v), v.len()) }),
      _ => {}
    }
    let blinds_vars = PolyCommitmentBlinds { blinds: random_tape.random_vector(b"poly_blinds", L_size) };
    let job = if upload_commit {
      let (g, v) = (&gens.gens_pc.gens.gens_n, match &vars { gpu::VarsSource::Host(v) => *v, _ => unreachable!() });
      let mut job = std::ptr::null_mut();
      gpu::ok(unsafe {
        gpu::sp_commit_rows_upload_start(c, g.dev.g, g.dev.G[0] as usize, g.dev.h as usize, poly_vars.dev.as_ref().unwrap().0, 0, gpu::limbs(v), L_size, g.n,
                                         gpu::limbs(&blinds_vars.blinds), &mut job)
      });
      Some(gpu::CommitJob { job, rows: L_size })
    } else if L_size > 8 {
      Some(poly_vars.commit_start(Some(&blinds_vars.blinds), &gens.gens_pc.gens.gens_n, 0, L_size))
    } else {
      None
    };
    (hooks.transcript_prefix)(transcript);
    transcript.append_protocol_name(R1CSProof::protocol_name());
    input.append_to_transcript(b"input", transcript);
    let comm_vars = match job {
      Some(j) => PolyCommitment { C: j.wait() }, // sp_job_wait
      None => poly_vars.commit_inner(&blinds_vars.blinds, &gens.gens_pc.gens.gens_n),
    };
    // z = vars | 1 | input | 0.. (:177-185) and Az, Bz, Cz (:187-196): queued now, built while the commitment is absorbed
    let z = DensePolynomial::from_dev(gpu::Table::alloc_zeroed(2 * num_vars));
    gpu::ok(unsafe { gpu::sp_table_copy(c, z.dev.as_ref().unwrap().0, 0, poly_vars.dev.as_ref().unwrap().0, 0, num_vars) });
    let mut tail = vec![Scalar::one()];
    tail.extend_from_slice(input);
    gpu::ok(unsafe { gpu::sp_table_write(c, z.dev.as_ref().unwrap().0, num_vars, gpu::limbs(&tail), tail.len()) });
    let (mut poly_Az, mut poly_Bz, mut poly_Cz) = inst.multiply_vec_dev(&z); // 3 x sp_sparse_mulvec (seams/sparse_mlpoly.rs)
    comm_vars.append_to_transcript(b"poly_commitment", transcript);

    let (num_rounds_x, num_rounds_y) = (inst.get_num_cons().log_2(), (2 * num_vars).log_2());
    let tau = transcript.challenge_vector(b"challenge_tau", num_rounds_x);
    let mut poly_tau = DensePolynomial::from_dev(EqPolynomial::new(tau).evals_dev());
    // prove_phase_one (:76-104): comb A*(B*C - D), gens_4
    let (sc_proof_phase1, rx, claims1, blind_claim_postsc1) = ZKSumcheckInstanceProof::prove_zk_gpu(
      2, &Scalar::zero(), &Scalar::zero(), num_rounds_x, &mut [&mut poly_tau, &mut poly_Az, &mut poly_Bz, &mut poly_Cz],
      &gens.gens_sc.gens_1, &gens.gens_sc.gens_4, transcript, random_tape);
    if let Some(f) = hooks.on_rx { f(&rx); }
    let (tau_claim, Az_claim, Bz_claim, Cz_claim) = (&claims1[0], &claims1[1], &claims1[2], &claims1[3]);
    let (Az_blind, Bz_blind, Cz_blind, prod_Az_Bz_blind) = (
      random_tape.random_scalar(b"Az_blind"), random_tape.random_scalar(b"Bz_blind"),
      random_tape.random_scalar(b"Cz_blind"), random_tape.random_scalar(b"prod_Az_Bz_blind"));
    let (pok_Cz_claim, comm_Cz_claim) = KnowledgeProof::prove(&gens.gens_sc.gens_1, transcript, random_tape, Cz_claim, &Cz_blind);
    let (proof_prod, comm_Az_claim, comm_Bz_claim, comm_prod_Az_Bz_claims) = {
      let prod = Az_claim * Bz_claim;
      ProductProof::prove(&gens.gens_sc.gens_1, transcript, random_tape, Az_claim, &Az_blind, Bz_claim, &Bz_blind, &prod, &prod_Az_Bz_blind)
    };
    comm_Az_claim.append_to_transcript(b"comm_Az_claim", transcript);
    comm_Bz_claim.append_to_transcript(b"comm_Bz_claim", transcript);
    comm_Cz_claim.append_to_transcript(b"comm_Cz_claim", transcript);
    comm_prod_Az_Bz_claims.append_to_transcript(b"comm_prod_Az_Bz_claims", transcript);
    let blind_expected_claim_postsc1 = tau_claim * (prod_Az_Bz_blind - Cz_blind);
    let claim_post_phase1 = (Az_claim * Bz_claim - Cz_claim) * tau_claim;
    let (proof_eq_sc_phase1, _C1, _C2) = EqualityProof::prove(&gens.gens_sc.gens_1, transcript, random_tape,
      &claim_post_phase1, &blind_expected_claim_postsc1, &claim_post_phase1, &blind_claim_postsc1);

    let r_A = transcript.challenge_scalar(b"challenge_Az");
    let r_B = transcript.challenge_scalar(b"challenge_Bz");
    let r_C = transcript.challenge_scalar(b"challenge_Cz");
    let claim_phase2 = r_A * Az_claim + r_B * Bz_claim + r_C * Cz_claim;
    let blind_claim_phase2 = r_A * Az_blind + r_B * Bz_blind + r_C * Cz_blind;
    // evals_ABC (:271-284): r_A A(rx, .) + r_B B(rx, .) + r_C C(rx, .) in one call
    let evals_rx = EqPolynomial::new(rx.clone()).evals_dev();
    let mut poly_ABC = inst.compute_eval_table_sparse_dev(&evals_rx, &[r_A, r_B, r_C]); // sp_sparse_eval_table
    let mut poly_z = z;
    // prove_phase_two (:106-138): comb A*B, gens_3
    let (sc_proof_phase2, ry, claims_phase2, blind_claim_postsc2) = ZKSumcheckInstanceProof::prove_zk_gpu(
      0, &claim_phase2, &blind_claim_phase2, num_rounds_y, &mut [&mut poly_z, &mut poly_ABC],
      &gens.gens_sc.gens_1, &gens.gens_sc.gens_3, transcript, random_tape);
    if let Some(f) = hooks.on_ry { f(&ry); }

    // eval_vars_at_ry (:299): unsharded, it comes out of the opening's own vector-matrix product (<LZ, R>, PolyEvalProof::prove_eval);
    // sharded (or with option polyeval.eval_from_opening = 0), a pass of its own (sp_evaluate, chunked over the shards)
    let from_opening = gpu::shard_ctxs().len() < 2 && gpu::opt("polyeval.eval_from_opening") != 0;
    let eval_first = if from_opening { None } else { Some(poly_vars.evaluate(&ry[1..])) };
    let blind_eval = random_tape.random_scalar(b"blind_eval");
    let (proof_eval_vars_at_ry, comm_vars_at_ry, eval_vars_at_ry) = PolyEvalProof::prove_eval(&poly_vars, Some(&blinds_vars), &ry[1..],
      eval_first.as_ref(), Some(&blind_eval), &gens.gens_pc, transcript, random_tape);
    let _ = eval_vars_at_ry;
    let blind_eval_Z_at_ry = (Scalar::one() - ry[0]) * blind_eval;
    let blind_expected_claim_postsc2 = claims_phase2[1] * blind_eval_Z_at_ry;
    let claim_post_phase2 = claims_phase2[0] * claims_phase2[1];
    let (proof_eq_sc_phase2, _C1, _C2) = EqualityProof::prove(&gens.gens_pc.gens.gens_1, transcript, random_tape,
      &claim_post_phase2, &blind_expected_claim_postsc2, &claim_post_phase2, &blind_claim_postsc2);
    (
      R1CSProof {
        comm_vars, sc_proof_phase1,
        claims_phase2: (comm_Az_claim, comm_Bz_claim, comm_Cz_claim, comm_prod_Az_Bz_claims),
        pok_claims_phase2: (pok_Cz_claim, proof_prod),
        proof_eq_sc_phase1, sc_proof_phase2, comm_vars_at_ry, proof_eval_vars_at_ry, proof_eq_sc_phase2,
      },
      rx, ry,
    )
  }
}
