// src/lib.rs — SNARK::prove (:339-420), NIZK::prove (:501-546) and VarsAssignment under `--features gpu`, plus the seeded twins the
// parity tests need (the only change there is the RandomTape constructor, :356 / :516). Production code keeps OS entropy.
// C++ rendering: spartan_amd/host/spark.inc (SNARK::prove), prover.cc (NIZK::prove, VarsAssignment).
//
// TWO ways to bind the library, both shipped:
//   (A) fine-grained: the function bodies of seams/*.rs — libspartan keeps its own structs, transcript and control flow and calls
//       the sp_* kernels' entry points (this file's prove_gpu and everything it reaches);
//   (B) coarse: SNARK::prove / NIZK::prove hand the whole proof to libspartan_host.so (spz_snark_prove_t / spz_nizk_prove_t): the
//       caller's merlin transcript crosses as its 203-byte STROBE state and comes back advanced; the proof comes back as the
//       bincode bytes libspartan deserialises. (B) is what bench.py times; tests/test_gpu_proofs.py::
//       test_prove_continues_a_caller_owned_transcript checks it against the oracle on a transcript that is not fresh.
use super::gpu;

impl VarsAssignment {
  /// VarsAssignment::new (:56-105) additionally uploads the parsed scalars once (sp_table_upload inside gpu::Table::upload):
  /// SNARK::prove / NIZK::prove start from the device copy. `dev: Option<gpu::Table>` is the added field.
  #[cfg(feature = "gpu")]
  pub fn upload(&mut self) { self.dev = Some(gpu::Table::upload(&self.assignment)); }
}

impl SNARK {
  #[cfg(feature = "gpu")]
  pub fn prove(
    inst: &Instance, comm: &ComputationCommitment, decomm: &ComputationDecommitment, vars: VarsAssignment, inputs: &InputsAssignment,
    gens: &SNARKGens, transcript: &mut Transcript,
  ) -> Self {
    let mut random_tape = RandomTape::new(b"proof");
    Self::prove_with_tape(inst, comm, decomm, vars, inputs, gens, transcript, &mut random_tape)
  }
  /// TEST HOOK (byte-parity contract): a seed fixes every blind of the proof; outside tests never.
  #[cfg(feature = "gpu")]
  pub fn prove_with_tape_seed(
    inst: &Instance, comm: &ComputationCommitment, decomm: &ComputationDecommitment, vars: VarsAssignment, inputs: &InputsAssignment,
    gens: &SNARKGens, transcript: &mut Transcript, tape_seed: &Scalar,
  ) -> Self {
    let mut random_tape = RandomTape::new_with_seed(b"proof", tape_seed);
    Self::prove_with_tape(inst, comm, decomm, vars, inputs, gens, transcript, &mut random_tape)
  }

  #[cfg(feature = "gpu")]
  fn prove_with_tape(
    inst: &Instance, comm: &ComputationCommitment, decomm: &ComputationDecommitment, vars: VarsAssignment, inputs: &InputsAssignment,
    gens: &SNARKGens, transcript: &mut Transcript, random_tape: &mut RandomTape,
  ) -> Self {
    // lib.rs:354-360 is executed by R1CSProof::prove_gpu while the witness commitment is in flight (same transcript order)
    let mut prefix = |t: &mut Transcript| {
      t.append_protocol_name(SNARK::protocol_name());
      comm.comm.append_to_transcript(b"comm", t);
    };
    // the row half of the derefs commitment starts as soon as rx is known (seams/sparse_mlpoly.rs: DerefsEarly)
    let overlap = gpu::opt("overlap.derefs") != 0;
    let ry_len = inst.inst.get_num_vars().log_2() + 1;
    let mut early: Option<DerefsEarly> = None;
    let mut rx_seen: Vec<Scalar> = Vec::new();
    let mut on_rx = |rx: &[Scalar]| {
      rx_seen = rx.to_vec();
      early = Some(SparseMatPolyEvalProof::derefs_early_begin(&decomm.decomm.dense, &gens.gens_r1cs_eval.gens.gens_derefs, rx, ry_len));
    };
    // R1CSInstance::evaluate (r1cs.rs:300-303) needs rx, ry only: queued on a low-priority stream when the second sum-check ends
    // (sp_sparse_evaluate_begin), it runs in the idle time of the witness opening; collected below with sp_job_wait
    let eval_ahead = overlap && gpu::opt("overlap.eval_ahead") != 0;
    let mut ahead: Option<(gpu::Table, gpu::Table, gpu::CommitJob)> = None;
    let mut on_ry = |ry: &[Scalar]| {
      let (tx, ty) = (gpu::Table::eq(&rx_seen), gpu::Table::eq(ry));
      let ms = [inst.inst.A.dev.as_ref().unwrap().0 as *const gpu::sp_sparse, inst.inst.B.dev.as_ref().unwrap().0 as *const gpu::sp_sparse,
                inst.inst.C.dev.as_ref().unwrap().0 as *const gpu::sp_sparse];
      let mut job = std::ptr::null_mut();
      gpu::ok(unsafe { gpu::sp_sparse_evaluate_begin(gpu::ctx(), ms.as_ptr(), 3, tx.0, ty.0, &mut job) });
      ahead = Some((tx, ty, gpu::CommitJob { job, rows: 3 }));
    };
    let (r1cs_sat_proof, rx, ry) = {
      let src = match &vars.dev { Some(t) => gpu::VarsSource::Resident(t), None => gpu::VarsSource::Host(&vars.assignment) };
      R1CSProof::prove_gpu(&inst.inst, src, &inputs.assignment, &gens.gens_r1cs_sat, transcript, random_tape,
        ProveHooks { transcript_prefix: &mut prefix, on_rx: if overlap { Some(&mut on_rx) } else { None }, on_ry: if eval_ahead { Some(&mut on_ry) } else { None } })
    };
    let inst_evals = {
      let (Ar, Br, Cr) = match ahead.take() {
        Some((_tx, _ty, job)) => { let v = job.wait_scalars(); (v[0], v[1], v[2]) } // sp_job_wait: 3 x 32 bytes of Montgomery limbs
        None => inst.inst.evaluate_dev(&rx, &ry), // 3 x sp_sparse_evaluate
      };
      Ar.append_to_transcript(b"Ar_claim", transcript);
      Br.append_to_transcript(b"Br_claim", transcript);
      Cr.append_to_transcript(b"Cr_claim", transcript);
      (Ar, Br, Cr)
    };
    // R1CSEvalProof::prove (r1cs.rs:326-349) -> SparseMatPolyEvalProof::prove
    let r1cs_eval_proof = R1CSEvalProof {
      proof: SparseMatPolyEvalProof::prove_gpu(&decomm.decomm.dense, &rx, &ry, &[inst_evals.0, inst_evals.1, inst_evals.2],
                                              &gens.gens_r1cs_eval.gens, transcript, random_tape, early.take()),
    };
    SNARK { r1cs_sat_proof, inst_evals, r1cs_eval_proof }
  }

  /// SNARK::encode (:325-336): R1CSShape::commit -> multi_commit with the dense representation built on the device
  /// (SparseMatPolynomial::multi_sparse_to_dense_rep_dev); the two commitments go through DensePolynomial::commit_inner.
  /// Binding (B): the whole proof in one call on the caller's transcript. `h` are the spz_* handles of the instance, generators and
  /// encoding created once by the same library (spz_instance_new, spz_snark_gens_new, spz_snark_encode).
  #[cfg(feature = "gpu")]
  pub fn prove_coarse(h: &gpu::HostHandles, vars: &VarsAssignment, inputs: &InputsAssignment, transcript: &mut Transcript) -> Self {
    let mut state = gpu::transcript_state(transcript); // the 203 bytes of merlin::Transcript { strobe: Strobe128 { state, pos, pos_begin, cur_flags } }
    let p = unsafe {
      gpu::spz_snark_prove_t(h.ctx, h.inst, h.gens, h.enc, std::ptr::null_mut(), gpu::limbs(&vars.assignment), vars.assignment.len(),
                             gpu::limbs(&inputs.assignment), inputs.assignment.len(), state.as_mut_ptr(), std::ptr::null(), std::ptr::null_mut())
    };
    assert!(!p.is_null(), "spz_snark_prove_t failed");
    gpu::set_transcript_state(transcript, &state);
    bincode::deserialize(&gpu::proof_bytes(p)).unwrap()
  }
}

impl NIZK {
  #[cfg(feature = "gpu")]
  pub fn prove(inst: &Instance, vars: VarsAssignment, input: &InputsAssignment, gens: &NIZKGens, transcript: &mut Transcript) -> Self {
    let mut random_tape = RandomTape::new(b"proof");
    let mut prefix = |t: &mut Transcript| {
      t.append_protocol_name(NIZK::protocol_name());
      t.append_message(b"R1CSShapeDigest", &inst.digest);
    };
    let src = match &vars.dev { Some(t) => gpu::VarsSource::Resident(t), None => gpu::VarsSource::Host(&vars.assignment) };
    let (proof, rx, ry) = R1CSProof::prove_gpu(&inst.inst, src, &input.assignment, &gens.gens_r1cs_sat, transcript, &mut random_tape,
      ProveHooks { transcript_prefix: &mut prefix, on_rx: None, on_ry: None });
    NIZK { r1cs_sat_proof: proof, r: (rx, ry) }
  }
}
// (seed_scalar, Instance::produce_synthetic_r1cs_seeded and the CPU path's prove_with_tape_seed come with rust_shim/seed_hooks.patch, which
// gpu_feature.patch applies on top of; under the gpu feature VarsAssignment additionally carries `dev: None`.)
