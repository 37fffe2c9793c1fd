// src/sumcheck.rs — SumcheckInstanceProof::prove_cubic_batched (:254-424) under `--features gpu`.
// The round body (:287-357 evaluations, :379-393 binds) is one C-ABI call per round, exactly as the C++ host driver issues
// it (spartan_amd/host/spark.inc: prove_cubic_batched): sp_sumcheck_eval_batched once, then sp_sumcheck_bind_eval_batched
// per round (bind at r_j fused with the next round's evaluations), and sp_table_bind_top_heads for the last round, which
// also returns the final claims (:395-419). Every transcript operation stays where the reference has it.
// comb_func is always the cubic product on this path (product_tree.rs:316-318), which is what the kernels compute.
#[cfg(feature = "gpu")]
impl SumcheckInstanceProof {
  pub fn prove_cubic_batched_gpu(
    claim: &Scalar,
    num_rounds: usize,
    poly_vec_par: (&mut Vec<&mut DensePolynomial>, &mut Vec<&mut DensePolynomial>, &mut DensePolynomial),
    poly_vec_seq: (&mut Vec<&mut DensePolynomial>, &mut Vec<&mut DensePolynomial>, &mut Vec<&mut DensePolynomial>),
    coeffs: &[Scalar],
    transcript: &mut Transcript,
  ) -> (Self, Vec<Scalar>, (Vec<Scalar>, Vec<Scalar>, Scalar), (Vec<Scalar>, Vec<Scalar>, Vec<Scalar>)) {
    use super::gpu::{self, sp_table};
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
    // every table once, for the last round: A_i, B_i interleaved, then the distinct C tables
    let mut all: Vec<*mut sp_table> = Vec::with_capacity(2 * ni + 1 + ns);
    for i in 0..ni {
      all.push(A[i]);
      all.push(B[i]);
    }
    all.push(dev(poly_C_par));
    all.extend(poly_C_vec_seq.iter().map(|p| dev(p)));

    let mut e = *claim;
    let mut r: Vec<Scalar> = Vec::new();
    let mut cubic_polys: Vec<CompressedUniPoly> = Vec::new();
    let mut ev = vec![Scalar::zero(); 3 * ni]; // (eval_point_0, eval_point_2, eval_point_3) per instance
    let mut heads = vec![Scalar::zero(); all.len()];
    let mut have_heads = false;
    if num_rounds > 0 {
      gpu::ok(unsafe { gpu::sp_sumcheck_eval_batched(gpu::ctx(), A.as_ptr(), B.as_ptr(), C.as_ptr(), ni, gpu::limbs_mut(&mut ev)) });
    }
    for _j in 0..num_rounds {
      let evals_combined_0: Scalar = (0..ni).map(|i| ev[3 * i] * coeffs[i]).sum();
      let evals_combined_2: Scalar = (0..ni).map(|i| ev[3 * i + 1] * coeffs[i]).sum();
      let evals_combined_3: Scalar = (0..ni).map(|i| ev[3 * i + 2] * coeffs[i]).sum();
      let evals = vec![evals_combined_0, e - evals_combined_0, evals_combined_2, evals_combined_3];
      let poly = UniPoly::from_evals(&evals);
      poly.append_to_transcript(b"poly", transcript);
      let r_j = transcript.challenge_scalar(b"challenge_nextround");
      r.push(r_j);
      let len = unsafe { gpu::sp_table_len(A[0]) };
      if len >= 4 {
        gpu::ok(unsafe {
          gpu::sp_sumcheck_bind_eval_batched(gpu::ctx(), A.as_ptr(), B.as_ptr(), C.as_ptr(), ni, gpu::limbs1(&r_j), gpu::limbs_mut(&mut ev))
        });
      } else {
        gpu::ok(unsafe { gpu::sp_table_bind_top_heads(gpu::ctx(), all.as_ptr(), all.len(), gpu::limbs1(&r_j), gpu::limbs_mut(&mut heads)) });
        have_heads = true;
      }
      e = poly.evaluate(&r_j);
      cubic_polys.push(poly.compress());
    }
    if !have_heads {
      gpu::ok(unsafe { gpu::sp_table_heads(gpu::ctx(), all.as_ptr(), all.len(), gpu::limbs_mut(&mut heads)) });
    }
    // host-side bookkeeping of the bound polynomials (their tables were halved num_rounds times on the device)
    for p in poly_A_vec_par.iter_mut().chain(poly_B_vec_par.iter_mut()).chain(poly_A_vec_seq.iter_mut())
      .chain(poly_B_vec_seq.iter_mut()).chain(poly_C_vec_seq.iter_mut()) {
      p.num_vars -= num_rounds;
      p.len >>= num_rounds;
    }
    poly_C_par.num_vars -= num_rounds;
    poly_C_par.len >>= num_rounds;

    let claims_prod = (
      (0..np).map(|k| heads[2 * k]).collect(),
      (0..np).map(|k| heads[2 * k + 1]).collect(),
      heads[2 * ni],
    );
    let claims_dotp = (
      (0..ns).map(|k| heads[2 * (np + k)]).collect(),
      (0..ns).map(|k| heads[2 * (np + k) + 1]).collect(),
      (0..ns).map(|k| heads[2 * ni + 1 + k]).collect(),
    );
    (SumcheckInstanceProof::new(cubic_polys), r, claims_prod, claims_dotp)
  }
}

// The C++ driver additionally advances TWO rounds per call while the tables are short (<= 512 entries, the latency-bound
// tail: ~290 of the 361 rounds of a 2^20 proof): sp_sumcheck_eval_coeffs_batched / sp_sumcheck_bind2_eval_batched return,
// next to the evaluations of a round, the coefficients (M0, M3, T1, T2 at t = 0, 2, 3) of the cubic in the NEXT challenge
// that the following round's evaluations are; the caller derives r_j, evaluates that cubic on the host, derives r_{j+1},
// and only then goes back to the device with both challenges (spartan_amd/host/spark.inc: prove_cubic_batched, evals_from_coeffs).
// Same field values, same transcript operations, half the round trips. Once the tables have at most 8 entries the call's
// `_tables_` variant (sp_sumcheck_bind2_eval_tables_batched) also returns the tables themselves and the driver runs the last
// <= 3 rounds with the reference's own loop body on the CPU (spark.inc: cubic_tail_rounds). The one-round form above is the same proof.

// ZKSumcheckInstanceProof::prove_quad (:428-586) and ::prove_cubic_with_additive_term (:588-776): the round body becomes
//   round 0:  sp_sumcheck_eval(kind, tabs) -> evals -> UniPoly -> comm_poly  (reference code: poly.commit(...))
//   round j:  sp_sumcheck_bind_eval_start(kind, tabs, r_j)        // bound_poly_var_top on every table (:485-486 / :673-676)
//                                                                 // fused with the next round's evaluations (:460-469 / :624-652)
//             ... the round's Sigma-protocol commitments, UNCHANGED reference code on the CPU (comm_eval, DotProductProof::prove's
//                 delta, Cy, beta: 2..5-term commitments under gens_1 / gens_n — dalek's multiscalar_mul on a handful of points) ...
//             sp_sumcheck_bind_eval_collect(evals_next)           // the device finished long ago
//             comm_poly of round j+1 from evals_next              // reference code
// kind 0 = A*B (prove_quad), kind 2 = A*(B*C - D) (prove_cubic_with_additive_term); tabs in that order.
// Why the few-term commitments stay on the CPU: each is a chain of ~100 dependent point additions plus one inverse square
// root that the transcript waits for; a host core finishes it in ~15 us, a lone wavefront in ~60 us plus the round trip
// (measured: DESIGN.md section 4). The C++ driver does exactly this with its own window tables for those generators
// (spartan_amd/host/small_msm.cc); sp_sumcheck_bind_eval_commit / sp_msm_indexed keep the all-device variant available.
// The complete C++ rendering of both provers, line-by-line against sumcheck.rs, is spartan_amd/host/prover.cc (zk_sumcheck_prove).
