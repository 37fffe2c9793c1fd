// src/product_tree.rs — under `--features gpu` a ProductCircuit is ONE device store of 2n scalars: the n hashed leaves in
// [0, n), layer k (n / 2^k elements, left half then right half) at offset 2n - 2n / 2^k. left_vec[k] / right_vec[k] are views
// (gpu::Table::view -> sp_table_view) of its halves. C++ rendering: spartan_amd/host/spark.inc (ProductCircuit,
// product_circuits_evaluate, product_batched_prove).
use super::gpu::{self, sp_table};

#[cfg(feature = "gpu")]
impl ProductCircuit {
  /// ProductCircuit::new (:36-56) for every circuit of one size at once: one launch per layer for all of them, the short
  /// layers in a single launch (sp_product_tree_many). `stores[i]` holds circuit i's leaves in [0, n).
  /// layers_done: 1 when layer 1 is already in the stores (sp_hash_layer_first), else 0.
  pub fn new_many(stores: Vec<gpu::Table>, n: usize, layers_done: usize) -> Vec<ProductCircuit> {
    let hs: Vec<*mut sp_table> = stores.iter().map(|t| t.0).collect();
    gpu::ok(unsafe { gpu::sp_product_tree_many_from(gpu::ctx(), hs.as_ptr(), hs.len(), n, layers_done) });
    stores.into_iter().map(|store| {
      let num_layers = n.log_2();
      let (mut left_vec, mut right_vec) = (Vec::new(), Vec::new());
      for k in 0..num_layers {
        let (off, len) = (2 * n - 2 * (n >> k), n >> k);
        left_vec.push(DensePolynomial::from_dev(store.view(off, len / 2)));
        right_vec.push(DensePolynomial::from_dev(store.view(off + len / 2, len / 2)));
      }
      ProductCircuit { left_vec, right_vec, store }
    }).collect()
  }
}

/// ProductCircuit::evaluate (:58-63) of several circuits in one round trip: the product of the two roots of each.
#[cfg(feature = "gpu")]
pub fn product_circuits_evaluate(ps: &[&ProductCircuit]) -> Vec<Scalar> {
  let tabs: Vec<*mut sp_table> = ps.iter().map(|p| p.store.0).collect();
  let offs: Vec<usize> = ps.iter().map(|p| { let n = p.store.len() / 2; 2 * n - 4 }).collect(); // the last layer: two elements
  let mut lr = vec![Scalar::zero(); 2 * ps.len()];
  gpu::ok(unsafe { gpu::sp_table_gather(gpu::ctx(), tabs.as_ptr(), offs.as_ptr(), tabs.len(), 2, gpu::limbs_mut(&mut lr)) });
  (0..ps.len()).map(|i| lr[2 * i] * lr[2 * i + 1]).collect()
}

#[cfg(feature = "gpu")]
impl DotProductCircuit {
  /// DotProductCircuit::evaluate (:84-88)
  pub fn evaluate(&self) -> Scalar {
    let mut out = Scalar::zero();
    let dev = |p: &DensePolynomial| p.dev.as_ref().expect("device-resident polynomial").0;
    gpu::ok(unsafe { gpu::sp_dot3(gpu::ctx(), dev(&self.left), dev(&self.right), dev(&self.weight), 0, self.left.len(), &mut out as *mut Scalar as *mut u64) });
    out
  }
}

#[cfg(feature = "gpu")]
impl ProductCircuitEvalProofBatched {
  /// ProductCircuitEvalProofBatched::prove (:259-383). `dotp_evals`: DotProductCircuit::evaluate of each dot-product circuit when
  /// the caller already has them (ProductLayerProof::prove computes exactly these as its claim_eval_dotp_left/right).
  pub fn prove_gpu(
    prod_circuit_vec: &mut [&mut ProductCircuit],
    dotp_circuit_vec: &mut [&mut DotProductCircuit],
    dotp_evals: Option<&[Scalar]>,
    roots: Option<&[Scalar]>, // ProductCircuit::evaluate of each circuit when the caller already has them (ProductLayerProof::prove has just absorbed them)
    transcript: &mut Transcript,
  ) -> (Self, Vec<Scalar>) {
    assert!(!prod_circuit_vec.is_empty());
    let mut claims_dotp_final = (Vec::new(), Vec::new(), Vec::new());
    let mut proof_layers: Vec<LayerProofBatched> = Vec::new();
    let num_layers = prod_circuit_vec[0].left_vec.len();
    let mut claims_to_verify = match roots {
      Some(v) => { assert_eq!(v.len(), prod_circuit_vec.len()); v.to_vec() }
      None => product_circuits_evaluate(&prod_circuit_vec.iter().map(|p| &**p).collect::<Vec<_>>()),
    };
    let mut rand: Vec<Scalar> = Vec::new();
    for layer_id in (0..num_layers).rev() {
      let len = prod_circuit_vec[0].left_vec[layer_id].len() + prod_circuit_vec[0].right_vec[layer_id].len();
      // EqPolynomial::new(rand).evals() (:279): on the device; the one-entry table of the first layer is uploaded
      let mut poly_C_par = DensePolynomial::from_dev(if rand.is_empty() { gpu::Table::upload(&[Scalar::one()]) } else { EqPolynomial::new(rand.clone()).evals_dev() });
      assert_eq!(poly_C_par.len(), len / 2);
      let num_rounds_prod = poly_C_par.len().log_2();
      let mut poly_A_batched_par: Vec<&mut DensePolynomial> = Vec::new();
      let mut poly_B_batched_par: Vec<&mut DensePolynomial> = Vec::new();
      for prod_circuit in prod_circuit_vec.iter_mut() {
        let (l, r) = (&mut prod_circuit.left_vec[layer_id] as *mut DensePolynomial, &mut prod_circuit.right_vec[layer_id] as *mut DensePolynomial);
        poly_A_batched_par.push(unsafe { &mut *l });
        poly_B_batched_par.push(unsafe { &mut *r });
      }
      let (mut poly_A_batched_seq, mut poly_B_batched_seq, mut poly_C_batched_seq): (Vec<&mut DensePolynomial>, Vec<&mut DensePolynomial>, Vec<&mut DensePolynomial>) =
        (Vec::new(), Vec::new(), Vec::new());
      if layer_id == 0 && !dotp_circuit_vec.is_empty() {
        // add additional claims (:303-314)
        for (k, item) in dotp_circuit_vec.iter_mut().enumerate() {
          claims_to_verify.push(match dotp_evals { Some(v) => v[k], None => item.evaluate() });
          assert_eq!(len / 2, item.left.len());
          poly_A_batched_seq.push(&mut item.left);
          poly_B_batched_seq.push(&mut item.right);
          poly_C_batched_seq.push(&mut item.weight);
        }
      }
      let coeff_vec = transcript.challenge_vector(b"rand_coeffs_next_layer", claims_to_verify.len());
      let claim = (0..claims_to_verify.len()).map(|i| claims_to_verify[i] * coeff_vec[i]).sum();
      let (proof, rand_prod, claims_prod, claims_dotp) = SumcheckInstanceProof::prove_cubic_batched_gpu(
        &claim, num_rounds_prod,
        (&mut poly_A_batched_par, &mut poly_B_batched_par, &mut poly_C_par),
        (&mut poly_A_batched_seq, &mut poly_B_batched_seq, &mut poly_C_batched_seq),
        &coeff_vec, transcript,
        if rand.is_empty() { None } else { Some(&rand[..]) }); // poly_C_par is EqPolynomial::new(rand).evals() (:279): the factored rounds
      let (claims_prod_left, claims_prod_right, _claims_eq) = claims_prod;
      for i in 0..prod_circuit_vec.len() {
        transcript.append_scalar(b"claim_prod_left", &claims_prod_left[i]);
        transcript.append_scalar(b"claim_prod_right", &claims_prod_right[i]);
      }
      if layer_id == 0 && !dotp_circuit_vec.is_empty() {
        let (claims_dotp_left, claims_dotp_right, claims_dotp_weight) = claims_dotp;
        for i in 0..dotp_circuit_vec.len() {
          transcript.append_scalar(b"claim_dotp_left", &claims_dotp_left[i]);
          transcript.append_scalar(b"claim_dotp_right", &claims_dotp_right[i]);
          transcript.append_scalar(b"claim_dotp_weight", &claims_dotp_weight[i]);
        }
        claims_dotp_final = (claims_dotp_left, claims_dotp_right, claims_dotp_weight);
      }
      // produce a random challenge to condense two claims into a single claim
      let r_layer = transcript.challenge_scalar(b"challenge_r_layer");
      claims_to_verify = (0..prod_circuit_vec.len()).map(|i| claims_prod_left[i] + r_layer * (claims_prod_right[i] - claims_prod_left[i])).collect();
      let mut ext = vec![r_layer];
      ext.extend(rand_prod);
      rand = ext;
      proof_layers.push(LayerProofBatched { proof, claims_prod_left, claims_prod_right });
    }
    (ProductCircuitEvalProofBatched { proof: proof_layers, claims_dotp: claims_dotp_final }, rand)
  }
}
