// src/r1cs.rs — seeded twin of R1CSShape::produce_synthetic_r1cs (r1cs.rs:160-238): the OsRng of :169 becomes a SHAKE256
// stream keyed by ("spartan-synthetic-r1cs" || LE64(seed)); everything after the draw of Z is the reference's code,
// unchanged. Z[i] = Scalar::from_bytes_wide(next 64 bytes) — what Scalar::random does with 64 random bytes
// (scalar/ristretto255.rs:374-380).
impl R1CSShape {
  pub fn produce_synthetic_r1cs_seeded(
    num_cons: usize,
    num_vars: usize,
    num_inputs: usize,
    seed: u64,
  ) -> (R1CSShape, Vec<Scalar>, Vec<Scalar>) {
    use sha3::digest::{ExtendableOutput, Input, XofReader};
    assert_eq!((num_cons.log_2()).pow2(), num_cons);
    assert_eq!((num_vars.log_2()).pow2(), num_vars);
    assert!(num_inputs < num_vars);
    let size_z = num_vars + num_inputs + 1;
    let mut shake = sha3::Shake256::default();
    shake.input(b"spartan-synthetic-r1cs");
    shake.input(seed.to_le_bytes());
    let mut xof = shake.xof_result();
    let Z = {
      let mut Z: Vec<Scalar> = (0..size_z)
        .map(|_i| {
          let mut buf = [0u8; 64];
          xof.read(&mut buf);
          Scalar::from_bytes_wide(&buf)
        })
        .collect();
      Z[num_vars] = Scalar::one(); // set the constant term to 1 (r1cs.rs:187)
      Z
    };
    // ---- from here on: r1cs.rs:190-238 verbatim (A, B, C with one entry per row; C_val = Z_A * Z_B / Z_C) ----
    let mut A: Vec<SparseMatEntry> = Vec::new();
    let mut B: Vec<SparseMatEntry> = Vec::new();
    let mut C: Vec<SparseMatEntry> = Vec::new();
    let one = Scalar::one();
    for i in 0..num_cons {
      let A_idx = i % size_z;
      let B_idx = (i + 2) % size_z;
      A.push(SparseMatEntry::new(i, A_idx, one));
      B.push(SparseMatEntry::new(i, B_idx, one));
      let AB_val = Z[A_idx] * Z[B_idx];
      let C_idx = (i + 3) % size_z;
      let C_val = Z[C_idx];
      if C_val == Scalar::zero() {
        C.push(SparseMatEntry::new(i, num_vars, AB_val));
      } else {
        C.push(SparseMatEntry::new(i, C_idx, AB_val * C_val.invert().unwrap()));
      }
    }
    let num_poly_vars_x = num_cons.log_2();
    let num_poly_vars_y = (2 * num_vars).log_2();
    let poly_A = SparseMatPolynomial::new(num_poly_vars_x, num_poly_vars_y, A);
    let poly_B = SparseMatPolynomial::new(num_poly_vars_x, num_poly_vars_y, B);
    let poly_C = SparseMatPolynomial::new(num_poly_vars_x, num_poly_vars_y, C);
    let inst = R1CSShape { num_cons, num_vars, num_inputs, A: poly_A, B: poly_B, C: poly_C };
    assert!(inst.is_sat(&Z[..num_vars], &Z[num_vars + 1..]));
    (inst, Z[..num_vars].to_vec(), Z[num_vars + 1..].to_vec())
  }
}
