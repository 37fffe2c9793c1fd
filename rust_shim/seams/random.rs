// src/random.rs — add next to RandomTape::new (random.rs:11-18). The seeded constructor is the determinism hook of the
// byte-parity contract: with it `SNARK::prove` / `NIZK::prove` reproduce tests/golden/proof_digests.json.
// A seed fixes every blind of the proof: outside tests it must be secret, >= 256 bits of entropy, and used once.
impl RandomTape {
  pub fn new_with_seed(name: &'static [u8], seed: &Scalar) -> Self {
    let mut tape = Transcript::new(name);
    tape.append_scalar(b"init_randomness", seed); // identical to :15 with the OsRng draw replaced
    Self { tape }
  }
}

/// seed scalar used by tests/, bench.py and tests/golden: from_bytes_wide(SHAKE256(domain || LE64(seed))[..64])
pub fn seed_scalar(domain: &[u8], seed: u64) -> Scalar {
  use sha3::digest::{ExtendableOutput, Input, XofReader};
  let mut shake = sha3::Shake256::default();
  shake.input(domain);
  shake.input(seed.to_le_bytes());
  let mut buf = [0u8; 64];
  shake.xof_result().read(&mut buf);
  Scalar::from_bytes_wide(&buf)
}
