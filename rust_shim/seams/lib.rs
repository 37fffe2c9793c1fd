// src/lib.rs — seeded twins of SNARK::prove (:339-420) and NIZK::prove (:501-546) for the parity tests: the only change is
// the RandomTape constructor (:356 / :516). Production code keeps calling `prove` (OS entropy).
impl SNARK {
  pub fn prove_with_tape_seed(
    inst: &Instance,
    comm: &ComputationCommitment,
    decomm: &ComputationDecommitment,
    vars: VarsAssignment,
    inputs: &InputsAssignment,
    gens: &SNARKGens,
    transcript: &mut Transcript,
    tape_seed: &Scalar,
  ) -> Self {
    let mut random_tape = RandomTape::new_with_seed(b"proof", tape_seed);
    Self::prove_with_tape(inst, comm, decomm, vars, inputs, gens, transcript, &mut random_tape) // body of :358-419, tape passed in
  }
}
pub use random::seed_scalar;
impl Instance {
  pub fn produce_synthetic_r1cs_seeded(num_cons: usize, num_vars: usize, num_inputs: usize, seed: u64) -> (Instance, VarsAssignment, InputsAssignment) {
    let (inst, vars, inputs) = R1CSInstance::produce_synthetic_r1cs_seeded(num_cons, num_vars, num_inputs, seed);
    let digest = inst.get_digest();
    (Instance { inst, digest }, VarsAssignment { assignment: vars }, InputsAssignment { assignment: inputs })
  }
}
