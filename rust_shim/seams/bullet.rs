// src/nizk/bullet.rs — BulletReductionProof::prove (:32-132) under `--features gpu`.
// The folded generators G^(k) are never built (no variable-base arithmetic on the prover): with s_0 = [1],
// s_(k+1)[2p] = s_k[p] u^-1, s_(k+1)[2p+1] = s_k[p] u, the vector of :108 is G^(k)[i] = sum_p s_k[p] G[p n_k + i], so every L, R
// (:83-97) is a fixed-base MSM over the ORIGINAL generators, computed by the library in one launch per round
// (sp_ipa_round_lr: lookups + tree + reduction of both rows; the c_L Q + blind_L H tails are added on the calling core).
// The caller (DotProductProofLog::prove, seams/nizk.rs) has opened the argument with sp_ipa_begin_dev — a, b, the generator
// run and Q = r * gens_1.G[0], H = h live behind the handle — and closes it with sp_ipa_finish_commit.
#[cfg(feature = "gpu")]
impl BulletReductionProof {
  /// The round loop of :72-119: (L_vec, R_vec) and blind_fin. `a_hat`, `b_hat`, `g_hat` are read by the caller.
  pub fn prove_rounds_gpu(
    transcript: &mut Transcript,
    ipa: &super::super::gpu::Ipa,
    blind: &Scalar,
    blinds_vec: &[(Scalar, Scalar)],
  ) -> (BulletReductionProof, Scalar) {
    use super::super::gpu;
    let mut L_vec = Vec::with_capacity(blinds_vec.len());
    let mut R_vec = Vec::with_capacity(blinds_vec.len());
    let mut blind_fin = *blind;
    for (blind_L, blind_R) in blinds_vec.iter() {
      let (mut L, mut R) = ([0u8; 32], [0u8; 32]);
      gpu::ok(unsafe { gpu::sp_ipa_round_lr(ipa.0, gpu::limbs1(blind_L), gpu::limbs1(blind_R), L.as_mut_ptr(), R.as_mut_ptr()) });
      let (L, R) = (CompressedGroup::from_slice(&L), CompressedGroup::from_slice(&R));
      transcript.append_point(b"L", &L);
      transcript.append_point(b"R", &R);
      let u = transcript.challenge_scalar(b"u");
      let u_inv = u.invert().unwrap();
      // :105-109: records (u, u^-1); the next round's launch (or sp_ipa_finish_commit) applies the fold on the way
      gpu::ok(unsafe { gpu::sp_ipa_round_fold(ipa.0, gpu::limbs1(&u), gpu::limbs1(&u_inv)) });
      blind_fin = blind_fin + blind_L * u * u + blind_R * u_inv * u_inv;
      L_vec.push(L);
      R_vec.push(R);
    }
    (BulletReductionProof { L_vec, R_vec }, blind_fin)
  }
}
