// src/nizk/bullet.rs — BulletReductionProof::prove (:32-132) under `--features gpu`.
// The folded generators G^(k) are never built (no variable-base arithmetic on the prover): with s_0 = [1],
// s_(k+1)[2p] = s_k[p] u^-1, s_(k+1)[2p+1] = s_k[p] u, the vector of :108 is G^(k)[i] = sum_p s_k[p] G[p n_k + i], so every L, R
// (:83-97) is a fixed-base MSM over the ORIGINAL generators. The caller (DotProductProofLog::prove, nizk/mod.rs:440-525)
// passes generator handles instead of G_vec/Q/H: gens_n (G and h) and gens_1 (whose G[0] scaled by r is Q, :479-480).
#[cfg(feature = "gpu")]
impl BulletReductionProof {
  pub fn prove_gpu(
    transcript: &mut Transcript,
    gens_n: &MultiCommitGens, // G_vec = gens_n.G, H = gens_n.h
    gens_1: &MultiCommitGens, // Q = q_scale * gens_1.G[0]
    q_scale: &Scalar,
    a_vec: &[Scalar],
    b_vec: &[Scalar],
    blind: &Scalar,
    blinds_vec: &[(Scalar, Scalar)],
  ) -> (BulletReductionProof, Scalar, Scalar, CompressedGroup, Scalar) {
    use super::super::gpu;
    let n = a_vec.len();
    assert!(n.is_power_of_two());
    let lg_n = n.log_2();
    assert_eq!(gens_n.n, n);
    assert_eq!(b_vec.len(), n);
    assert_eq!(blinds_vec.len(), lg_n);
    // one device list holds gens_n.G (n points), gens_n.h, gens_1.G[0]: the SHAKE stream of DotProductProofGens::new
    // (nizk/mod.rs:415-418) is G[0..n), Q-base at n, h at n+1 — here uploaded as [G..., h] and [Q-base, h] share `h`.
    let g = gpu::gens_for_dot_product(gens_n, gens_1); // points: G[0..n), gens_1.G[0] at n, h at n+1
    let mut ipa = std::ptr::null_mut();
    gpu::ok(unsafe {
      gpu::sp_ipa_begin(gpu::ctx(), g, 0, n, n, n + 1, gpu::limbs1(q_scale), gpu::limbs(a_vec), gpu::limbs(b_vec), &mut ipa)
    });
    let mut L_vec = Vec::with_capacity(lg_n);
    let mut R_vec = Vec::with_capacity(lg_n);
    let mut blind_final = *blind;
    for (blind_L, blind_R) in blinds_vec.iter() {
      let (mut L, mut R) = ([0u8; 32], [0u8; 32]);
      gpu::ok(unsafe { gpu::sp_ipa_round_lr(ipa, gpu::limbs1(blind_L), gpu::limbs1(blind_R), L.as_mut_ptr(), R.as_mut_ptr()) });
      let (L, R) = (CompressedGroup::from_slice(&L), CompressedGroup::from_slice(&R));
      transcript.append_point(b"L", &L);
      transcript.append_point(b"R", &R);
      let u = transcript.challenge_scalar(b"u");
      let u_inv = u.invert().unwrap();
      gpu::ok(unsafe { gpu::sp_ipa_round_fold(ipa, gpu::limbs1(&u), gpu::limbs1(&u_inv)) });
      blind_final = blind_final + blind_L * u * u + blind_R * u_inv * u_inv;
      L_vec.push(L);
      R_vec.push(R);
    }
    let (mut a_hat, mut b_hat) = (Scalar::zero(), Scalar::zero());
    let mut g_hat = [0u8; 32];
    gpu::ok(unsafe {
      gpu::sp_ipa_finish(ipa, &mut a_hat as *mut Scalar as *mut u64, &mut b_hat as *mut Scalar as *mut u64, g_hat.as_mut_ptr())
    });
    unsafe { gpu::sp_ipa_free(ipa) };
    // Gamma_hat (:121-122) is only asserted against by the caller's verifier-side algebra; the prover needs a_hat, b_hat,
    // g_hat and blind_final (nizk/mod.rs:489-512)
    (BulletReductionProof { L_vec, R_vec }, a_hat, b_hat, CompressedGroup::from_slice(&g_hat), blind_final)
  }
}
