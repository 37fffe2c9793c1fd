// src/dense_mlpoly.rs — bodies swapped under `--features gpu` (same pattern as the `multicore` pair at :148-177).
// DensePolynomial gains `#[cfg(feature = "gpu")] dev: Option<gpu::Table>`: tables produced on the device (eq tables,
// Az/Bz/Cz, bound polynomials, product-circuit layers) stay there; `Z` is only materialised by `download()` when host code
// indexes it.
use super::gpu;

impl DensePolynomial {
  /// DensePolynomial::commit_inner (:164-177): L row commitments in one call.
  #[cfg(feature = "gpu")]
  fn commit_inner(&self, blinds: &[Scalar], gens: &MultiCommitGens) -> PolyCommitment {
    let L_size = blinds.len();
    let R_size = self.len() / L_size;
    assert_eq!(L_size * R_size, self.len());
    assert_eq!(gens.n, R_size);
    let mut out = vec![0u8; 32 * L_size];
    let g = gpu::gens_for(gens); // points G[0..n), h at index n
    match &self.dev {
      Some(t) => gpu::ok(unsafe {
        gpu::sp_commit_rows_dev(gpu::ctx(), g, 0, gens.n, t.0, 0, L_size, R_size, gpu::limbs(blinds), out.as_mut_ptr())
      }),
      None => gpu::ok(unsafe {
        gpu::sp_commit_rows(gpu::ctx(), g, 0, gens.n, gpu::limbs(&self.Z), L_size, R_size, gpu::limbs(blinds), out.as_mut_ptr())
      }),
    }
    PolyCommitment {
      C: out.chunks_exact(32).map(|c| CompressedGroup::from_slice(c)).collect(),
    }
  }

  /// DensePolynomial::bound (:206-213): LZ = L * Z.
  #[cfg(feature = "gpu")]
  pub fn bound(&self, L: &[Scalar]) -> Vec<Scalar> {
    let (left_num_vars, right_num_vars) = EqPolynomial::compute_factored_lens(self.get_num_vars());
    let (L_size, R_size) = (left_num_vars.pow2(), right_num_vars.pow2());
    assert_eq!(L.len(), L_size);
    let owned;
    let t = match &self.dev {
      Some(t) => t,
      None => {
        owned = gpu::Table::upload(&self.Z);
        &owned
      }
    };
    let mut out = vec![Scalar::zero(); R_size];
    gpu::ok(unsafe { gpu::sp_vecmat(gpu::ctx(), gpu::limbs(L), L_size, t.0, gpu::limbs_mut(&mut out)) });
    out
  }

  /// DensePolynomial::bound_poly_var_top (:215-223).
  #[cfg(feature = "gpu")]
  pub fn bound_poly_var_top(&mut self, r: &Scalar) {
    let t = self.dev.as_ref().expect("device-resident polynomial");
    let tabs = [t.0];
    gpu::ok(unsafe { gpu::sp_table_bind_top(gpu::ctx(), tabs.as_ptr(), 1, gpu::limbs1(r)) });
    self.num_vars -= 1;
    self.len /= 2;
  }

  /// DensePolynomial::evaluate (:236-242): <Z, chi(r)> with chi generated on the device.
  #[cfg(feature = "gpu")]
  pub fn evaluate(&self, r: &[Scalar]) -> Scalar {
    assert_eq!(r.len(), self.get_num_vars());
    let owned;
    let t = match &self.dev {
      Some(t) => t,
      None => {
        owned = gpu::Table::upload(&self.Z);
        &owned
      }
    };
    let mut out = Scalar::zero();
    gpu::ok(unsafe { gpu::sp_evaluate(gpu::ctx(), t.0, gpu::limbs(r), r.len(), &mut out as *mut Scalar as *mut u64) });
    out
  }
}

impl EqPolynomial {
  /// EqPolynomial::evals (:68-84) as a device table (r[0] <-> most significant index bit, as in the reference).
  #[cfg(feature = "gpu")]
  pub fn evals_dev(&self) -> gpu::Table {
    gpu::Table::eq(&self.r)
  }
}
