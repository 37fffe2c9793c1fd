// src/commitments.rs — under `--features gpu` a MultiCommitGens also knows where its points live on the device.
//
// Every generator set of the prover is a prefix (or a split of a prefix) of ONE SHAKE256 stream per label
// (MultiCommitGens::new(n, label), :15-33; R1CSGens / SparseMatPolyCommitmentGens build all their sets from "gens_r1cs_sat" /
// "gens_r1cs_eval", r1csproof.rs:48-73, sparse_mlpoly.rs:291-318). The device holds one sp_gens per label — the longest
// prefix asked for so far, with its fixed-base window tables — and a MultiCommitGens carries stream indices into it. That is
// what lets one C-ABI call commit under generators of gens_n AND gens_1 at once (the rounds of the ZK sum-checks) and is the
// same representation the C++ host driver uses (spartan_amd/host/libspartan.hpp: MultiCommitGens { g, G, h }).
#[cfg(feature = "gpu")]
#[derive(Clone)]
pub struct DevGens {
  pub g: *const gpu::sp_gens, // the label's stream on the device (kept for the life of the process by gpu::stream_gens)
  pub G: Vec<u32>,            // stream indices of G[0..n)
  pub h: u32,                 // stream index of h
}

impl MultiCommitGens {
  /// MultiCommitGens::new (:15-33): the SHAKE256 stream is drawn here exactly as in the reference; `from_uniform_bytes` and
  /// the window tables are the library's (sp_gens_from_uniform inside gpu::stream_gens). `dev` indexes the stream.
  #[cfg(feature = "gpu")]
  pub fn new(n: usize, label: &[u8]) -> Self {
    let (g, compressed) = gpu::stream_gens(label, n + 1); // >= n + 1 points of the label's stream, uploaded once per process
    let pts: Vec<GroupElement> = (0..n + 1)
      .map(|i| CompressedGroup::from_slice(&compressed[32 * i..32 * i + 32]).decompress().unwrap())
      .collect();
    MultiCommitGens {
      n,
      G: pts[..n].to_vec(),
      h: pts[n],
      dev: DevGens { g, G: (0..n as u32).collect(), h: n as u32 },
    }
  }

  /// split_at (:51-67): the index lists split with the points.
  #[cfg(feature = "gpu")]
  pub fn split_at(&self, mid: usize) -> (MultiCommitGens, MultiCommitGens) {
    let (G1, G2) = self.G.split_at(mid);
    let (i1, i2) = self.dev.G.split_at(mid);
    (
      MultiCommitGens { n: G1.len(), G: G1.to_vec(), h: self.h, dev: DevGens { g: self.dev.g, G: i1.to_vec(), h: self.dev.h } },
      MultiCommitGens { n: G2.len(), G: G2.to_vec(), h: self.h, dev: DevGens { g: self.dev.g, G: i2.to_vec(), h: self.dev.h } },
    )
  }
}

/// The few-term commitments of the Sigma protocols (Scalar::commit :73-78, [Scalar]::commit :87-92 with n <= 8, UniPoly::commit):
/// rows of scalars over an explicit list of stream indices -> one CompressedGroup per row. On the calling thread's core through
/// the library's host-side engine (sp_host_commit_small: a 2..5-term commitment is a chain of ~100 dependent point additions
/// the transcript waits for — ~15 us there, ~60 us + a round trip on a lone wavefront), or on the device with
/// option commit.small_device = 1 (sp_msm_indexed). Same bytes either way. `addend[r]`: a point computed ahead (sp_host_zk_ahead_*).
#[cfg(feature = "gpu")]
pub fn commit_small(
  g: *const gpu::sp_gens,
  idx: &[u32],
  scalars: &[Scalar], // rows x idx.len(), row-major
  rows: usize,
  addend: Option<&[*const gpu::sp_host_point]>,
) -> Vec<CompressedGroup> {
  assert_eq!(scalars.len(), rows * idx.len());
  let mut out = vec![0u8; 32 * rows];
  if idx.len() <= 8 && gpu::small_msm_on_host() {
    let ap = addend.map_or(std::ptr::null(), |a| a.as_ptr());
    gpu::ok(unsafe { gpu::sp_host_commit_small(g, idx.as_ptr(), idx.len(), gpu::limbs(scalars), rows, ap, out.as_mut_ptr()) });
  } else {
    assert!(addend.is_none());
    gpu::ok(unsafe { gpu::sp_msm_indexed(gpu::ctx(), g, idx.as_ptr(), idx.len(), gpu::limbs(scalars), rows, out.as_mut_ptr()) });
  }
  out.chunks_exact(32).map(CompressedGroup::from_slice).collect()
}

impl Commitments for Scalar {
  /// Scalar::commit (:73-78) — the compressed form is what every caller on the prover path wants.
  #[cfg(feature = "gpu")]
  fn commit_compressed(&self, blind: &Scalar, gens_n: &MultiCommitGens) -> CompressedGroup {
    assert_eq!(gens_n.n, 1);
    commit_small(gens_n.dev.g, &[gens_n.dev.G[0], gens_n.dev.h], &[*self, *blind], 1, None)[0]
  }
}
