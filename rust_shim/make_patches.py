#!/usr/bin/env python3
"""Builds rust_shim/seed_hooks.patch and rust_shim/gpu_feature.patch from a pristine microsoft/Spartan tree (default /root/reference).

  seed_hooks.patch   the two determinism hooks the byte-parity contract needs on the CPU path of REAL libspartan (seeded RandomTape,
                     seeded produce_synthetic_r1cs) + examples/golden_digest.rs, the driver of
                     scripts/compare_with_libspartan.sh. Uses no crate outside the reference's Cargo.toml (sha3 for SHA3-256, manual hex).
  gpu_feature.patch  (applies on top of seed_hooks.patch) the `gpu` cargo feature: Cargo.toml, build.rs, src/gpu.rs (= the generated binding),
                     the seam bodies as src/gpu_seams/*.rs pulled into their modules with `include!`, and `#[cfg(not(feature = "gpu"))]`
                     on every reference function a seam redefines.

Both are produced by anchored edits of the reference's own files (every anchor is asserted) and `diff -ruN`; tests/test_rust_patches.py runs
`git apply --check` on a fresh copy of the reference. Neither has been COMPILED: no Rust toolchain exists here.
usage: python rust_shim/make_patches.py [reference_dir]"""
import os, re, shutil, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
KEEP = ("src", "examples", "profiler", "benches", "Cargo.toml", "build.rs")


def copy_ref(dst):
    os.makedirs(dst)
    for k in KEEP:
        p = os.path.join(REF, k)
        if os.path.isdir(p):
            shutil.copytree(p, os.path.join(dst, k))
        elif os.path.exists(p):
            shutil.copy(p, os.path.join(dst, k))


def edit(root, rel, pairs):
    p = os.path.join(root, rel)
    s = open(p).read()
    for a, b in pairs:
        assert s.count(a) == 1, (rel, a[:70], s.count(a))
        s = s.replace(a, b)
    open(p, "w").write(s)


def write(root, rel, text):
    p = os.path.join(root, rel)
    os.makedirs(os.path.dirname(p), exist_ok=True)
    open(p, "w").write(text)


def diff(a, b, out):
    r = subprocess.run(["diff", "-ruN", os.path.basename(a), os.path.basename(b)], cwd=os.path.dirname(a), capture_output=True, text=True)
    assert r.returncode in (0, 1), r.stderr
    txt = re.sub(r"^(---|\+\+\+) (%s|%s)/(\S+).*$" % (os.path.basename(a), os.path.basename(b)),
                 lambda m: "%s %s/%s" % (m.group(1), "a" if m.group(1) == "---" else "b", m.group(3)), r.stdout, flags=re.M)
    txt = re.sub(r"^diff -ruN \S+?/(\S+) \S+?/(\S+)$", lambda m: "diff -ruN a/%s b/%s" % (m.group(1), m.group(2)), txt, flags=re.M)
    open(out, "w").write(txt)
    return txt.count("\n@@")


GOLDEN_DIGEST_RS = r'''// Driver of scripts/compare_with_libspartan.sh (added by rust_shim/seed_hooks.patch): proves the seeded instances of
// tests/golden/proof_digests.json on the CPU path of real libspartan and prints / writes the proof bytes.
//   golden_digest snark <log2_size> <seed> [out_file]     SNARK::prove, transcript b"snark_example", tape seed_scalar(b"tape", 100 + seed)
//   golden_digest nizk  <log2_size> <seed> [out_file]     NIZK::prove,  transcript b"nizk_example",  tape seed_scalar(b"tape", seed)
// Output: kind, key, proof length, SHA3-256 of bincode(proof) (sha3 is a dependency of the crate; SHA-256 is not), for NIZK also the
// length and SHA3-256 of the R1CSShapeDigest (the zlib stream flate2 produced). With out_file the bincode bytes are written there, so
// the calling script can take their SHA-256 with python.
use libspartan::{seed_scalar, Instance, NIZKGens, SNARKGens, NIZK, SNARK};
use merlin::Transcript;
use sha3::{Digest, Sha3_256};

fn hex(b: &[u8]) -> String {
  b.iter().map(|x| format!("{:02x}", x)).collect()
}

fn main() {
  let a: Vec<String> = std::env::args().collect();
  let kind = a[1].as_str();
  let (s, seed): (usize, u64) = (a[2].parse().unwrap(), a[3].parse().unwrap());
  let n = 1usize << s;
  let ni = if n > 16 { 10 } else { 1 };
  let (inst, vars, inputs) = Instance::produce_synthetic_r1cs_seeded(n, n, ni, seed);
  let bytes = if kind == "snark" {
    let gens = SNARKGens::new(n, n, ni, n);
    let (comm, decomm) = SNARK::encode(&inst, &gens);
    let mut t = Transcript::new(b"snark_example");
    let proof = SNARK::prove_with_tape_seed(&inst, &comm, &decomm, vars, &inputs, &gens, &mut t, &seed_scalar(b"tape", 100 + seed));
    println!("comm_sha3_256 {}", hex(&Sha3_256::digest(&bincode::serialize(&comm).unwrap())));
    bincode::serialize(&proof).unwrap()
  } else {
    let gens = NIZKGens::new(n, n, ni);
    let d = inst.shape_digest();
    println!("shape_digest_len {} shape_digest_sha3_256 {}", d.len(), hex(&Sha3_256::digest(&d)));
    let mut t = Transcript::new(b"nizk_example");
    let proof = NIZK::prove_with_tape_seed(&inst, vars, &inputs, &gens, &mut t, &seed_scalar(b"tape", seed));
    bincode::serialize(&proof).unwrap()
  };
  println!("{} s{}_seed{} len {} sha3_256 {}", kind, s, seed, bytes.len(), hex(&Sha3_256::digest(&bytes)));
  if a.len() > 4 {
    std::fs::write(&a[4], &bytes).unwrap();
  }
}
'''


def seed_hooks(root):
    # src/random.rs:11-18 — the seeded constructor next to RandomTape::new
    edit(root, "src/random.rs", [("""    Self { tape }
  }

  pub fn random_scalar(""", """    Self { tape }
  }

  /// Determinism hook of the byte-parity contract: the tape of `new` with the OsRng draw replaced by `seed`.
  /// A seed fixes every blind of the proof: outside tests it must be secret, >= 256 bits of entropy, and used once.
  pub fn new_with_seed(name: &'static [u8], seed: &Scalar) -> Self {
    let mut tape = Transcript::new(name);
    tape.append_scalar(b"init_randomness", seed);
    Self { tape }
  }

  pub fn random_scalar(""")])
    # src/r1cs.rs:160-238 — the draw of Z becomes a parameter; the OsRng form and the seeded form share everything after it
    edit(root, "src/r1cs.rs", [
        ("""    Timer::print(&format!("number_of_inputs {num_inputs}"));

    let mut csprng: OsRng = OsRng;
""", """    Timer::print(&format!("number_of_inputs {num_inputs}"));

    let mut csprng: OsRng = OsRng;
    Self::produce_synthetic_r1cs_with(num_cons, num_vars, num_inputs, &mut |_i| Scalar::random(&mut csprng))
  }

  /// `produce_synthetic_r1cs` with the assignment drawn from SHAKE256("spartan-synthetic-r1cs" || LE64(seed)): Z[i] is
  /// `Scalar::from_bytes_wide` of the next 64 bytes of the stream, i.e. what `Scalar::random` makes of 64 random bytes
  pub fn produce_synthetic_r1cs_seeded(
    num_cons: usize,
    num_vars: usize,
    num_inputs: usize,
    seed: u64,
  ) -> (R1CSShape, Vec<Scalar>, Vec<Scalar>) {
    use digest::{ExtendableOutput, Input, XofReader};
    let mut shake = sha3::Shake256::default();
    shake.input(b"spartan-synthetic-r1cs");
    shake.input(seed.to_le_bytes());
    let mut reader = shake.xof_result();
    Self::produce_synthetic_r1cs_with(num_cons, num_vars, num_inputs, &mut |_i| {
      let mut buf = [0u8; 64];
      reader.read(&mut buf);
      Scalar::from_bytes_wide(&buf)
    })
  }

  fn produce_synthetic_r1cs_with(
    num_cons: usize,
    num_vars: usize,
    num_inputs: usize,
    draw: &mut dyn FnMut(usize) -> Scalar,
  ) -> (R1CSShape, Vec<Scalar>, Vec<Scalar>) {
"""),
        ("""        .map(|_i| Scalar::random(&mut csprng))
""", """        .map(|i| draw(i))
"""),
    ])
    # src/lib.rs — seed_scalar, the seeded instance, the provers with a caller-supplied tape seed
    edit(root, "src/lib.rs", [
        ("""/// `ComputationCommitment` holds a public preprocessed NP statement (e.g., R1CS)
""", """/// TEST HOOK: the reproducible 64-bit-seed -> scalar map behind the tape seeds of the byte-parity fixtures:
/// `from_bytes_wide(SHAKE256(domain || LE64(seed))[..64])`
pub fn seed_scalar(domain: &[u8], seed: u64) -> Scalar {
  use digest::{ExtendableOutput, Input, XofReader};
  let mut shake = sha3::Shake256::default();
  shake.input(domain);
  shake.input(seed.to_le_bytes());
  let mut buf = [0u8; 64];
  shake.xof_result().read(&mut buf);
  Scalar::from_bytes_wide(&buf)
}

/// `ComputationCommitment` holds a public preprocessed NP statement (e.g., R1CS)
"""),
        ("""/// `SNARKGens` holds public parameters for producing and verifying proofs with the Spartan SNARK
""", """impl Instance {
  /// `produce_synthetic_r1cs` with a seeded assignment (`R1CSShape::produce_synthetic_r1cs_seeded`): the instances of the byte-parity fixtures
  pub fn produce_synthetic_r1cs_seeded(
    num_cons: usize,
    num_vars: usize,
    num_inputs: usize,
    seed: u64,
  ) -> (Instance, VarsAssignment, InputsAssignment) {
    let (inst, vars, inputs) =
      R1CSShape::produce_synthetic_r1cs_seeded(num_cons, num_vars, num_inputs, seed);
    let digest = inst.get_digest();
    (
      Instance { inst, digest },
      VarsAssignment { assignment: vars },
      InputsAssignment { assignment: inputs },
    )
  }

  /// the `R1CSShapeDigest` bytes `NIZK::prove` absorbs (the zlib stream of the serialized shape)
  pub fn shape_digest(&self) -> Vec<u8> {
    self.digest.clone()
  }
}

/// `SNARKGens` holds public parameters for producing and verifying proofs with the Spartan SNARK
"""),
        ("""    gens: &SNARKGens,
    transcript: &mut Transcript,
  ) -> Self {
    let timer_prove = Timer::new("SNARK::prove");

    // we create a Transcript object seeded with a random Scalar
    // to aid the prover produce its randomness
    let mut random_tape = RandomTape::new(b"proof");
""", """    gens: &SNARKGens,
    transcript: &mut Transcript,
  ) -> Self {
    // we create a Transcript object seeded with a random Scalar
    // to aid the prover produce its randomness
    let random_tape = RandomTape::new(b"proof");
    Self::prove_with_tape(inst, comm, decomm, vars, inputs, gens, transcript, random_tape)
  }

  /// TEST HOOK of the byte-parity contract: `prove` with the prover's randomness derived from `tape_seed` instead of the OS
  #[allow(clippy::too_many_arguments)]
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
    let random_tape = RandomTape::new_with_seed(b"proof", tape_seed);
    Self::prove_with_tape(inst, comm, decomm, vars, inputs, gens, transcript, random_tape)
  }

  #[allow(clippy::too_many_arguments)]
  fn prove_with_tape(
    inst: &Instance,
    comm: &ComputationCommitment,
    decomm: &ComputationDecommitment,
    vars: VarsAssignment,
    inputs: &InputsAssignment,
    gens: &SNARKGens,
    transcript: &mut Transcript,
    mut random_tape: RandomTape,
  ) -> Self {
    let timer_prove = Timer::new("SNARK::prove");
"""),
        ("""    gens: &NIZKGens,
    transcript: &mut Transcript,
  ) -> Self {
    let timer_prove = Timer::new("NIZK::prove");
    // we create a Transcript object seeded with a random Scalar
    // to aid the prover produce its randomness
    let mut random_tape = RandomTape::new(b"proof");
""", """    gens: &NIZKGens,
    transcript: &mut Transcript,
  ) -> Self {
    // we create a Transcript object seeded with a random Scalar
    // to aid the prover produce its randomness
    let random_tape = RandomTape::new(b"proof");
    Self::prove_with_tape(inst, vars, input, gens, transcript, random_tape)
  }

  /// TEST HOOK of the byte-parity contract: `prove` with the prover's randomness derived from `tape_seed` instead of the OS
  pub fn prove_with_tape_seed(
    inst: &Instance,
    vars: VarsAssignment,
    input: &InputsAssignment,
    gens: &NIZKGens,
    transcript: &mut Transcript,
    tape_seed: &Scalar,
  ) -> Self {
    let random_tape = RandomTape::new_with_seed(b"proof", tape_seed);
    Self::prove_with_tape(inst, vars, input, gens, transcript, random_tape)
  }

  fn prove_with_tape(
    inst: &Instance,
    vars: VarsAssignment,
    input: &InputsAssignment,
    gens: &NIZKGens,
    transcript: &mut Transcript,
    mut random_tape: RandomTape,
  ) -> Self {
    let timer_prove = Timer::new("NIZK::prove");
"""),
    ])
    write(root, "examples/golden_digest.rs", GOLDEN_DIGEST_RS)


# reference functions the seams redefine under `#[cfg(feature = "gpu")]` (same name, same impl): the original gets `cfg(not(gpu))`
SEAM_TARGET = {"commitments": "src/commitments.rs", "dense_mlpoly": "src/dense_mlpoly.rs", "sumcheck": "src/sumcheck.rs", "r1csproof": "src/r1csproof.rs",
               "product_tree": "src/product_tree.rs", "sparse_mlpoly": "src/sparse_mlpoly.rs", "nizk": "src/nizk/mod.rs", "bullet": "src/nizk/bullet.rs", "lib": "src/lib.rs"}
REDEFINED = {
    "src/commitments.rs": ["  pub fn new(n: usize, label: &[u8]) -> Self {", "  pub fn split_at(&self, mid: usize) -> (MultiCommitGens, MultiCommitGens) {"],
    "src/dense_mlpoly.rs": ["  pub fn bound_poly_var_top(&mut self, r: &Scalar) {", "  pub fn evaluate(&self, r: &[Scalar]) -> Scalar {"],
    "src/product_tree.rs": ["  pub fn evaluate(&self) -> Scalar {"],
}


def gpu_feature(root):
    edit(root, "Cargo.toml", [('multicore = ["rayon"]\n', 'multicore = ["rayon"]\n# MI355X prover hot path behind the C ABI of libspartan_hip.so (do not combine with multicore)\ngpu = []\n')])
    write(root, "build.rs", '''// added by rust_shim/gpu_feature.patch: link the MI355X library when the `gpu` feature is on
fn main() {
  if std::env::var_os("CARGO_FEATURE_GPU").is_some() {
    let dir = std::env::var("SPARTAN_HIP_LIB_DIR").expect("SPARTAN_HIP_LIB_DIR = the directory of libspartan_hip.so / libspartan_host.so");
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=spartan_hip");
    println!("cargo:rustc-link-lib=dylib=spartan_host");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
  }
}
''')
    edit(root, "src/lib.rs", [("mod unipoly;\n", "mod unipoly;\n\n#[cfg(feature = \"gpu\")]\n#[allow(missing_docs, clippy::all)]\nmod gpu;\n")])
    shutil.copy(os.path.join(ROOT, "rust_shim", "src", "gpu.rs"), os.path.join(root, "src", "gpu.rs"))
    for name, target in SEAM_TARGET.items():
        src = open(os.path.join(ROOT, "rust_shim", "seams", name + ".rs")).read()
        write(root, "src/gpu_seams/%s.rs" % name, src)
        rel = os.path.relpath(os.path.join(root, "src", "gpu_seams", name + ".rs"), os.path.dirname(os.path.join(root, target)))
        p = os.path.join(root, target)
        s = open(p).read()
        # the fragment is pasted into its module by the compiler: include! resolves relative to the including file
        s = s.rstrip("\n") + "\n\n// MI355X seam bodies of this module (rust_shim/seams/%s.rs)\n#[cfg(feature = \"gpu\")]\ninclude!(\"%s\");\n" % (name, rel)
        for sig in REDEFINED.get(target, []):
            assert s.count(sig) >= 1, (target, sig)
            s = s.replace(sig, "  #[cfg(not(feature = \"gpu\"))]\n" + sig, 1)
        if target == "src/lib.rs":
            # the seam redefines SNARK::{prove, prove_with_tape_seed, prove_with_tape} and NIZK::prove: the CPU bodies (as left by seed_hooks.patch)
            # are compiled only without the feature; NIZK's seeded CPU twin stays (the seam has none)
            gate = "  #[cfg(not(feature = \"gpu\"))]\n"
            i_snark, i_nizk = s.index("impl SNARK {"), s.index("impl NIZK {")
            snark, rest = s[i_snark:i_nizk], s[i_nizk:]
            for sig in ("  pub fn prove(\n", "  pub fn prove_with_tape_seed(\n", "  fn prove_with_tape(\n"):
                assert snark.count(sig) == 1, sig
                snark = snark.replace(sig, gate + sig)
            assert rest.count("  pub fn prove(\n") == 1
            rest = rest.replace("  pub fn prove(\n", gate + "  pub fn prove(\n", 1)
            s = s[:i_snark] + snark + rest
            # SNARKGens::new: the table geometries of its two generator streams are planned together before either is created
            sig = "    let gens_r1cs_sat = R1CSGens::new(b\"gens_r1cs_sat\", num_cons, num_vars_padded);\n"
            assert s.count(sig) == 2 and s.index(sig) > s.index("impl SNARKGens {") and s.index(sig) < s.index("impl NIZKGens {"), sig   # SNARKGens::new first, NIZKGens::new second
            s = s.replace(sig, "    #[cfg(feature = \"gpu\")]\n    gpu::plan_snark_gens(num_cons, num_vars_padded, num_nz_entries);\n" + sig, 1)
        open(p, "w").write(s)


def main():
    tmp = tempfile.mkdtemp(prefix="spartan_patches_")
    a, b, c = os.path.join(tmp, "pristine"), os.path.join(tmp, "seeded"), os.path.join(tmp, "gpu")
    copy_ref(a)
    copy_ref(b); seed_hooks(b)
    shutil.copytree(b, c); gpu_feature(c)
    n1 = diff(a, b, os.path.join(ROOT, "rust_shim", "seed_hooks.patch"))
    n2 = diff(b, c, os.path.join(ROOT, "rust_shim", "gpu_feature.patch"))
    shutil.rmtree(tmp)
    print("seed_hooks.patch: %d hunks; gpu_feature.patch: %d hunks" % (n1, n2))


if __name__ == "__main__":
    main()
