"""The Rust side of the parity link, as far as it can be checked without a Rust toolchain (missing #1): rust_shim/seed_hooks.patch
(seeded RandomTape + seeded produce_synthetic_r1cs + the golden_digest example) and rust_shim/gpu_feature.patch (the `gpu` cargo feature) must
APPLY to the reference snapshot this repo was built against — `git apply --check` on a copy of /root/reference — and the example
scripts/compare_with_libspartan.sh builds must use no crate outside the reference's own Cargo.toml."""
import os, re, shutil, subprocess, sys
import pytest
from tests.helpers import ROOT

REF = "/root/reference"
needs_ref = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src")), reason="the reference snapshot is not on this machine")


def _copy_ref(tmp_path):
    dst = tmp_path / "spartan"
    shutil.copytree(REF, dst, ignore=shutil.ignore_patterns(".git", "target"))
    return str(dst)


def _apply(cwd, patch, check_only=False):
    cmd = ["git", "apply", "--check" if check_only else "--verbose", os.path.join(ROOT, "rust_shim", patch)]
    return subprocess.run(cmd, cwd=cwd, capture_output=True, text=True)


@needs_ref
def test_seed_hooks_and_gpu_feature_patches_apply_to_the_reference(tmp_path):
    d = _copy_ref(tmp_path)
    r = _apply(d, "seed_hooks.patch", check_only=True)
    assert r.returncode == 0, r.stderr
    assert _apply(d, "seed_hooks.patch").returncode == 0
    lib = open(os.path.join(d, "src", "lib.rs")).read()
    for name in ("pub fn seed_scalar", "pub fn produce_synthetic_r1cs_seeded", "pub fn prove_with_tape_seed", "fn prove_with_tape", "pub fn shape_digest"):
        assert name in lib, name
    assert lib.count("pub fn prove_with_tape_seed") == 2                                  # SNARK and NIZK
    assert "pub fn new_with_seed" in open(os.path.join(d, "src", "random.rs")).read()
    r1cs = open(os.path.join(d, "src", "r1cs.rs")).read()
    assert "produce_synthetic_r1cs_seeded" in r1cs and "spartan-synthetic-r1cs" in r1cs and ".map(|i| draw(i))" in r1cs
    assert os.path.exists(os.path.join(d, "examples", "golden_digest.rs"))
    # the gpu feature goes on top
    r = _apply(d, "gpu_feature.patch", check_only=True)
    assert r.returncode == 0, r.stderr
    assert _apply(d, "gpu_feature.patch").returncode == 0
    assert "gpu = []" in open(os.path.join(d, "Cargo.toml")).read()
    assert open(os.path.join(d, "src", "gpu.rs")).read() == open(os.path.join(ROOT, "rust_shim", "src", "gpu.rs")).read()
    for seam in os.listdir(os.path.join(ROOT, "rust_shim", "seams")):
        assert open(os.path.join(d, "src", "gpu_seams", seam)).read() == open(os.path.join(ROOT, "rust_shim", "seams", seam)).read(), seam
    libg = open(os.path.join(d, "src", "lib.rs")).read()
    assert libg.count("gpu::plan_snark_gens(num_cons, num_vars_padded, num_nz_entries);") == 1 and libg.index("gpu::plan_snark_gens") < libg.index("impl NIZKGens {")
    assert 'include!("gpu_seams/sumcheck.rs");' in open(os.path.join(d, "src", "sumcheck.rs")).read()
    assert 'include!("../gpu_seams/bullet.rs");' in open(os.path.join(d, "src", "nizk", "bullet.rs")).read()


@needs_ref
def test_patches_are_what_the_generator_emits(tmp_path):
    """the committed patches are reproducible from the reference + rust_shim/ (make_patches.py is idempotent on the committed state)"""
    before = {p: open(os.path.join(ROOT, "rust_shim", p)).read() for p in ("seed_hooks.patch", "gpu_feature.patch")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "rust_shim", "make_patches.py"), REF], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    for p, txt in before.items():
        assert open(os.path.join(ROOT, "rust_shim", p)).read() == txt, p


def test_golden_digest_example_uses_only_the_reference_dependencies():
    """the crate roots the example names are libspartan itself, std, and [dependencies] of the reference's Cargo.toml (sha2 / hex are not)"""
    patch = open(os.path.join(ROOT, "rust_shim", "seed_hooks.patch")).read()
    ex = "\n".join(l[1:] for l in patch.split("diff -ruN a/examples/golden_digest.rs")[1].split("\ndiff -ruN ")[0].splitlines() if l.startswith("+") and not l.startswith("+++"))
    code = re.sub(r"//[^\n]*", "", ex)
    roots = set(re.findall(r"\buse\s+([a-z_0-9]+)::", code)) | set(re.findall(r"(?<![\w:])([a-z_][a-z_0-9]*)::[A-Za-z_]", code))
    allowed = {"libspartan", "std", "curve25519_dalek", "merlin", "rand", "rand_core", "digest", "sha3", "byteorder", "serde", "bincode", "subtle", "itertools", "flate2"}
    assert roots and roots <= allowed, roots - allowed
    script = open(os.path.join(ROOT, "scripts", "compare_with_libspartan.sh")).read()
    assert "sha2::" not in script and "use sha2" not in script and "hex::" not in script and "seed_hooks.patch" in script and "nizk" in script
