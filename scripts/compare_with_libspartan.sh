#!/usr/bin/env bash
# Compares real libspartan (Rust, CPU path) against the digests in tests/golden/proof_digests.json.
# Needs: a checkout of microsoft/Spartan patched with rust_shim/seams/{random,r1cs}.rs (the two seed hooks), cargo, python3.
# Usage: scripts/compare_with_libspartan.sh /path/to/Spartan [log2_size seed]...   (default: the small SNARK cases + 2^20)
set -euo pipefail
SPARTAN=${1:?path to the patched Spartan checkout}; shift || true
HERE=$(cd "$(dirname "$0")/.." && pwd)
CASES=("$@"); [ ${#CASES[@]} -gt 0 ] || CASES=(3 1 5 2 8 3 12 4 15 5 16 6 20 0)
mkdir -p "$SPARTAN/examples"
cat > "$SPARTAN/examples/golden_digest.rs" <<'RS'
// proves produce_synthetic_r1cs_seeded(2^s, 2^s, ni, seed) with RandomTape::new_with_seed(seed_scalar("tape", 100 + seed))
// and prints sha256(bincode(proof)): the protocol of tests/golden/make_golden.py
use libspartan::{Instance, SNARKGens, SNARK};
use merlin::Transcript;
use sha2::{Digest, Sha256};
fn main() {
  let a: Vec<String> = std::env::args().collect();
  let (s, seed): (usize, u64) = (a[1].parse().unwrap(), a[2].parse().unwrap());
  let n = 1usize << s;
  let ni = if n > 16 { 10 } else { 1 };
  let (inst, vars, inputs) = Instance::produce_synthetic_r1cs_seeded(n, n, ni, seed);
  let gens = SNARKGens::new(n, n, ni, n);
  let (comm, decomm) = SNARK::encode(&inst, &gens);
  let mut t = Transcript::new(b"snark_example");
  let proof = SNARK::prove_with_tape_seed(&inst, &comm, &decomm, vars, &inputs, &gens, &mut t, &libspartan::seed_scalar(b"tape", 100 + seed));
  let bytes = bincode::serialize(&proof).unwrap();
  println!("snark s{}_seed{} len {} sha256 {}", s, seed, bytes.len(), hex::encode(Sha256::digest(&bytes)));
}
RS
( cd "$SPARTAN" && cargo build --release --example golden_digest )
for ((i = 0; i < ${#CASES[@]}; i += 2)); do
  s=${CASES[i]}; seed=${CASES[i+1]}
  got=$("$SPARTAN/target/release/examples/golden_digest" "$s" "$seed")
  want=$(python3 - "$HERE" "$s" "$seed" <<'PY'
import json, sys
root, s, seed = sys.argv[1], sys.argv[2], sys.argv[3]
g = json.load(open(f"{root}/tests/golden/proof_digests.json"))
key = f"s{s}_seed{seed}"
e = g["snark"].get(key) or g.get("big", {}).get("snark", {}).get(key)
print(f"len {e['len']} sha256 {e['sha256']}" if e else "no fixture")
PY
)
  echo "$got"; echo "  fixture: $want"
  case "$got" in *"$want"*) echo "  MATCH";; *) echo "  MISMATCH"; fi=1;; esac
done
exit ${fi:-0}
