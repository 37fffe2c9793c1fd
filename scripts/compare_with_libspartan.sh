#!/usr/bin/env bash
# Compares REAL libspartan (Rust, CPU path) against the digests in tests/golden/proof_digests.json — the one link of the parity chain that
# cannot be closed without a Rust toolchain (oracle and parity).
# Needs: cargo, python3, git, and a pristine checkout of microsoft/Spartan 0.9.0. The script applies rust_shim/seed_hooks.patch to a COPY
# of the checkout (seeded RandomTape, seeded produce_synthetic_r1cs, examples/golden_digest.rs; tests/test_rust_patches.py checks that the
# patch applies to the snapshot this repo was built against), builds the example with the crate's own dependency set — no crate outside
# the reference's Cargo.toml: the proof bytes are written to a file and hashed with python — and compares length + SHA-256.
# Usage: scripts/compare_with_libspartan.sh /path/to/Spartan [kind log2_size seed]...
#        default cases: every small fixture (SNARK and NIZK) + SNARK 2^16, 2^20 and NIZK 2^16, 2^20 of the "big" group
set -euo pipefail
SPARTAN=${1:?path to a pristine microsoft/Spartan checkout}; shift || true
HERE=$(cd "$(dirname "$0")/.." && pwd)
CASES=("$@")
[ ${#CASES[@]} -gt 0 ] || CASES=(snark 3 1 snark 5 2 snark 8 3 snark 12 4 snark 15 5 nizk 4 2 nizk 7 3 nizk 12 5 snark 16 6 nizk 16 6 snark 20 0 nizk 20 0)
WORK=$(mktemp -d)
trap 'rm -rf "$WORK"' EXIT
cp -r "$SPARTAN" "$WORK/spartan"
( cd "$WORK/spartan" && git apply --check "$HERE/rust_shim/seed_hooks.patch" && git apply "$HERE/rust_shim/seed_hooks.patch" )
( cd "$WORK/spartan" && cargo build --release --example golden_digest )
fail=0
for ((i = 0; i < ${#CASES[@]}; i += 3)); do
  kind=${CASES[i]}; s=${CASES[i+1]}; seed=${CASES[i+2]}
  "$WORK/spartan/target/release/examples/golden_digest" "$kind" "$s" "$seed" "$WORK/proof.bin" | sed 's/^/  libspartan: /'
  python3 - "$HERE" "$kind" "$s" "$seed" "$WORK/proof.bin" <<'PY' || fail=1
import hashlib, json, sys
root, kind, s, seed, path = sys.argv[1:6]
b = open(path, "rb").read()
g = json.load(open(f"{root}/tests/golden/proof_digests.json"))
key = f"s{s}_seed{seed}"
e = g[kind].get(key) or g.get("big", {}).get(kind, {}).get(key)
got = (len(b), hashlib.sha256(b).hexdigest())
if not e:
    print(f"  {kind} {key}: no fixture (libspartan: len {got[0]} sha256 {got[1]})"); sys.exit(0)
ok = got == (e["len"], e["sha256"])
print(f"  {kind} {key}: libspartan len {got[0]} sha256 {got[1]}\n  {' ' * (len(kind) + len(key) + 2)}fixture    len {e['len']} sha256 {e['sha256']}  ->  {'MATCH' if ok else 'MISMATCH'}")
if not ok and "sat_len" in e:   # localise: the R1CS satisfiability proof | everything after it
    l0 = e["sat_len"]
    print("     r1cs_sat_proof", "equal" if hashlib.sha256(b[:l0]).hexdigest() == e["sat_sha256"] else "DIFFERS", "| rest", "equal" if hashlib.sha256(b[l0:]).hexdigest() == e["rest_sha256"] else "DIFFERS")
sys.exit(0 if ok else 1)
PY
done
exit $fail
