#!/usr/bin/env python3
"""Generates tests/golden/proof_digests.json from the ORACLE (oracle/liboracle.so): SHA-256 of the bincode proof bytes of
seeded NIZK::prove / SNARK::prove runs, plus the generator-stream heads and the first witness commitment share.

Why digests of the oracle and not of the reference: /root/reference cannot be executed (no Rust toolchain), and
its own tests pin no proof byte. These fixtures are regression pins — they freeze today's oracle output so that a later
change to the oracle (or to the HIP path, which the GPU tests compare against the same file) cannot drift silently.
Run:  python tests/golden/make_golden.py   (from the repo root)"""
import ctypes, hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.helpers import load_oracle, sz, vp, u64x4, gens_bytes

CASES = {"nizk": [(4, 2), (7, 3), (12, 5)], "snark": [(3, 1), (5, 2), (8, 3), (12, 4), (15, 5)]}  # (log2 size, seed)


def proof_bytes(orc, p):
    n = orc.orc_proof_bytes(p, None, sz(0)); b = (ctypes.c_uint8 * n)(); orc.orc_proof_bytes(p, b, sz(n)); return bytes(b)


def run(orc):
    out = {"generators": {}, "nizk": {}, "snark": {}}
    for label in (b"gens_r1cs_sat", b"gens_r1cs_eval"):
        out["generators"][label.decode()] = gens_bytes(orc, 3, label).hex()
    for s, seed in CASES["nizk"]:
        N = 1 << s; ni = 10 if N > 16 else 1
        inst = vp(orc.orc_instance_synthetic(sz(N), sz(N), sz(ni), ctypes.c_uint64(seed)))
        g = vp(orc.orc_nizk_gens_new(sz(N), sz(N), sz(ni)))
        tape = u64x4(); orc.orc_seed_scalar(b"tape", ctypes.c_uint64(seed), tape)
        d = b"digest-%d" % s
        p = vp(orc.orc_nizk_prove(inst, g, d, sz(len(d)), b"nizk_example", tape, None))
        b = proof_bytes(orc, p)
        out["nizk"][f"s{s}_seed{seed}"] = {"len": len(b), "sha256": hashlib.sha256(b).hexdigest(), "first_share": b[8:40].hex()}
    for s, seed in CASES["snark"]:
        N = 1 << s; ni = 10 if N > 16 else 1
        inst = vp(orc.orc_instance_synthetic(sz(N), sz(N), sz(ni), ctypes.c_uint64(seed)))
        g = vp(orc.orc_snark_gens_new(sz(N), sz(N), sz(ni), sz(N)))
        e = vp(orc.orc_snark_encode(inst, g))
        tape = u64x4(); orc.orc_seed_scalar(b"tape", ctypes.c_uint64(100 + seed), tape)
        p = vp(orc.orc_snark_prove(inst, g, e, b"snark_example", tape, None))
        b = proof_bytes(orc, p)
        out["snark"][f"s{s}_seed{seed}"] = {"len": len(b), "sha256": hashlib.sha256(b).hexdigest(), "first_share": b[8:40].hex()}
    return out


if __name__ == "__main__":
    res = run(load_oracle())
    path = os.path.join(ROOT, "tests", "golden", "proof_digests.json")
    json.dump(res, open(path, "w"), indent=1, sort_keys=True)
    print("wrote", path)
