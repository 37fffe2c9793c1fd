#!/usr/bin/env python3
"""Generates tests/golden/proof_digests.json from the ORACLE (oracle/liboracle.so): SHA-256 of the bincode proof bytes of
seeded NIZK::prove / SNARK::prove runs, plus the generator-stream heads and the first witness commitment share.

Why digests of the oracle and not of the reference: /root/reference cannot be executed (no Rust toolchain), and
its own tests pin no proof byte. These fixtures are regression pins — they freeze today's oracle output so that a later
change to the oracle (or to the HIP path, which the GPU tests compare against the same file) cannot drift silently.

Two groups:
  CASES      small instances; the CPU test suite regenerates them on every run (tests/test_golden.py).
  BIG_CASES  the BASELINE.json configurations (SNARK 2^16 / 2^20 / 2^22, NIZK 2^16 / 2^20): minutes of oracle time each,
             generated once with `--big` (OpenMP threads = all cores) and kept under the "big" key; the GPU suite proves
             the same instances through the HIP path and compares length, SHA-256 and SHA-256 of each proof part.
Run:  python tests/golden/make_golden.py [--big]   (from the repo root)"""
import ctypes, hashlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.helpers import load_oracle, sz, vp, u64x4, gens_bytes, real_miniz_zlib, oracle_shape_bincode

CASES = {"nizk": [(4, 2), (7, 3), (12, 5)], "snark": [(3, 1), (5, 2), (8, 3), (12, 4), (15, 5)]}  # (log2 size, seed)
BIG_CASES = {"nizk": [(16, 6), (20, 0)], "snark": [(16, 6), (20, 0), (22, 0)]}  # seed 0 = bench.py's instance and tape


def proof_bytes(orc, p):
    n = orc.orc_proof_bytes(p, None, sz(0)); b = (ctypes.c_uint8 * n)(); orc.orc_proof_bytes(p, b, sz(n)); return bytes(b)


def digest_entry(orc, p):
    b = proof_bytes(orc, p)
    lens = (ctypes.c_size_t * 3)()
    orc.orc_proof_part_lens(p, lens)
    e = {"len": len(b), "sha256": hashlib.sha256(b).hexdigest(), "first_share": b[8:40].hex()}
    # digests of the two proof halves localise a mismatch: r1cs_sat_proof | everything after it (NIZK: rx, ry; SNARK:
    # inst_evals + r1cs_eval_proof)
    l0 = int(lens[0])
    e["sat_len"] = l0
    e["sat_sha256"] = hashlib.sha256(b[:l0]).hexdigest()
    e["rest_sha256"] = hashlib.sha256(b[l0:]).hexdigest()
    return e


def nizk_case(orc, s, seed):
    N = 1 << s; ni = 10 if N > 16 else 1
    inst = vp(orc.orc_instance_synthetic(sz(N), sz(N), sz(ni), ctypes.c_uint64(seed)))
    g = vp(orc.orc_nizk_gens_new(sz(N), sz(N), sz(ni)))
    tape = u64x4(); orc.orc_seed_scalar(b"tape", ctypes.c_uint64(seed), tape)
    # R1CSShapeDigest as the reference forms it (src/r1cs.rs:154-158): zlib level 6 of bincode(shape) — here through the REAL C miniz
    # bundled in libtorch (tests/helpers.py: real_miniz_zlib), not through the product's deflate.cc: the GPU tests prove with the digest
    # the product computes itself and must land on these bytes
    d = real_miniz_zlib(oracle_shape_bincode(orc, inst), 6)
    p = vp(orc.orc_nizk_prove(inst, g, d, sz(len(d)), b"nizk_example", tape, None))
    e = digest_entry(orc, p)
    e["shape_digest_len"] = len(d)
    e["shape_digest_sha256"] = hashlib.sha256(d).hexdigest()
    orc.orc_proof_free(p); orc.orc_nizk_gens_free(g); orc.orc_instance_free(inst)
    return e


def snark_case(orc, s, seed):
    N = 1 << s; ni = 10 if N > 16 else 1
    inst = vp(orc.orc_instance_synthetic(sz(N), sz(N), sz(ni), ctypes.c_uint64(seed)))
    g = vp(orc.orc_snark_gens_new(sz(N), sz(N), sz(ni), sz(N)))
    e = vp(orc.orc_snark_encode(inst, g))
    tape = u64x4(); orc.orc_seed_scalar(b"tape", ctypes.c_uint64(100 + seed), tape)
    p = vp(orc.orc_snark_prove(inst, g, e, b"snark_example", tape, None))
    ent = digest_entry(orc, p)
    n = orc.orc_commitment_bincode(e, None, sz(0)); cb = (ctypes.c_uint8 * n)(); orc.orc_commitment_bincode(e, cb, sz(n))
    ent["comm_sha256"] = hashlib.sha256(bytes(cb)).hexdigest()  # bincode(ComputationCommitment): pins SNARK::encode too
    orc.orc_proof_free(p); orc.orc_encode_free(e); orc.orc_snark_gens_free(g); orc.orc_instance_free(inst)
    return ent


def run(orc, cases=CASES):
    out = {"generators": {}, "nizk": {}, "snark": {}}
    for label in (b"gens_r1cs_sat", b"gens_r1cs_eval"):
        out["generators"][label.decode()] = gens_bytes(orc, 3, label).hex()
    for s, seed in cases["nizk"]:
        out["nizk"][f"s{s}_seed{seed}"] = nizk_case(orc, s, seed)
    for s, seed in cases["snark"]:
        out["snark"][f"s{s}_seed{seed}"] = snark_case(orc, s, seed)
    return out


if __name__ == "__main__":
    orc = load_oracle()
    path = os.path.join(ROOT, "tests", "golden", "proof_digests.json")
    if "--big" in sys.argv:
        orc.orc_set_threads(ctypes.c_int(os.cpu_count() or 1))
        res = json.load(open(path))
        big = res.setdefault("big", {"nizk": {}, "snark": {}})
        for kind, fn in (("nizk", nizk_case), ("snark", snark_case)):
            for s, seed in BIG_CASES[kind]:
                key = f"s{s}_seed{seed}"
                if key in big[kind] and "--force" not in sys.argv and not (kind == "nizk" and "shape_digest_sha256" not in big[kind][key]):
                    continue
                t0 = time.time()
                big[kind][key] = fn(orc, s, seed)
                big[kind][key]["oracle_seconds"] = round(time.time() - t0, 1)
                print(kind, key, big[kind][key], flush=True)
                json.dump(res, open(path, "w"), indent=1, sort_keys=True)
    else:
        res = run(orc)
        old = json.load(open(path)) if os.path.exists(path) else {}
        if "big" in old:
            res["big"] = old["big"]
        json.dump(res, open(path, "w"), indent=1, sort_keys=True)
    print("wrote", path)
