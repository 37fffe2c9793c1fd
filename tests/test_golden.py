"""Golden fixtures (tests/golden/proof_digests.json, made by tests/golden/make_golden.py): the oracle must keep reproducing
them (CPU), and the HIP path must produce the same bytes (GPU)."""
import ctypes, hashlib, json, os
import pytest
from tests.helpers import *

GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "proof_digests.json")))


def test_oracle_reproduces_golden_fixtures(orc):
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(ROOT, "tests", "golden", "make_golden.py"))
    mg = importlib.util.module_from_spec(spec); spec.loader.exec_module(mg)
    small = {k: v for k, v in GOLD.items() if k != "big"}
    assert mg.run(orc) == small
    # the BASELINE-sized fixtures are minutes of oracle time each: generated once (make_golden.py --big), only checked for presence here
    for kind, cases in mg.BIG_CASES.items():
        for s, seed in cases:
            e = GOLD["big"][kind][f"s{s}_seed{seed}"]
            assert len(e["sha256"]) == 64 and e["len"] > 0 and e["sat_len"] > 0
    # README.md:362,374 proof lengths at 2^20 (len_r1cs_sat_proof 47024; SNARK = 47024 + 96 + 133720)
    assert GOLD["big"]["snark"]["s20_seed0"]["sat_len"] == 47024 and GOLD["big"]["snark"]["s20_seed0"]["len"] == 47024 + 96 + 133720
    assert GOLD["big"]["nizk"]["s20_seed0"]["sat_len"] == 47024
    # the generator heads are the values recorded in SURVEY.md §8c (computed through libsodium there)
    assert GOLD["generators"]["gens_r1cs_sat"][:64] == "f8dad3b0fba18ec2a61684952cbfd51372cbdcca26b05e5b0b4637157c98ca43"


def _check(b, want, what):
    l0 = want["sat_len"]
    assert len(b) == want["len"], what
    assert hashlib.sha256(b[:l0]).hexdigest() == want["sat_sha256"], what + ": r1cs_sat_proof differs"
    assert hashlib.sha256(b[l0:]).hexdigest() == want["rest_sha256"], what + ": the part after r1cs_sat_proof differs"
    assert hashlib.sha256(b).hexdigest() == want["sha256"] and b[8:40].hex() == want["first_share"], what


@pytest.mark.gpu
@pytest.mark.parametrize("kind,key", [("nizk", "s16_seed6"), ("snark", "s16_seed6"), ("nizk", "s20_seed0"), ("snark", "s20_seed0"), ("snark", "s22_seed0")])
def test_hip_path_reproduces_baseline_sized_fixtures(kind, key):
    """Byte identity at the BASELINE.json configurations (2^16, 2^20 headline, 2^22): the HIP path proves the instance the
    oracle proved when tests/golden/make_golden.py --big ran, and must produce the same bincode bytes (SHA-256 of the whole
    proof and of its two halves; for SNARKs also of bincode(ComputationCommitment), i.e. SNARK::encode)."""
    from spartan_amd import prover as P
    want = GOLD["big"][kind][key]
    s, seed = int(key.split("_")[0][1:]), int(key.split("seed")[1])
    N = 1 << s
    ctx = P.Ctx(0)
    inst = P.Instance.produce_synthetic_r1cs(ctx, N, N, 10, seed=seed)
    if kind == "snark":
        gens = P.SNARKGens(ctx, N, N, 10, N)
        enc = P.SNARK.encode(ctx, inst, gens)
        assert hashlib.sha256(enc.serialize_commitment()).hexdigest() == want["comm_sha256"]
        b = P.SNARK.prove(ctx, inst, enc, inst.vars, inst.inputs, gens, b"snark_example", P.seed_scalar(b"tape", 100 + seed))
        enc.free()
    else:
        # no set_digest: the product computes R1CSShapeDigest itself (deflate.cc); the fixture was made with the real miniz's stream
        dg = inst.digest()
        assert len(dg) == want["shape_digest_len"] and hashlib.sha256(dg).hexdigest() == want["shape_digest_sha256"]
        gens = P.NIZKGens(ctx, N, N, 10)
        b = P.NIZK.prove(ctx, inst, inst.vars, inst.inputs, gens, b"nizk_example", P.seed_scalar(b"tape", seed))
    gens.free(); inst.free(); ctx.close()
    _check(b, want, f"{kind} {key}")


@pytest.mark.gpu
def test_hip_path_reproduces_golden_fixtures():
    from spartan_amd import prover as P
    ctx = P.Ctx(0)
    for key, want in GOLD["snark"].items():
        s, seed = int(key.split("_")[0][1:]), int(key.split("seed")[1])
        N = 1 << s; ni = 10 if N > 16 else 1
        inst = P.Instance.produce_synthetic_r1cs(ctx, N, N, ni, seed=seed)
        gens = P.SNARKGens(ctx, N, N, ni, N)
        enc = P.SNARK.encode(ctx, inst, gens)
        b = P.SNARK.prove(ctx, inst, enc, inst.vars, inst.inputs, gens, b"snark_example", P.seed_scalar(b"tape", 100 + seed))
        assert len(b) == want["len"] and hashlib.sha256(b).hexdigest() == want["sha256"] and b[8:40].hex() == want["first_share"]
        enc.free(); gens.free(); inst.free()
    for key, want in GOLD["nizk"].items():
        s, seed = int(key.split("_")[0][1:]), int(key.split("seed")[1])
        N = 1 << s; ni = 10 if N > 16 else 1
        inst = P.Instance.produce_synthetic_r1cs(ctx, N, N, ni, seed=seed)
        assert hashlib.sha256(inst.digest()).hexdigest() == want["shape_digest_sha256"]  # computed by the product, fixture from the real miniz
        gens = P.NIZKGens(ctx, N, N, ni)
        b = P.NIZK.prove(ctx, inst, inst.vars, inst.inputs, gens, b"nizk_example", P.seed_scalar(b"tape", seed))
        assert len(b) == want["len"] and hashlib.sha256(b).hexdigest() == want["sha256"]
        gens.free(); inst.free()
    ctx.close()
