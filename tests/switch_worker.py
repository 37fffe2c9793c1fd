"""Worker of tests/test_gpu_proofs.py::test_every_ab_switch_gives_the_same_proof: the options of the library and of the host driver
are handed to a fresh process through SPARTAN_OPTIONS, so each setting proves in a process of its own: SNARK::prove at 2^17 (the smallest size at which the batched sum-checks
have throughput-sized rounds, i.e. at which the eq-factor path and its hand-over run), seed and tape fixed; prints the SHA-256 of the proof."""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from spartan_amd import prover as P

s, seed = int(sys.argv[1]), int(sys.argv[2])
N = 1 << s
ctx = P.Ctx(0)
inst = P.Instance.produce_synthetic_r1cs(ctx, N, N, 10, seed=seed)
gens = P.SNARKGens(ctx, N, N, 10, N)
enc = P.SNARK.encode(ctx, inst, gens)
proof = P.SNARK.prove(ctx, inst, enc, inst.vars, inst.inputs, gens, b"snark_example", P.seed_scalar(b"tape", seed))
print("PROOF_SHA256", hashlib.sha256(proof).hexdigest(), len(proof), P.keccak_variant(), flush=True)
enc.free(); gens.free(); inst.free(); ctx.close()
