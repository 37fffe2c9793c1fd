"""Scratch probe: SPARTAN_OPTIONS=testing.unlock=1,host.callstats=1 python bench/callstats_probe.py [log2] -> per-entry-point wall time of ONE proof."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spartan_amd import prover as P
s = int(sys.argv[1]) if len(sys.argv) > 1 else 20
N = 1 << s
ctx = P.Ctx(0)
inst = P.Instance.produce_synthetic_r1cs(ctx, N, N, 10, seed=0)
gens = P.SNARKGens(ctx, N, N, 10, N)
enc = P.SNARK.encode(ctx, inst, gens)
tape = P.seed_scalar(b"tape", 0)
for i in range(3):
    t0 = time.perf_counter()
    P.SNARK.prove(ctx, inst, enc, inst.vars, inst.inputs, gens, b"snark_example", tape)
    print("prove ms", (time.perf_counter() - t0) * 1e3, file=sys.stderr)
