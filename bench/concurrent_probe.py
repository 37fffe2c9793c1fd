"""Scratch probe: K independent SNARK::prove streams on ONE GPU (one context + host thread each). A single proof leaves the
GPU idle about half the time (Fiat-Shamir round trips), so concurrent proofs fill each other's gaps."""
import sys, os, time, threading, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spartan_amd import prover as P
s = int(sys.argv[1]) if len(sys.argv) > 1 else 20
N = 1 << s
for K in (1, 2, 4, 8):
    workers = []
    for k in range(K):
        ctx = P.Ctx(0)
        inst = P.Instance.produce_synthetic_r1cs(ctx, N, N, 10, seed=k)
        gens = P.SNARKGens(ctx, N, N, 10, N)
        enc = P.SNARK.encode(ctx, inst, gens)
        workers.append((ctx, inst, gens, enc, P.seed_scalar(b"tape", k)))
    steps = 6
    def run(w):
        ctx, inst, gens, enc, seed = w
        for _ in range(steps):
            P.SNARK.prove(ctx, inst, enc, inst.vars, inst.inputs, gens, b"snark_example", seed)
    for w in workers: run((w[0], w[1], w[2], w[3], w[4])) if False else None
    ths = [threading.Thread(target=run, args=(w,)) for w in workers]
    # warm-up
    for w in workers: P.SNARK.prove(w[0], w[1], w[3], w[1].vars, w[1].inputs, w[2], b"snark_example", w[4])
    t0 = time.perf_counter()
    for t in ths: t.start()
    for t in ths: t.join()
    dt = time.perf_counter() - t0
    print("K=%d concurrent proofs: %.1f ms per proof-slot, aggregate %.2f M constraints/s" % (K, dt / steps * 1e3, K * steps * N / dt / 1e6), flush=True)
    for w in workers:
        w[3].free(); w[2].free(); w[1].free(); w[0].close()
