"""Probe: K independent SNARK::prove streams on ONE GPU (one context + host thread each), with and without the proof gate
(option host.proof_gate: one proof at a time in the throughput-bound first part, spark.inc). A single proof leaves the GPU idle during its
latency-bound second part, so concurrent proofs fill each other's gaps; the gate turns that into a pipeline.
usage: python bench/concurrent_probe.py [log2_constraints] [max_K] [steps]"""
import sys, os, time, threading, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spartan_amd import prover as P
s = int(sys.argv[1]) if len(sys.argv) > 1 else 20
KMAX = int(sys.argv[2]) if len(sys.argv) > 2 else 4
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
N = 1 << s
workers = []
for k in range(KMAX):
    ctx = P.Ctx(0)
    inst = P.Instance.produce_synthetic_r1cs(ctx, N, N, 10, seed=1000 + k)
    gens = P.SNARKGens(ctx, N, N, 10, N)
    enc = P.SNARK.encode(ctx, inst, gens)
    workers.append((ctx, inst, gens, enc, P.seed_scalar(b"tape", 1000 + k)))
ref = []
for w in workers:  # warm-up, and the bytes every later proof of this worker must reproduce
    ref.append(P.SNARK.prove(w[0], w[1], w[3], w[1].vars, w[1].inputs, w[2], b"snark_example", w[4]))
out = []
for gate in (0, 1):
    P.H.spz_ctx_set_option(None, b"host.proof_gate", str(gate).encode())
    for K in range(1, KMAX + 1):
        ok = [True] * K
        def run(i):
            w = workers[i]
            for _ in range(steps):
                if P.SNARK.prove(w[0], w[1], w[3], w[1].vars, w[1].inputs, w[2], b"snark_example", w[4]) != ref[i]:
                    ok[i] = False
        ths = [threading.Thread(target=run, args=(i,)) for i in range(K)]
        t0 = time.perf_counter()
        for t in ths: t.start()
        for t in ths: t.join()
        dt = time.perf_counter() - t0
        r = {"gate": gate, "proofs_in_flight": K, "ms_per_proof_slot": dt / steps * 1e3, "M_constraints_per_s": K * steps * N / dt / 1e6, "bytes_identical": all(ok)}
        out.append(r)
        print(json.dumps(r), flush=True)
P.H.spz_ctx_set_option(None, b"host.proof_gate", b"0")
for w in workers:
    w[3].free(); w[2].free(); w[1].free(); w[0].close()
