"""Worker of tests/test_gpu_sanitizers.py, started with LD_PRELOAD=<clang's libclang_rt.tsan> and the ThreadSanitizer builds of both libraries:
SNARK::prove + NIZK::prove at 2^12 with the assignment as a host buffer (the upload thread and the ZK look-ahead thread run), then three
contexts proving from three host threads at once (the process-wide table cache, the proof gate). Proofs must equal the single-thread ones."""
import os, sys, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from spartan_amd import prover as P

s = int(sys.argv[1]) if len(sys.argv) > 1 else 12
N = 1 << s


def setup(seed):
    ctx = P.Ctx(0)
    inst = P.Instance.produce_synthetic_r1cs(ctx, N, N, 10, seed=seed)
    inst.set_digest(b"tsan")
    gens = P.SNARKGens(ctx, N, N, 10, N)
    ngens = P.NIZKGens(ctx, N, N, 10)
    enc = P.SNARK.encode(ctx, inst, gens)
    return ctx, inst, gens, ngens, enc


def prove(w, seed):
    ctx, inst, gens, ngens, enc = w
    a = P.SNARK.prove(ctx, inst, enc, inst.vars, inst.inputs, gens, b"snark_example", P.seed_scalar(b"tape", seed))
    b = P.NIZK.prove(ctx, inst, inst.vars, inst.inputs, ngens, b"nizk_example", P.seed_scalar(b"tape", seed))
    return a, b


workers = [setup(k) for k in range(3)]
single = [prove(w, k) for k, w in enumerate(workers)]
P.H.spz_ctx_set_option(None, b"host.proof_gate", b"1")
got = [None] * 3


def run(k):
    for _ in range(2):
        got[k] = prove(workers[k], k)


ths = [threading.Thread(target=run, args=(k,)) for k in range(3)]
for t in ths:
    t.start()
for t in ths:
    t.join()
assert got == single, "concurrent proofs differ from the single-thread ones"
for ctx, inst, gens, ngens, enc in workers:
    enc.free(); ngens.free(); gens.free(); inst.free(); ctx.close()
print("TSAN_WORKER_OK")
