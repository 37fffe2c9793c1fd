"""One SNARK::prove at 2^s (default 24) on one GPU: generator table widths chosen, memory used, time, proof size; twice, to
check the bytes repeat. The oracle cannot follow at this size (hours of CPU): this is a does-it-fit / does-it-run probe."""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spartan_amd import prover as P

s = int(sys.argv[1]) if len(sys.argv) > 1 else 24
N = 1 << s
ctx = P.Ctx(0)
t0 = time.time(); inst = P.Instance.produce_synthetic_r1cs(ctx, N, N, 10, seed=0); t_inst = time.time() - t0
t0 = time.time(); gens = P.SNARKGens(ctx, N, N, 10, N); t_gens = time.time() - t0
free, total = torch.cuda.mem_get_info(0)
print(f"2^{s}: instance {t_inst:.1f} s, generators {t_gens:.1f} s, window bits sat/eval {gens.window_bits(0)}/{gens.window_bits(1)}, HBM used {(total - free) / 1e9:.1f} GB", flush=True)
t0 = time.time(); enc = P.SNARK.encode(ctx, inst, gens); t_enc = time.time() - t0
va = P.VarsAssignment(ctx, inst.vars)
tape = P.seed_scalar(b"tape", 100)
out = []
for k in range(3):
    tm = {}
    t0 = time.time(); p = P.SNARK.prove(ctx, inst, enc, va, inst.inputs, gens, b"snark_example", tape, tm); dt = time.time() - t0
    out.append(p)
    free, total = torch.cuda.mem_get_info(0)
    print(f"prove #{k}: {dt * 1e3:.1f} ms, {len(p)} bytes, sha256 {hashlib.sha256(p).hexdigest()[:16]}, HBM used {(total - free) / 1e9:.1f} GB, phases {({k_: round(v * 1e3, 1) for k_, v in tm.items()})}", flush=True)
print("encode %.2f s; repeatable: %s; constraints/s %.3g" % (t_enc, out[0] == out[1] == out[2], N / dt))
