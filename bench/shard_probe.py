"""What would sharding the Fiat-Shamir-ordered rounds over GPUs cost? (SURVEY 8e / DESIGN.md section 6, the break-even table.)
Measured on ONE GPU:
  * the per-exchange latency of the library's RCCL transport at one rank (H2D of 96 bytes, ncclAllGather, D2H, stream sync) — a lower
    bound for the per-round exchange of partial sums between 8 GPUs;
  * the time of one zero-knowledge sum-check round (bind + next evaluation, sp_sumcheck_bind_eval) on 4 tables of 2^k entries, k = 10..24
    — the work a shard would divide by W;
  * SNARK::prove at 2^s with W = 8 virtual shards on the one GPU (all shards share it, so this shows the ORDER of the exchanges and
    their host-side cost, not a speed-up), with and without the residue-sharded rounds.
Run on the GPU box from the repo root:  python bench/shard_probe.py [log2_size]"""
import ctypes, json, os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from spartan_amd import prover as P, capi
from tests.helpers import mont_bulk, fast_scalars, sz, vp, Q

s = int(sys.argv[1]) if len(sys.argv) > 1 else 20
out = {}
ctx = P.Ctx(0)
P.H.spz_rccl_allgather_probe.restype = ctypes.c_double
ctx.set_commit_shard_rccl(0, 1, P.rccl_unique_id())
out["rccl_1rank_exchange_us"] = {str(b): round(P.H.spz_rccl_allgather_probe(ctx.h, sz(b), ctypes.c_int(2000)), 2) for b in (96, 4096, 65536)}
ctx.set_commit_shard_virtual(1)

raw = ctx.raw()
rng = random.Random(3)
rounds = {}
for k in (10, 12, 14, 16, 18, 20, 22, 24):
    n = 1 << k
    base = mont_bulk(fast_scalars(rng, min(n, 1 << 16)))
    tabs = []
    for _ in range(4):
        t = capi.Table.alloc(capi.Ctx.__new__(capi.Ctx), 1) if False else None
    hs = []
    for _ in range(4):
        h = vp()
        assert capi.lib.sp_table_alloc(raw, sz(n), ctypes.byref(h)) == 0
        for off in range(0, n, 1 << 16):
            assert capi.lib.sp_table_write(raw, h, sz(off), base, sz(min(n, 1 << 16))) == 0
        hs.append(h)
    arr = (vp * 4)(*hs)
    ev = (ctypes.c_uint64 * 12)()
    r = mont_bulk([rng.randrange(Q)])
    best = 1e9
    reps = 5 if k >= 20 else 20
    for _ in range(reps):
        for h in hs:
            capi.lib.sp_table_set_len(h, sz(n))
        t0 = time.perf_counter()
        assert capi.lib.sp_sumcheck_bind_eval(raw, ctypes.c_int(2), arr, sz(4), r, ev) == 0
        best = min(best, (time.perf_counter() - t0) * 1e6)
    rounds[str(k)] = round(best, 1)
    for h in hs:
        capi.lib.sp_table_free(h)
out["zk_round_us_by_log2_len"] = rounds

N = 1 << s
inst = P.Instance.produce_synthetic_r1cs(ctx, N, N, 10, seed=0)
gens = P.SNARKGens(ctx, N, N, 10, N)
enc = P.SNARK.encode(ctx, inst, gens)
tape = P.seed_scalar(b"tape", 100)
run = lambda: P.SNARK.prove(ctx, inst, enc, inst.vars, inst.inputs, gens, b"snark_example", tape)
def timed(k=5):
    run(); t0 = time.perf_counter()
    for _ in range(k): pr = run()
    return (time.perf_counter() - t0) / k * 1e3, pr
ms0, ref = timed()
ctx.set_commit_shard_virtual(8); ctx.shard_stats(reset=True)
ms8, p8 = timed(); st8 = ctx.shard_stats(reset=True)
ctx.set_option("shard.residues", 0)
ctx.set_commit_shard_virtual(8); ctx.shard_stats(reset=True)  # the switch is resolved when the sharding is configured
ms8c, p8c = timed(); st8c = ctx.shard_stats()
ctx.set_option("shard.residues", 1)
ctx.set_commit_shard_virtual(1)
assert p8 == ref and p8c == ref
out["snark_prove_2p%d_ms" % s] = {"unsharded": round(ms0, 2), "8_virtual_shards_commits_only": round(ms8c, 2), "8_virtual_shards_commits_rounds_bound_evaluate": round(ms8, 2),
                                  "exchanges_per_proof": {"commits_only": st8c["gathers"] / 6, "all": st8["gathers"] / 6},
                                  "note": "all shards share ONE GPU: ordering and host-side cost of the exchanges, not a speed-up; proofs byte-identical"}
print(json.dumps(out, indent=1))
