"""Worker of tests/test_gpu_shard.py::test_residue_sharded_sumchecks_over_the_process_transport: one of two lock-step ranks (gloo) on the
test box's single GPU. With option shard.residue_transport = 1 each rank keeps one residue class of the ZK sum-check tables and the rounds'
partial sums travel over the commit transport (SURVEY 8e, K3/K4 over real ranks): the proof must equal the one the same rank computed
before sharding was configured, on both ranks, and the transport must have carried the rounds."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch.distributed as dist
from spartan_amd import prover as P

dist.init_process_group(backend="gloo")
rank = dist.get_rank()
s = int(sys.argv[1]); N = 1 << s
ctx = P.Ctx(0)
inst = P.Instance.produce_synthetic_r1cs(ctx, N, N, 10, seed=7)
gens = P.SNARKGens(ctx, N, N, 10, N)
ngens = P.NIZKGens(ctx, N, N, 10)
enc = P.SNARK.encode(ctx, inst, gens)
tape = P.seed_scalar(b"tape", 11)
ref = P.SNARK.prove(ctx, inst, enc, inst.vars, inst.inputs, gens, b"snark_example", tape)
nref = P.NIZK.prove(ctx, inst, inst.vars, inst.inputs, ngens, b"nizk_example", tape)
ctx.set_commit_shard(dist, "cpu")
ctx.shard_stats(reset=True)
got = P.SNARK.prove(ctx, inst, enc, inst.vars, inst.inputs, gens, b"snark_example", tape)
st = ctx.shard_stats()
ngot = P.NIZK.prove(ctx, inst, inst.vars, inst.inputs, ngens, b"nizk_example", tape)
assert got == ref, "sharded SNARK proof differs on rank %d" % rank
assert ngot == nref, "sharded NIZK proof differs on rank %d" % rank
# two commitments + per sharded sum-check: first evaluation, one exchange per round while a class holds >= 4 entries, the hand-back
lw = 1
want_rounds = (s - lw + 1) + (s + 1 - lw + 1)
assert st["gathers"] >= 2 + want_rounds, (st, want_rounds)
if "shard.cubic_min_len" in os.environ.get("SPARTAN_OPTIONS", ""):
    # the batched cubic sum-checks sharded over the two ranks as well (pack -> gather -> scatter hand-over), plus the two chunk-sharded
    # evaluation batches of the hash layer: strictly more exchanges than the ZK sum-checks alone
    assert st["gathers"] >= 2 + want_rounds + 2 + 5, (st, want_rounds)
box = [None, None]
dist.all_gather_object(box, got)
assert box[0] == box[1]
if rank == 0:
    print("RESIDUE_TRANSPORT_OK gathers=%d" % st["gathers"])
ctx.set_commit_shard(None)
dist.destroy_process_group()
