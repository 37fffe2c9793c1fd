"""Worker of tests/test_gpu_shard.py::test_rccl_transport_on_every_visible_gpu: one of WORLD_SIZE lock-step ranks, ONE GPU EACH, the
library's own RCCL transport (ncclAllGather on device buffers over xGMI, spartan_amd/host/shard.cc) carrying everything the sharded proof
exchanges: the row-sharded commitments, the shared tape seed, and — with option shard.residue_transport = 1 — the partial sums and hand-overs of
the residue-sharded ZK and batched cubic sum-checks and the chunk-sharded evaluations. The sharded proofs (SNARK with a fixed tape, NIZK,
and a SNARK with an OS-entropy tape checked for equality ACROSS ranks) must equal the unsharded ones."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist
from spartan_amd import prover as P

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dev = int(os.environ.get("LOCAL_RANK", rank)) % torch.cuda.device_count()
torch.cuda.set_device(dev)
dist.init_process_group(backend="gloo")   # bootstrap only (the unique id, the final comparison): the data path is RCCL inside the library
s = int(sys.argv[1]); N = 1 << s
ctx = P.Ctx(dev)
inst = P.Instance.produce_synthetic_r1cs(ctx, N, N, 10, seed=7)
gens = P.SNARKGens(ctx, N, N, 10, N)
ngens = P.NIZKGens(ctx, N, N, 10)
enc = P.SNARK.encode(ctx, inst, gens)
tape = P.seed_scalar(b"tape", 11)
ref = P.SNARK.prove(ctx, inst, enc, inst.vars, inst.inputs, gens, b"snark_example", tape)
nref = P.NIZK.prove(ctx, inst, inst.vars, inst.inputs, ngens, b"nizk_example", tape)
box = [P.rccl_unique_id() if rank == 0 else None]
dist.broadcast_object_list(box, src=0)
ctx.set_commit_shard_rccl(rank, world, box[0])
ctx.shard_stats(reset=True)
got = P.SNARK.prove(ctx, inst, enc, inst.vars, inst.inputs, gens, b"snark_example", tape)
st = ctx.shard_stats()
assert got == ref, "RCCL-sharded SNARK proof differs on rank %d" % rank
assert P.NIZK.prove(ctx, inst, inst.vars, inst.inputs, ngens, b"nizk_example", tape) == nref, "RCCL-sharded NIZK proof differs on rank %d" % rank
assert st["gathers"] >= 2, st
fresh = P.SNARK.prove(ctx, inst, enc, inst.vars, inst.inputs, gens, b"snark_example", None)   # OS-entropy tape: the ranks must agree on one seed
allp = [None] * world
dist.all_gather_object(allp, fresh)
assert all(p == allp[0] for p in allp) and fresh != ref and len(fresh) == len(ref), "ranks disagree on the proof made with a shared OS-entropy tape"
if rank == 0:
    print("RCCL_MULTI_OK world=%d gathers=%d bytes=%d" % (world, st["gathers"], st["bytes"]))
ctx.set_commit_shard_virtual(1)
enc.free(); gens.free(); ngens.free(); inst.free(); ctx.close()
dist.destroy_process_group()
