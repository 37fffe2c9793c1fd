"""Worker of tests/test_gpu_shard.py::test_sharded_proving_with_an_os_entropy_tape: one of two lock-step ranks (gloo) on the test box's
single GPU proves with tape_seed=None while the commitments are row-sharded. The ranks must end up with ONE tape (rank 0 draws
the seed, the commit transport distributes it: shard.cc commit_shard_shared_seed): identical proofs, accepted by the oracle's verifier."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, torch.distributed as dist
from spartan_amd import prover as P
from tests import helpers as H

dist.init_process_group(backend="gloo")
rank = dist.get_rank()
s = int(sys.argv[1]); N = 1 << s
ctx = P.Ctx(0)
inst = P.Instance.produce_synthetic_r1cs(ctx, N, N, 10, seed=4)
gens = P.SNARKGens(ctx, N, N, 10, N)
enc = P.SNARK.encode(ctx, inst, gens)
ctx.set_commit_shard(dist, "cpu")
ctx.shard_stats(reset=True)
proofs = [P.SNARK.prove(ctx, inst, enc, inst.vars, inst.inputs, gens, b"snark_example", None) for _ in range(2)]
assert ctx.shard_stats()["gathers"] >= 2, "no commitment went through the transport"
box = [None, None]
dist.all_gather_object(box, proofs)
if rank == 0:
    assert box[0] == box[1], "the ranks proved on different tapes"
    assert proofs[0] != proofs[1], "two proofs with OS entropy must differ"
    orc = H.load_oracle()
    og = H.vp(orc.orc_snark_gens_new(H.sz(N), H.sz(N), H.sz(10), H.sz(N)))
    ops, mem = enc.comm(0), enc.comm(1)
    for pr in proofs:
        rc = orc.orc_snark_verify_bytes(pr, H.sz(len(pr)), og, H.sz(N), H.sz(N), H.sz(10), H.sz(N), H.sz(2 * N), ops, H.sz(len(ops) // 32), mem,
                                        H.sz(len(mem) // 32), inst.inputs, b"snark_example")
        assert rc == 1, "verifier rejected a sharded OS-entropy proof"
    print("SHARD_SEED_OK")
ctx.set_commit_shard(None)
dist.destroy_process_group()
