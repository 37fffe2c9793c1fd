"""Row-sharded commitments end to end: two lock-step ranks share the single GPU of the test box and
exchange commitment bytes over gloo (RCCL refuses two ranks on one device; on an 8-GPU node the same code runs with the
`nccl` backend). bench.py --shard-commits itself compares every sharded proof with the unsharded bytes."""
import ctypes, json, os, socket, subprocess, sys

import pytest

from tests.helpers import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("s", [10, 14])
def test_sharded_commit_proof_is_byte_identical(s):
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   BENCH_DIST_BACKEND="gloo", BENCH_FORCE_DEVICE="0", BENCH_NO_PROF="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--shard-commits", "--log2-cons", str(s),
                                       "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--concurrent", "0"],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=600)
        assert p.returncode == 0, e[-3000:]
        outs.append(o)
    line = json.loads([l for l in outs[0].splitlines() if l.startswith("{")][-1])
    assert line["scaling"] == "strong" and line["n_gpus"] == 2
    assert "sharded over 2 GPUs" in line["config"]["parallelism"]
    # witness commit (r1csproof.rs:159) + derefs commit (sparse_mlpoly.rs:64-67): 2^(s/2) and 2^((s+3)/2) rows of 32 bytes
    assert line["config"]["all_gathers_per_proof"] == 2
    assert line["config"]["all_gather_bytes_per_proof"] == 32 * ((1 << (s // 2)) + (1 << ((s + 3) // 2)))
    assert line["value"] == pytest.approx((1 << s) / (line["ms_per_step"] * 1e-3), rel=1e-6)  # one proof, not two


def test_default_multi_rank_run_reports_replicas_and_strong_leg():
    """`python bench.py --gpus 2` with NO launcher and no RANK / WORLD_SIZE in the environment, as the driver starts its N=1 run:
    bench.py launches its own two ranks (torch.distributed.run on 127.0.0.1; two gloo ranks on the one GPU of the box). The headline
    is the replica throughput ("weak": one independent proof per rank, no data-path collective) and the same command adds the
    strong-scaling leg at every size of --strong-log2 (one proof, row commitments sharded, byte-identical to the unsharded proof),
    each with its exchange counts and the Amdahl ceiling of its own unsharded run."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(BENCH_DIST_BACKEND="gloo", BENCH_FORCE_DEVICE="0", BENCH_NO_PROF="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--log2-cons", "14", "--steps", "2", "--warmup", "1",
                        "--strong-log2", "12,14"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                                                              # rank 0 alone prints
    line = json.loads(lines[0])
    assert line["scaling"] == "weak" and line["n_gpus"] == 2 and line["n_ranks_seen"] == 2
    assert line["value"] == pytest.approx(2 * (1 << 14) / (line["ms_per_step"] * 1e-3), rel=1e-6)  # two proofs per step
    assert [st.get("log2_cons") for st in line["strong"]] == [12, 14]
    for st in line["strong"]:
        s = st["log2_cons"]
        assert "error" not in st, st
        assert st["scaling"] == "strong" and st["byte_identical_to_unsharded"] and st["all_gathers_per_proof"] == 2
        assert st["all_gather_bytes_per_proof"] == 32 * ((1 << (s // 2)) + (1 << ((s + 3) // 2)))
        assert st["value"] == pytest.approx((1 << s) / (st["ms_per_step"] * 1e-3), rel=1e-6)         # one proof per step
        assert st["amdahl"]["ceiling_commits_only"] >= 1.0 and st["ms_per_step_unsharded"] > 0


@pytest.mark.parametrize("s,nshards", [(14, 8), (16, 4)])
def test_virtual_shards_on_one_gpu_are_byte_identical(s, nshards):
    """SURVEY §8e "Test reality": W row shards on one physical GPU (W sub-contexts with their own streams, in-process gather).
    Partition, blind offsets and result layout are the code the RCCL path runs; the proof must not change."""
    from spartan_amd import prover as P
    N = 1 << s
    ctx = P.Ctx(0)
    inst = P.Instance.produce_synthetic_r1cs(ctx, N, N, 10, seed=s)
    gens = P.SNARKGens(ctx, N, N, 10, N)
    enc = P.SNARK.encode(ctx, inst, gens)
    tape = P.seed_scalar(b"tape", s)
    ref = P.SNARK.prove(ctx, inst, enc, inst.vars, inst.inputs, gens, b"snark_example", tape)
    ctx.set_commit_shard_virtual(nshards)
    ctx.shard_stats(reset=True)
    got = P.SNARK.prove(ctx, inst, enc, inst.vars, inst.inputs, gens, b"snark_example", tape)
    st = ctx.shard_stats()
    assert got == ref
    # exchanges of one proof: the two row-sharded commitments (witness, derefs) + the residue-sharded sum-checks of R1CSProof::prove
    # (one partial-sum gather per round while the shards hold >= 2 entries, one hand-back of the survivors: phase one
    # s - lw + 1, phase two s - lw + 2) + the row-sharded `bound` of the four openings + the chunked evaluate of the witness
    # + (round 4) the chunk-sharded <chi, T_k> evaluations of HashLayerProof::prove (two exchanges) and the residue-sharded rounds of the
    # batched cubic sum-checks whose tables are long enough (test_batched_cubic_sumchecks_shard_by_residue counts those exactly)
    lw = nshards.bit_length() - 1
    assert st["gathers"] >= 2 + (s - lw + 1) + (s - lw + 2) + 4 + 1 + 2
    assert st["bytes"] > 32 * ((1 << (s // 2)) + (1 << ((s + 3) // 2)))
    enc2 = P.SNARK.encode(ctx, inst, gens)    # SNARK::encode's multi_commit shards the same way
    assert enc2.serialize_commitment() == enc.serialize_commitment()
    ctx.set_commit_shard_virtual(1)
    assert P.SNARK.prove(ctx, inst, enc, inst.vars, inst.inputs, gens, b"snark_example", tape) == ref
    enc2.free(); enc.free(); gens.free(); inst.free(); ctx.close()


@pytest.mark.parametrize("s,nshards,min_len", [(10, 4, 64), (12, 8, 128)])
def test_batched_cubic_sumchecks_shard_by_residue(s, nshards, min_len):
    """SURVEY 8e for the phase that dominates the proof (ProductCircuitEvalProofBatched::prove -> prove_cubic_batched, src/product_tree.rs:259-383,
    src/sumcheck.rs:254-424): every table of a batch split by index residue over W virtual shards, the throughput-sized rounds run per shard
    with 96 * ninst bytes of partial evaluations exchanged per round, then the sub-tables are packed, gathered and scattered back
    (sp_tables_pack / sp_tables_unpack_residues) and the latency-sized rounds continue unsharded. Option shard.cubic_min_len lowers the
    hand-over length (8192 by default) so that a 2^10 / 2^12 instance has sharded rounds. Bytes must equal the unsharded proof's, and the
    exchange count must be the formula's."""
    from spartan_amd import prover as P
    N = 1 << s
    ctx = P.Ctx(0)
    inst = P.Instance.produce_synthetic_r1cs(ctx, N, N, 10, seed=s)
    gens = P.SNARKGens(ctx, N, N, 10, N)
    enc = P.SNARK.encode(ctx, inst, gens)
    tape = P.seed_scalar(b"tape", s)
    ref = P.SNARK.prove(ctx, inst, enc, inst.vars, inst.inputs, gens, b"snark_example", tape)
    ctx.set_option("testing.unlock", 1)
    ctx.set_option("shard.cubic_min_len", 1073741824)   # no batched sum-check is long enough: the round-3 exchanges alone
    try:
        ctx.set_commit_shard_virtual(nshards)
        ctx.shard_stats(reset=True)
        assert P.SNARK.prove(ctx, inst, enc, inst.vars, inst.inputs, gens, b"snark_example", tape) == ref
        base = ctx.shard_stats(reset=True)["gathers"]
        ctx.set_option("shard.cubic_min_len", min_len)
        assert P.SNARK.prove(ctx, inst, enc, inst.vars, inst.inputs, gens, b"snark_example", tape) == ref
        got = ctx.shard_stats(reset=True)["gathers"]
    finally:
        ctx.set_option("shard.cubic_min_len", 0)
    # Two product-circuit batches per proof (row/col layers of the ops circuits: leaves 2^s; of the memory circuits: leaves 2^(s+1)); a
    # layer sum-check over tables of L = 2^k * min_len entries (k >= 2) costs 1 first evaluation + (k + 1) binds + 1 hand-back exchanges
    def per_circuit(leaves):
        n, L = 0, leaves // 2
        while L >= 4 * min_len:
            n += (L // min_len).bit_length() - 1 + 3
            L //= 2
        return n
    want = per_circuit(N) + per_circuit(2 * N)
    assert want > 0 and got - base == want, (got, base, want)
    ctx.set_commit_shard_virtual(1)
    enc.free(); gens.free(); inst.free(); ctx.close()


def test_residue_shards_can_be_switched_off_and_nizk_matches():
    """NIZK::prove under 8 virtual shards (residue-sharded sum-checks, row-sharded bound, chunked evaluate) equals the unsharded
    proof; option shard.residues = 0 keeps only the commitment sharding (the A/B switch of DESIGN.md section 6)."""
    from spartan_amd import prover as P
    s = 14
    N = 1 << s
    ctx = P.Ctx(0)
    inst = P.Instance.produce_synthetic_r1cs(ctx, N, N, 10, seed=2)
    inst.set_digest(b"d")
    gens = P.NIZKGens(ctx, N, N, 10)
    tape = P.seed_scalar(b"tape", 9)
    ref = P.NIZK.prove(ctx, inst, inst.vars, inst.inputs, gens, b"nizk_example", tape)
    ctx.set_commit_shard_virtual(8)
    ctx.shard_stats(reset=True)
    assert P.NIZK.prove(ctx, inst, inst.vars, inst.inputs, gens, b"nizk_example", tape) == ref
    full = ctx.shard_stats(reset=True)["gathers"]
    ctx.set_option("shard.residues", 0)
    try:
        ctx.set_commit_shard_virtual(8)  # the switch is resolved when the sharding is configured (and compared across ranks there), not per proof
        ctx.shard_stats(reset=True)
        assert P.NIZK.prove(ctx, inst, inst.vars, inst.inputs, gens, b"nizk_example", tape) == ref
        assert ctx.shard_stats()["gathers"] == 1 and full == 1 + (s - 3 + 1) + (s - 3 + 2) + 1 + 1   # commit | + sum-checks, bound, evaluate
    finally:
        ctx.set_option("shard.residues", 1)
    ctx.set_commit_shard_virtual(1)
    gens.free(); inst.free(); ctx.close()


@pytest.mark.parametrize("s", [10, 6])
def test_sharded_proving_with_an_os_entropy_tape(s):
    """tape_seed=None (the production setting) with lock-step ranks: rank 0 draws the RandomTape seed and the commit transport
    hands it to the others — otherwise every rank would blind its row slice with its own tape (round-2 advisor finding).
    s = 10: the witness commitment is sharded by rows (16 per rank); s = 6: by columns (8 rows: partial points gathered and added),
    both over the callback transport of two real processes."""
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "shard_seed_worker.py"), str(s)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=600)
        assert p.returncode == 0, e[-3000:]
        outs.append(o)
    assert "SHARD_SEED_OK" in outs[0]


def test_rccl_transport_inside_the_library_single_rank():
    """the RCCL path of spartan_amd/host/shard.cc on the one GPU of the box: librccl.so is dlopen'ed, rank 0 draws the
    ncclUniqueId, joins a 1-rank communicator, and every commitment goes through ncclAllGather on device buffers."""
    from spartan_amd import prover as P
    s = 12
    N = 1 << s
    ctx = P.Ctx(0)
    inst = P.Instance.produce_synthetic_r1cs(ctx, N, N, 10, seed=1)
    gens = P.SNARKGens(ctx, N, N, 10, N)
    enc = P.SNARK.encode(ctx, inst, gens)
    tape = P.seed_scalar(b"tape", 2)
    ref = P.SNARK.prove(ctx, inst, enc, inst.vars, inst.inputs, gens, b"snark_example", tape)
    uid = P.rccl_unique_id()
    assert len(uid) == 128 and any(uid)
    ctx.set_commit_shard_rccl(0, 1, uid)
    ctx.shard_stats(reset=True)
    assert P.SNARK.prove(ctx, inst, enc, inst.vars, inst.inputs, gens, b"snark_example", tape) == ref
    # with one rank the witness commitment keeps its overlapped single-GPU path; the derefs commitment goes through RCCL
    assert ctx.shard_stats()["gathers"] >= 1
    ctx.set_commit_shard_virtual(1)
    enc.free(); gens.free(); inst.free(); ctx.close()


def test_column_sharded_commitment_of_a_small_instance_is_byte_identical():
    """SURVEY §8e, the north-star's "partial sums": a commitment with fewer rows than a shard is worth (2^6 constraints: the witness is
    8 rows x 8 columns) is sharded by COLUMNS — every shard sums its slice of the generators into one partial point per row
    (sp_commit_rows_partial), the points are gathered and added, the blind terms added and the sums encoded
    (sp_host_points_sum_encode). Same points, same bytes; option shard.cols = 0 is the single-GPU path for comparison."""
    from spartan_amd import prover as P
    s = 6
    N = 1 << s
    ctx = P.Ctx(0)
    inst = P.Instance.produce_synthetic_r1cs(ctx, N, N, 10, seed=s)
    gens = P.SNARKGens(ctx, N, N, 10, N)
    ngens = P.NIZKGens(ctx, N, N, 10)
    enc = P.SNARK.encode(ctx, inst, gens)
    tape = P.seed_scalar(b"tape", s)
    ref = P.SNARK.prove(ctx, inst, enc, inst.vars, inst.inputs, gens, b"snark_example", tape)
    nref = P.NIZK.prove(ctx, inst, inst.vars, inst.inputs, ngens, b"nizk_example", tape)
    ctx.set_commit_shard_virtual(4)
    ctx.shard_stats(reset=True)
    assert P.SNARK.prove(ctx, inst, enc, inst.vars, inst.inputs, gens, b"snark_example", tape) == ref
    st = ctx.shard_stats()
    assert st["gathers"] >= 1 and st["bytes"] >= 4 * 8 * 128          # the witness commitment: 4 shards x 8 rows x one 128-byte point
    assert P.NIZK.prove(ctx, inst, inst.vars, inst.inputs, ngens, b"nizk_example", tape) == nref
    ctx.set_commit_shard_virtual(1)
    enc.free(); gens.free(); ngens.free(); inst.free(); ctx.close()


def test_commit_rows_partial_and_sum_encode_equal_the_whole_commitment():
    """the two entry points of the column-sharded form against sp_commit_rows_dev: 5 rows x 96 columns in three slices of 32 columns,
    blind terms through sp_host_commit_point"""
    import hashlib
    import numpy as np
    from spartan_amd import capi
    from tests.helpers import sz, vp
    ctx = capi.Ctx(0)
    rows, cols, W = 5, 96, 3
    g = capi.Gens(ctx, uniform=hashlib.shake_256(b"cols_partial").digest(64 * (cols + 1)))
    rng = np.random.default_rng(9)
    Z = rng.integers(0, 2**64, size=(rows * cols, 4), dtype=np.uint64); Z[:, 3] &= np.uint64((1 << 60) - 1)
    B = rng.integers(0, 2**64, size=(rows, 4), dtype=np.uint64); B[:, 3] &= np.uint64((1 << 60) - 1)
    zp = Z.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)); bp = B.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))
    t = capi.Table.upload(ctx, zp, rows * cols)
    want = (ctypes.c_uint8 * (32 * rows))()
    assert capi.lib.sp_commit_rows_dev(ctx.h, g.h, sz(0), sz(cols), t.h, sz(0), sz(rows), sz(cols), bp, want) == 0
    pts = (ctypes.c_uint64 * (16 * rows * (W + 1)))()
    base = ctypes.addressof(pts)
    per = cols // W
    for k in range(W):
        assert capi.lib.sp_commit_rows_partial(ctx.h, g.h, sz(k * per), t.h, sz(k * per), sz(cols), sz(rows), sz(per), ctypes.c_void_p(base + 128 * rows * k)) == 0
    hidx = (ctypes.c_uint32 * 1)(cols)
    for r in range(rows):
        assert capi.lib.sp_host_commit_point(g.h, hidx, sz(1), ctypes.cast(ctypes.c_void_p(B.ctypes.data + 32 * r), ctypes.POINTER(ctypes.c_uint64)),
                                             ctypes.c_void_p(base + 128 * (rows * W + r))) == 0
    got = (ctypes.c_uint8 * (32 * rows))()
    assert capi.lib.sp_host_points_sum_encode(ctypes.c_void_p(base), sz(W + 1), sz(rows), got) == 0
    assert bytes(got) == bytes(want)
    t.free(); g.free(); ctx.close()


def test_residue_sharded_sumchecks_over_the_process_transport():
    """SURVEY §8e K3/K4 over REAL ranks (not only virtual shards): two processes, each keeping one residue class of the ZK sum-check
    tables, the rounds' partial sums (96 bytes per rank) and the final hand-back over the callback transport. Opt-in
    (option shard.residue_transport = 1: at 2^20 the exchange costs more than the round); the proofs equal the unsharded ones."""
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   SPARTAN_OPTIONS="testing.unlock=1,shard.residue_transport=1,shard.cubic_min_len=128")  # (round 4) the batched cubic sum-checks of SPARK and the hash layer's evaluations shard over the two ranks too
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "residue_transport_worker.py"), "12"], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=600)
        assert p.returncode == 0, e[-3000:]
        outs.append(o)
    assert "RESIDUE_TRANSPORT_OK" in outs[0]


def test_rccl_transport_on_every_visible_gpu():
    """Armed for the first multi-GPU box (SCALE runs): world = torch.cuda.device_count() lock-step ranks, one GPU each, the library's RCCL
    transport over xGMI carrying the sharded commitments, the agreed tape seed and — shard.residue_transport = 1, a lowered hand-over length —
    every residue-sharded sum-check (ZK and batched cubic) and the chunk-sharded evaluations. Skips on a single-GPU box, where the same
    transport has only ever been driven with a 1-rank communicator (test_rccl_transport_inside_the_library_single_rank)."""
    import torch
    world = torch.cuda.device_count()
    if world < 2:
        pytest.skip("needs >= 2 visible GPUs (one rank per GPU over RCCL)")
    world = 1 << (world.bit_length() - 1)   # residue classes want a power of two
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   SPARTAN_OPTIONS="testing.unlock=1,shard.residue_transport=1,shard.cubic_min_len=256", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "rccl_worker.py"), "14"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=900)
        assert p.returncode == 0, e[-3000:]
        outs.append(o)
    assert "RCCL_MULTI_OK world=%d" % world in outs[0]
