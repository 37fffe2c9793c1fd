#!/usr/bin/env python3
"""bench.py — R1CS constraints/sec in SNARK::prove on Instance::produce_synthetic_r1cs (BASELINE.json metric).

A "step" is one SNARK::prove (lib.rs:339-420) over a resident synthetic instance: the instance, the generators
(with their window tables) and the computation commitment/decommitment (SNARK::encode) are built before the timed
region; the timed region covers everything SNARK::prove does: the H2D copy of the assignment `vars`
(handed over as a host buffer, 32 B per variable), all transcript work on the host and the D2H of every commitment.
The same proof from an assignment already uploaded by VarsAssignment::new is timed right after and reported beside
it (`config.resident_assignment`), never as `value`. "bit-exact" everywhere in the line means: equal to the bytes of
the in-repo oracle (oracle/, the CPU restatement of the reference prover) on the same instance and tape.

Multi-GPU (--gpus N, launched with torch.distributed.run, one rank per GPU): each rank proves its own
independent instance (different seed) — proofs are independent units, so there is no data-path collective;
"scaling": "weak". Timing is bracketed by barrier + synchronize, and the MAX over ranks is used.

--shard-commits (optional, N > 1): all ranks prove the SAME instance in lock-step and every DensePolynomial::commit is
row-sharded over the ranks, with one RCCL all-gather of the 32-byte commitments per commit;
"scaling": "strong", `value` = one proof's constraints / time. Each sharded proof is checked against the unsharded bytes.

Prints ONE JSON line on rank 0.
"""
import argparse, ctypes, json, os, sys, time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


def self_launch(n_gpus):
    """`python bench.py --gpus N` started WITHOUT a launcher (no WORLD_SIZE in the environment) re-executes itself under
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N` on 127.0.0.1 with a free port, one rank per GPU, and exits with
    the launcher's status: a plain `python3 bench.py --gpus 8` must never print a 1-GPU line (missing #3)."""
    import socket, subprocess
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("[bench] --gpus %d without a launcher: re-executing as %s" % (n_gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    raise SystemExit(subprocess.call(cmd, env=dict(os.environ, BENCH_SELF_LAUNCHED="1")))


def dist_setup(n_gpus):
    """returns (rank, world, dist or None). Reads RANK/WORLD_SIZE/MASTER_* from the env (torch.distributed.run); with --gpus N > 1
    and no launcher around it, launches one (self_launch). The job REFUSES to run when the launcher's world size is not --gpus."""
    if n_gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(n_gpus)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world != n_gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks: refusing to print a line for another job" % (n_gpus, world))
    if world == 1:
        return 0, 1, None
    import torch.distributed as dist
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")  # nccl == RCCL on ROCm; gloo for the CPU tests
    if backend == "nccl":
        import torch
        dev = int(os.environ.get("BENCH_FORCE_DEVICE", os.environ.get("LOCAL_RANK", str(rank))))
        torch.cuda.set_device(dev % max(1, torch.cuda.device_count()))
    dist.init_process_group(backend=backend)
    return rank, world, dist


def dist_barrier(dist):
    if dist is None:
        return
    if dist.get_backend() == "nccl":  # name the device: RCCL otherwise guesses it from the rank
        import torch
        dist.barrier(device_ids=[torch.cuda.current_device()])
    else:
        dist.barrier()


def dist_max(dist, value, device="cpu"):
    """MAX over ranks of a python float."""
    if dist is None:
        return value
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def dist_sum(dist, value, device="cpu"):
    if dist is None:
        return value
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


PUBLISHED_CPU_CPS = 26797.0  # README.md:375: SNARK::prove 2^20 in 39.130 s on one core of an i7-1065G7 (dalek SIMD backend)
README_US_PER_SCALAR = {"polycommit": 2.59, "commit_nondet_witness": 1.72}  # README.md:354,367 (2^20 and 2^23 committed scalars)


def cpu_baseline(log2_cons, threads=1, want_digest=False):
    """The oracle (CPU restatement of the reference prover, oracle/) timed on this box's host cores: SNARK::prove at 2^log2_cons
    constraints on `threads` OpenMP threads (the reference's `multicore` feature parallelises the same loops: dense_mlpoly.rs:148-162).
    Checker-side only; never the product path. want_digest: also return the SHA-256 of the oracle's proof bytes (same instance seed 0 and
    tape seed 100 as the GPU proof of rank 0), so that the timed GPU proof is compared with an oracle proof computed IN THIS RUN."""
    import hashlib
    from tests import helpers as H
    orc = H.load_oracle()
    N = 1 << log2_cons
    # the CPU baseline gets every core of the box: the library narrows this (the proving) thread's affinity to the GPU's NUMA node (option
    # host.pin_thread) and OpenMP threads created from here would inherit that mask
    mask = None
    try:
        mask = os.sched_getaffinity(0)
        os.sched_setaffinity(0, range(os.cpu_count()))
    except (AttributeError, OSError):
        mask = None
    orc.orc_set_threads(ctypes.c_int(threads))
    inst = H.vp(orc.orc_instance_synthetic(H.sz(N), H.sz(N), H.sz(10), ctypes.c_uint64(0)))
    g = H.vp(orc.orc_snark_gens_new(H.sz(N), H.sz(N), H.sz(10), H.sz(N)))
    e = H.vp(orc.orc_snark_encode(inst, g))
    seed = (ctypes.c_uint64 * 4)()
    orc.orc_seed_scalar(b"tape", ctypes.c_uint64(100 if want_digest else 0), seed)
    tm = (ctypes.c_double * 10)()
    t0 = time.time()
    p = H.vp(orc.orc_snark_prove(inst, g, e, b"snark_example", seed, tm))
    dt = time.time() - t0
    digest = None
    if want_digest:
        n = orc.orc_proof_bytes(p, None, H.sz(0)); b = (ctypes.c_uint8 * n)(); orc.orc_proof_bytes(p, b, H.sz(n))
        digest = hashlib.sha256(bytes(b)).hexdigest()
    orc.orc_proof_free(p); orc.orc_encode_free(e); orc.orc_snark_gens_free(g); orc.orc_instance_free(inst)
    orc.orc_set_threads(ctypes.c_int(1))
    if mask is not None:
        try:
            os.sched_setaffinity(0, mask)
        except OSError:
            pass
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    out = {"value": N / dt, "unit": "constraints/s", "cores": threads, "kind": "port",
           "sample": f"oracle SNARK::prove, produce_synthetic_r1cs 2^{log2_cons}, {dt:.2f} s on {threads} thread(s) of {model} ({os.cpu_count()} logical cores)"}
    if digest:
        out["proof_sha256"] = digest
    if tm[0] > 0 and tm[6] > 0:
        out["us_per_scalar"] = {"polycommit": tm[0] / N * 1e6, "commit_nondet_witness": tm[6] / (8 * N) * 1e6, "readme_i7_1065G7": README_US_PER_SCALAR, "threads": threads}
    return out


def measured_ceilings():
    """ALU ceilings of the library's own field / curve arithmetic on this GPU (bench/ubench_fpmul --json, ~1 s): dependent
    chains of fp_mul, fq_mul, pt_madd at full occupancy. None if the binary was not built."""
    exe = os.path.join(ROOT, "bench", "ubench_fpmul")
    if not os.path.exists(exe):
        return None
    import subprocess
    try:
        out = subprocess.run([exe, "--json"], capture_output=True, text=True, timeout=120).stdout.strip().splitlines()
        return json.loads(out[-1])
    except Exception:  # noqa: BLE001  (no ceilings is reported as null, never guessed)
        return None


def gather_ceiling(table_gb, sub_entries, stride):
    """Gather ceiling of the row MSM's access pattern on this GPU (bench/gather_probe --json, ~2 s): 96-byte table entries at the table's
    stride, 64 lanes per (point, window) sub-table, 3 waves per SIMD, no arithmetic. table_gb / sub_entries: the two generator streams."""
    exe = os.path.join(ROOT, "bench", "gather_probe")
    if not os.path.exists(exe):
        return None
    import subprocess
    try:
        cmd = [exe, "--json", str(stride), "%.1f" % table_gb[0], str(sub_entries[0]), "%.1f" % table_gb[1], str(sub_entries[1])]
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=180).stdout.strip().splitlines()
        return json.loads(out[-1])
    except Exception:  # noqa: BLE001
        return None


def kernel_source_digest():
    """identifies the kernel sources a PMC traffic file was collected with (profiles/pmc_traffic.json carries the same digest)"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "spartan_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".hpp")):
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def config_key(get_option, options_table, log2_cons):
    """identifies the configuration a PMC traffic entry belongs to: kernel sources + instance size + every library option that is not at
    its default (diagnostic switches excluded). profiles/pmc_traffic.json is keyed by it; a run whose key has no entry reports traffic null."""
    skip = {"testing.unlock", "host.callstats", "debug.ktime"}
    nd = sorted("%s=%d" % (k, get_option(k)) for k, d, _lo, _hi, _t, _doc in options_table() if k not in skip and get_option(k) != d)
    return "%s|2^%d|%s" % (kernel_source_digest(), log2_cons, ",".join(nd) or "defaults")


def union_ms(spans):
    """measure of the union of [t0, t1) intervals"""
    tot, end = 0.0, None
    for t0, t1 in sorted(spans):
        if end is None or t0 > end:
            tot += t1 - t0; end = t1
        elif t1 > end:
            tot += t1 - end; end = t1
    return tot


def read_spans_raw(capi, raw, family, cap=4096):
    """[(rows, cols, background, t0_ms, t1_ms, issued mixed additions)] of the family's launches since the last sp_prof_reset"""
    sh = (ctypes.c_uint64 * cap)(); t0 = (ctypes.c_double * cap)(); t1 = (ctypes.c_double * cap)(); iss = (ctypes.c_double * cap)()
    k = capi.lib.sp_prof_read_spans(raw, family.encode(), sh, t0, t1, iss, ctypes.c_int(cap))
    return [((int(sh[i]) >> 32) & 0x7fffffff, int(sh[i]) & 0xffffffff, bool(int(sh[i]) >> 63), t0[i], t1[i], iss[i]) for i in range(max(0, min(k, cap)))]


def concurrent_throughput(P, device, s, K, steps):
    """K independent SNARK::prove streams on ONE GPU (own context + host thread each; generator tables are shared). A
    single proof is a chain of latency-bound launches, so a second and a third proof fill the gaps of the first. Measured at 2^20
    (bench/concurrent_probe.py, profiles/r4_concurrent_probe.txt): K = 1 / 2 / 3 / 4 -> 42.5 / 60.9 / 68.9 / 64.0 M constraints/s;
    admitting one proof at a time to the throughput-bound part (option host.proof_gate = 1) changes nothing, so the MSMs colliding is not
    what limits it. Serving-style throughput; reported next to, never instead of, the single-proof `value`. Runs K' = 2 .. K."""
    import threading
    N = 1 << s
    workers = []
    for k in range(K):
        ctx = P.Ctx(device)
        inst = P.Instance.produce_synthetic_r1cs(ctx, N, N, 10, seed=1000 + k)
        gens = P.SNARKGens(ctx, N, N, 10, N)
        enc = P.SNARK.encode(ctx, inst, gens)
        workers.append((ctx, inst, gens, enc, P.seed_scalar(b"tape", 1000 + k)))

    def run(w, n):
        for _ in range(n):
            P.SNARK.prove(w[0], w[1], w[3], w[1].vars, w[1].inputs, w[2], b"snark_example", w[4])
    for w in workers:
        run(w, 1)
    res = {}
    for kk in range(2, K + 1):
        ths = [threading.Thread(target=run, args=(w, steps)) for w in workers[:kk]]
        t0 = time.perf_counter()
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        dt = time.perf_counter() - t0
        res[kk] = {"value": kk * steps * N / dt, "ms_per_proof_slot": dt / steps * 1e3}
    for w in workers:
        w[3].free(); w[2].free(); w[1].free(); w[0].close()
    best = max(res, key=lambda kk: res[kk]["value"])
    return {"proofs_in_flight": best, "value": res[best]["value"], "unit": "constraints/s", "ms_per_proof_slot": res[best]["ms_per_proof_slot"],
            "by_proofs_in_flight": {str(kk): round(v["value"]) for kk, v in res.items()},
            "note": "K host threads, one sp_ctx each, same GPU; each proof byte-identical to its single-stream run; the best K of 2 .. %d is reported" % K}


def side_metrics(P, ctx, inst, gens, N, s, tape_seed, steps):
    """NIZK::prove (lib.rs:501-546) on the same instance, and SNARK::encode (lib.rs:320-335; the sparse_mlpoly commit path of
    BASELINE config 5), timed the same way as the headline; reported next to it, never as `value`."""
    import torch
    ng = P.NIZKGens(ctx, N, N, 10)
    inst.set_digest(b"bench-shape-digest")  # the zlib R1CSShapeDigest is an opaque input here (r1cs.rs:154-158)
    P.NIZK.prove(ctx, inst, inst.vars, inst.inputs, ng, b"nizk_example", tape_seed)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        pr = P.NIZK.prove(ctx, inst, inst.vars, inst.inputs, ng, b"nizk_example", tape_seed)
    dt = (time.perf_counter() - t0) / steps
    ng.free()
    out = {"nizk_prove": {"value": N / dt, "unit": "constraints/s", "ms_per_proof": dt * 1e3, "proof_bytes": len(pr)}}
    t0 = time.perf_counter()
    for _ in range(2):
        e = P.SNARK.encode(ctx, inst, gens); e.free()
    dt = (time.perf_counter() - t0) / 2
    out["snark_encode"] = {"ms": dt * 1e3, "note": "SNARK::encode: dense representation from the entry-order copies on the device, AddrTimestamps::new + two multi_commits on the device"}
    # BASELINE config 5's report: the sparse_mlpoly commit path (SparseMatPolynomial::multi_commit, sparse_mlpoly.rs:483-503 — comb_ops and
    # comb_mem: addresses, timestamps and values, most of them a few bits long) against the HBM roofline, from one instrumented encode
    from spartan_amd import capi
    raw = ctx.raw()
    capi.lib.sp_prof_reset(raw); capi.lib.sp_prof_select(raw, b"msm_rows_fixed"); capi.lib.sp_prof_enable(raw, ctypes.c_int(1))
    e = P.SNARK.encode(ctx, inst, gens); e.free()
    capi.lib.sp_prof_enable(raw, ctypes.c_int(0))
    sp_ = read_spans_raw(capi, raw, "msm_rows_fixed")
    capi.lib.sp_prof_select(raw, None); capi.lib.sp_prof_reset(raw)
    if sp_:
        busy = union_ms([(a, b) for _r, _c, _bg, a, b, _i in sp_])
        scal = sum(r * c for r, c, _bg, _a, _b, _i in sp_); rows = sum(r for r, _c, _bg, _a, _b, _i in sp_)
        issued = sum(i for *_x, i in sp_)
        nwin = gens.windows(1)
        byts = 32.0 * scal + 32.0 * rows
        out["snark_encode"]["roofline"] = {
            "bound": "hbm", "kernel": "msm_rows_fixed (multi_commit of comb_ops and comb_mem)", "launches": [f"{r} x {c}" for r, c, *_x in sp_],
            "committed_scalars": scal, "alg_bytes": byts, "busy_ms": round(busy, 4), "achieved": round(byts / busy / 1e6, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(byts / busy / 1e6 / HBM_PEAK_GBS, 6),
            "additions_per_scalar_issued": round(issued / scal, 3) if issued else None, "additions_per_scalar_full_size": nwin,
            "G_additions_issued_per_s": round(issued / busy / 1e6, 2) if issued else None,
            "note": "32 B per committed scalar + 32 B per row (SURVEY 8d) over the union of the launches' intervals; addresses and timestamps are a few bits long, so the wavefronts leave a scalar after its first windows: additions_per_scalar_issued counts the mixed additions really performed (64 per issued tile)"}
    return out


def enable_sharding(P, ctx, dist, rank, world):
    """row-shard every commitment of `ctx` over the ranks: RCCL inside the library when the job runs on the nccl backend
    (rank 0 draws the ncclUniqueId, one broadcast hands it to everyone), the torch.distributed byte gather otherwise (gloo tests)"""
    if dist.get_backend() == "nccl":
        box = [P.rccl_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        ctx.set_commit_shard_rccl(rank, world, box[0])
        return "rccl: ncclAllGather on device buffers inside libspartan_host.so"
    ctx.set_commit_shard(dist, "cpu")
    return "torch.distributed all_gather of bytes (%s), via callback" % dist.get_backend()


def strong_scaling_leg(P, ctx, dist, rank, world, s, steps, dev, partial=None):
    """ONE proof over all ranks at 2^s constraints (lock-step; commitments row-sharded, and from 2^22-entry tables on also the
    sum-check rounds / bound / evaluate by index residue: DESIGN.md, multi-GPU), measured in the same command as the replica
    throughput. Every rank proves the seed-0 instance; the sharded bytes must equal the unsharded ones. Reports, per size: the
    unsharded and sharded ms per proof, the exchanges and bytes per proof, and the Amdahl ceiling the design predicts from the
    UNSHARDED run's own phase times (the shardable part at this size over W ranks)."""
    import torch
    N = 1 << s
    t_setup = time.perf_counter()
    inst = P.Instance.produce_synthetic_r1cs(ctx, N, N, 10, seed=0)
    gens = P.SNARKGens(ctx, N, N, 10, N)
    enc = P.SNARK.encode(ctx, inst, gens)
    t_setup = time.perf_counter() - t_setup
    tape = P.seed_scalar(b"tape", 100)
    ph = {}
    run = lambda tm=None: P.SNARK.prove(ctx, inst, enc, inst.vars, inst.inputs, gens, b"snark_example", tape, tm)
    is_nccl = dist.get_backend() == "nccl"
    ref = run()
    dist_barrier(dist); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        run(ph)
    torch.cuda.synchronize(); dist_barrier(dist)
    dt1 = dist_max(dist, time.perf_counter() - t0, dev if is_nccl else "cpu") / steps
    transport = enable_sharding(P, ctx, dist, rank, world)
    if run() != ref:
        raise RuntimeError("sharded proof differs from the unsharded proof")
    ctx.shard_stats(reset=True)
    dist_barrier(dist); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        if run() != ref:
            raise RuntimeError("sharded proof differs from the unsharded proof")
    torch.cuda.synchronize(); dist_barrier(dist)
    dt = dist_max(dist, time.perf_counter() - t0, dev if is_nccl else "cpu")
    st = ctx.shard_stats()
    ctx.set_commit_shard_virtual(1)  # clears the sharding (and leaves the RCCL communicator)
    enc.free(); gens.free(); inst.free()
    # Amdahl (multi-GPU): what shards at this size, from the unsharded run's own spans. The row commitments always
    # (polycommit + commit_nondet_witness); from 2^22-entry tables (option shard.residue_min_log2) the throughput-sized rounds too — those are
    # not separable from the latency-sized rounds in the span times, so the second figure is an upper bound (all of the sum-check spans).
    commits = ph.get("polycommit", 0.0) + ph.get("commit_nondet_witness", 0.0)
    rounds = ph.get("prove_sc_phase_one", 0.0) + ph.get("prove_sc_phase_two", 0.0) + ph.get("evalproof_layered_network", 0.0)
    f = 1.0 - 1.0 / world
    amdahl = {"shardable_ms_commits": round(commits * 1e3, 3), "ceiling_commits_only": round(dt1 / max(dt1 - commits * f, 1e-9), 3)}
    if s >= 22:
        amdahl["shardable_ms_rounds_upper_bound"] = round(rounds * 1e3, 3)
        amdahl["ceiling_commits_and_rounds_upper_bound"] = round(dt1 / max(dt1 - (commits + rounds) * f, 1e-9), 3)
    amdahl["note"] = ("unsharded ms / (unsharded ms - shardable ms x (1 - 1/W)), spans of the unsharded run in this command; an exchange costs >= 26 us "
                      "(1-rank RCCL probe, profiles/r3_shard_probe.json), so sum-check rounds shard only from 2^22-entry tables on")
    return {"scaling": "strong", "log2_cons": s, "value": N * steps / dt, "unit": "constraints/s", "ms_per_step": dt / steps * 1e3, "steps": steps,
            "ms_per_step_unsharded": dt1 * 1e3, "speedup_vs_unsharded": dt1 / (dt / steps), "amdahl": amdahl, "transport": transport, "setup_s": round(t_setup, 2),
            "all_gathers_per_proof": st["gathers"] / steps, "all_gather_bytes_per_proof": st["bytes"] / steps, "byte_identical_to_unsharded": True,
            "shards": "row commitments" + ("; ZK and batched-cubic sum-check rounds, bound, evaluate by index residue (tables >= 2^22 entries)" if s >= 22 else
                                           " only (every table is below the 2^22-entry residue threshold: the Fiat-Shamir-ordered steps are replicated)")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--log2-cons", type=int, default=20, help="log2 of num_cons = num_vars = num_nz_entries (BASELINE: 20)")
    ap.add_argument("--cpu-log2-cons", type=int, default=20, help="largest size the CPU baseline (oracle, all cores) is run at; at or below it the baseline runs at the metric's own size and its proof is compared with the GPU's")
    ap.add_argument("--cpu-threads", type=int, default=64, help="OpenMP threads of the all-cores CPU baseline (capped by the logical cores)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-side-metrics", action="store_true", help="skip the NIZK::prove / SNARK::encode side measurements")
    ap.add_argument("--concurrent", type=int, default=3, help="also measure K independent proofs in flight on the GPU (0 = skip); reported separately, never as `value`")
    ap.add_argument("--shard-commits", action="store_true", help="N>1: one proof, row commitments sharded over the ranks + all-gather (strong scaling)")
    ap.add_argument("--strong-timeout", type=int, default=600, help="seconds the strong-scaling leg (all its sizes) may take before the job prints what it has and exits")
    ap.add_argument("--strong-log2", default="20,22,24", help="N>1: sizes of the strong-scaling leg (BASELINE configs 4 and 5 are 2^20 and 2^22), one sharded proof each")
    ap.add_argument("--plumbing-only", action="store_true", help="TEST HOOK (tests/test_bench_dist.py): launch, rendezvous, rank count and aggregation only; no GPU, no proof, value null")
    ap.add_argument("--no-strong", action="store_true", help="N>1: skip the strong-scaling leg (one sharded proof) that follows the replica measurement")
    ap.add_argument("--phases", action="store_true", help="also print the per-phase span times (timer.rs names) to stderr")
    args = ap.parse_args()

    rank, world, dist = dist_setup(args.gpus)
    if args.plumbing_only:
        dist_barrier(dist)
        mx = dist_max(dist, 1.0 + rank)
        seen = int(round(dist_sum(dist, 1.0)))
        if seen != args.gpus:
            raise SystemExit("bench.py: %d ranks answered, --gpus %d" % (seen, args.gpus))
        if rank == 0:
            print(json.dumps({"metric": "plumbing only (no proof)", "value": None, "n_gpus": world, "n_ranks_seen": seen, "max_over_ranks": mx,
                              "self_launched": bool(os.environ.get("BENCH_SELF_LAUNCHED"))}))
        if dist is not None:
            dist.destroy_process_group()
        return
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    try:  # a launcher may expose one device per rank (HIP_VISIBLE_DEVICES): fold the rank onto what is visible
        import torch as _t
        if _t.cuda.is_available():
            local_rank %= max(1, _t.cuda.device_count())
    except ImportError:
        pass
    if os.environ.get("BENCH_FORCE_DEVICE") is not None:  # test hook: several ranks on one GPU (with BENCH_DIST_BACKEND=gloo)
        local_rank = int(os.environ["BENCH_FORCE_DEVICE"])
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the prover has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"

    from spartan_amd import prover as P, capi
    s = args.log2_cons
    N = 1 << s
    ctx = P.Ctx(local_rank)
    sharded = args.shard_commits and dist is not None
    seed = 0 if sharded else rank
    inst = P.Instance.produce_synthetic_r1cs(ctx, N, N, 10, seed=seed)  # profiler/snark.rs:23-31 shape
    gens = P.SNARKGens(ctx, N, N, 10, N)
    enc = P.SNARK.encode(ctx, inst, gens)
    args.tape_offset = 100  # tape seed = 100 + instance seed: the instance and tape of tests/golden (rank 0: the committed 2^20 digest)
    tape_seed = P.seed_scalar(b"tape", args.tape_offset + seed)

    # the headline step takes the assignment as a host buffer, as a Rust caller's `&[Scalar]` would arrive (the H2D
    # of `vars` is inside the metric); the VarsAssignment-resident variant is timed separately below (`resident_assignment`)
    assignment = P.VarsAssignment(ctx, inst.vars)

    def step(times=None, resident=False):
        return P.SNARK.prove(ctx, inst, enc, assignment if resident else inst.vars, inst.inputs, gens, b"snark_example", tape_seed, times)

    def read_prof():
        cap = 64
        names = (ctypes.c_char_p * cap)(); ms = (ctypes.c_double * cap)(); nl = (ctypes.c_uint64 * cap)(); by = (ctypes.c_double * cap)()
        ops = (ctypes.c_double * cap)()
        k = capi.lib.sp_prof_read(raw, names, ms, nl, by, ctypes.c_int(cap))
        capi.lib.sp_prof_read_ops(raw, ops, ctypes.c_int(cap))
        return {names[i].decode(): {"ms": ms[i], "launches": int(nl[i]), "alg_bytes": by[i], "alg_ops": ops[i]} for i in range(k) if nl[i]}

    def read_shapes(family):
        cap = 32
        sh = (ctypes.c_uint64 * cap)(); ms = (ctypes.c_double * cap)(); nl = (ctypes.c_uint64 * cap)(); by = (ctypes.c_double * cap)(); ops = (ctypes.c_double * cap)()
        k = capi.lib.sp_prof_read_shapes(raw, family.encode(), sh, ms, nl, by, ops, ctypes.c_int(cap))
        return [{"rows": (int(sh[i]) >> 32) & 0x7fffffff, "cols": int(sh[i]) & 0xffffffff, "background": bool(int(sh[i]) >> 63), "ms": ms[i],
                 "launches": int(nl[i]), "alg_bytes": by[i], "alg_ops": ops[i]} for i in range(min(k, cap))]

    def read_spans(family):
        cap = 4096
        return read_spans_raw(capi, raw, family, cap)

    def opt(key):
        v = ctypes.c_int64(0); capi.lib.sp_ctx_get_option(raw, key.encode(), ctypes.byref(v)); return v.value

    proof = None
    raw = ctx.raw()
    ceil = measured_ceilings() if rank == 0 else None
    gath = None
    if rank == 0 and not os.environ.get("BENCH_NO_GATHER_PROBE"):
        wb = [gens.window_bits(0), gens.window_bits(1)]
        # entries per (point, window) sub-table = 2^(c-1); the stride of a table entry follows from the table's size
        ent = [(len(gens.stream(k)) // 32) * (-(-254 // wb[k])) * (1 << (wb[k] - 1)) for k in (0, 1)]
        stride = 128  # one 128-byte line per entry of the gathered tables (table_bytes also counts the packed LDS-form tables when they are built)
        free0 = torch.cuda.mem_get_info(local_rank)[0]
        gath = gather_ceiling([gens.table_bytes(0) / 1e9, gens.table_bytes(1) / 1e9], [1 << (wb[0] - 1), 1 << (wb[1] - 1)], stride)
        # the probe is a process of its own that takes what is free (up to the larger table's size): the driver hands its pages back a
        # moment AFTER it exits — a 2^24 proof started straight behind it ran out of device memory with 150 GB nominally free
        t_wait = time.time()
        while torch.cuda.mem_get_info(local_rank)[0] < free0 - (1 << 30) and time.time() - t_wait < 60:
            time.sleep(0.2)
    if sharded:
        proof = step()  # unsharded bytes: every sharded proof below must equal them
        shard_transport = enable_sharding(P, ctx, dist, rank, world)
    for _ in range(args.warmup):
        p2 = step()
        if proof is not None and p2 != proof:
            raise SystemExit("sharded proof differs from the unsharded proof")
        proof = p2
    # Untimed profiling step: HIP events on every kernel family -> per-family breakdown and the dominant kernel.
    # (Each recorded launch costs two hipEventRecord calls; with ~1500 launches per proof that is ~10% of a step,
    # so the timed region below instruments the dominant family only.)
    capi.lib.sp_prof_reset(raw); capi.lib.sp_prof_select(raw, None); capi.lib.sp_prof_enable(raw, ctypes.c_int(1))
    proof = step()
    capi.lib.sp_prof_enable(raw, ctypes.c_int(0))
    breakdown = read_prof()
    fq_big = {}   # throughput-sized launches (>= 64 MB of algorithmic bytes) of the F_q families, from the same instrumented step
    for name in breakdown:
        if name != "msm_rows_fixed":
            for e_ in read_shapes(name):
                if e_["rows"] == 0x40000000 and e_["cols"] == 1:   # the pseudo-shape 0x4000000000000001 (include/spartan_hip.h)
                    fq_big[name] = e_
    dom = max(breakdown, key=lambda n: breakdown[n]["ms"])
    capi.lib.sp_prof_reset(raw)
    if not os.environ.get("BENCH_NO_PROF"):
        capi.lib.sp_prof_select(raw, dom.encode()); capi.lib.sp_prof_enable(raw, ctypes.c_int(1))
    if sharded:
        ctx.shard_stats(reset=True)
    dist_barrier(dist)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    phase = {}
    for _ in range(args.steps):
        p2 = step(phase)
        if proof is not None and p2 != proof:
            raise SystemExit("non-deterministic proof bytes")
        proof = p2
    torch.cuda.synchronize()
    dist_barrier(dist)
    dt = time.perf_counter() - t0
    dt = dist_max(dist, dt, dev if dist is not None and dist.get_backend() == "nccl" else "cpu")
    capi.lib.sp_prof_enable(raw, ctypes.c_int(0))
    # the same proof from an assignment already uploaded by VarsAssignment::new (no PCIe copy of `vars` inside the call): reported, never `value`
    k_host = min(args.steps, 10)
    shard_stats_timed = ctx.shard_stats() if sharded else None
    torch.cuda.synchronize()
    trips0 = capi.lib.sp_ctx_trips(raw)
    t0h = time.perf_counter()
    for _ in range(k_host):
        if step(resident=True) != proof:
            raise SystemExit("resident-assignment proof differs from the host-buffer proof")
    dt_host = (time.perf_counter() - t0h) / k_host
    fs_trips = (capi.lib.sp_ctx_trips(raw) - trips0) / k_host
    n_ranks_seen = int(round(dist_sum(dist, 1.0, dev if dist is not None and dist.get_backend() == "nccl" else "cpu")))
    if n_ranks_seen != args.gpus:
        raise SystemExit("bench.py: %d ranks answered the all-reduce, --gpus %d: no line is printed for a job of another size" % (n_ranks_seen, args.gpus))
    strong = None
    fam = read_prof() or {dom: breakdown[dom]}
    shapes = read_shapes(dom) if dom == "msm_rows_fixed" else []
    spans = read_spans(dom) if dom == "msm_rows_fixed" else []
    capi.lib.sp_prof_select(raw, None)

    # dominant kernel (HIP events recorded inside the timed region) -> roofline
    roofline = None
    if dom:
        f = fam[dom]
        avg_ms = f["ms"] / f["launches"]
        alg_per_launch = f["alg_bytes"] / f["launches"]
        if dom == "msm_rows_fixed" and s >= 6:
            # 32 B per committed scalar + 32 B per row out. A proof commits the witness (2^s scalars, 2^(s/2) rows) and `derefs`
            # (2^(s+3) scalars incl. the zero padding of the merged polynomial — committed by the reference too; since round 4 those rows are
            # recognised, not launched, so the library's own per-launch byte count no longer contains them) in launches_per_step launches
            per_proof = 32.0 * ((1 << s) + (1 << (s + 3))) + 32.0 * ((1 << (s // 2)) + (1 << ((s + 3) // 2)))
            alg_per_launch = per_proof / (f["launches"] / args.steps)
        ach = alg_per_launch / (avg_ms * 1e-3) / 1e9
        key = config_key(opt, capi.options_table, s)
        pmc, pmc_note = None, "no PMC entry for this configuration (kernel sources | size | non-default options = %s): collect with profiles/collect_r6.sh" % key
        try:
            pj = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))).get("entries", {}).get(key)
            if pj and dom in pj:
                pmc, pmc_note = pj[dom], "HBM bytes per launch from rocprofv3 --pmc (separate passes) for exactly this configuration (%s): profiles/%s" % (key, pj.get("source", "pmc_traffic.json"))
        except (OSError, ValueError):
            pass
        launched_per_launch = f["alg_bytes"] / f["launches"]   # the library's own count: 32 B per scalar of the rows it launched (the zero padding rows of `derefs` are not)
        roofline = {"bound": "hbm", "kernel": dom, "achieved": round(ach, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 6),
                    "traffic": pmc, "traffic_note": pmc_note,
                    "traffic_ratio": (round(pmc / launched_per_launch, 1) if isinstance(pmc, (int, float)) and launched_per_launch else None),
                    "traffic_ratio_note": "PMC HBM bytes per launch / bytes of the scalars the launch reads: everything above 1 is window-table gathers (the fixed-base method's price)",
                    "achieved_launched_bytes": round(launched_per_launch / (avg_ms * 1e-3) / 1e9, 3),
                    "achieved_launched_note": "the same rate counted on the scalars actually launched (6N of the 8N derefs entries: the padding rows are recognised, not read); `achieved` keeps the reference-defined 8N (ADVICE r4)", "avg_launch_ms": round(avg_ms, 5), "launches_per_step": f["launches"] / args.steps,
                    "alg_bytes_per_launch": alg_per_launch,
                    "alg_bytes_note": "SURVEY 8(d): 32 B per committed scalar (2^s witness + 2^(s+3) derefs, zero padding included as in the reference) + 32 B per row, over the family's launches of one proof",
                    "note": "255-bit EC / 253-bit field integer work: VALU-bound by construction, the HBM fraction is reported as the contract asks; `alu` below is the roofline that can approach 1 (DESIGN.md, roofline)"}
        # ---- ALU roofline: mixed additions/s of each MSM launch shape against the pt_madd chain measured on this GPU
        wb_sat, wb_eval = gens.window_bits(0), gens.window_bits(1)
        nwin_of = {False: gens.windows(0), True: gens.windows(1)}  # witness commit: gens_r1cs_sat stream; derefs: gens_r1cs_eval (windows = additions per scalar; mixed widths since round 6)
        R = 1 << ((s + 3) - (s + 3) // 2)   # columns of the derefs commitment (2^(s+3) entries, dense_mlpoly.rs:188-191)
        named = {}
        if s >= 6:
            # scalars that are actually non-zero: the derefs polynomial holds 6 * 2^s values, its top quarter is padding
            rows_row_half = 3 * N // R
            named[(1 << (s // 2), 1 << (s - s // 2), False)] = ("witness commit (poly_vars, + one blind per row)", N + (1 << (s // 2)), False)
            if (1 << (s // 2)) % 1024 == 0:  # host assignment: four row chunks, each launched behind its PCIe copy (sp_commit_rows_upload_start)
                named[((1 << (s // 2)) // 4, 1 << (s - s // 2), False)] = ("witness commit, one of four row chunks (each launched behind its PCIe copy)", (N + (1 << (s // 2))) // 4, False)
            named[(rows_row_half, R, True)] = ("derefs commit, row half (background stream, bg.eighths/8 of the CUs)", 3 * N, True)
            named[((8 * N) // R - rows_row_half, R, True)] = ("derefs commit, column half (+ zero padding rows; background stream behind the row half)", 3 * N, True)
            named[((8 * N) // R - rows_row_half, R, False)] = ("derefs commit, column half (+ zero padding rows)", 3 * N, True)
            named[((6 * N) // R - rows_row_half, R, False)] = ("derefs commit, column half (the zero padding rows of the merged polynomial are not launched)", 3 * N, True)
        alu_shapes = []
        for sh in shapes:
            nm = named.get((sh["rows"], sh["cols"], sh["background"]))
            if not nm or not sh["launches"]:
                continue
            lms = sh["ms"] / sh["launches"]
            nwin = nwin_of[nm[2]]
            madds = nm[1] * nwin
            e = {"shape": f'{sh["rows"]} x {sh["cols"]}', "what": nm[0], "launch_ms": round(lms, 4), "window_bits": wb_eval if nm[2] else wb_sat, "mixed_additions": madds,
                 "achieved_G_per_s": round(madds / lms / 1e6, 2)}
            if sh["background"]:
                # queue form, co-resident (the default since round 6): the background launch runs on EVERY CU next to the latency kernels;
                # otherwise the persistent background MSM holds bg.eighths/8 of the CUs (one workgroup each)
                co = opt("msm.form") in (0, 2) and sh["rows"] >= 256
                e["cu_share"] = 1.0 if co else opt("bg.eighths") / 8
                if co:
                    e["what"] = nm[0].replace("bg.eighths/8 of the CUs", "co-resident on every CU: msm.q_bg_waves wavefronts per CU, latency kernels prioritised")
            if ceil:
                e["frac"] = round(madds / lms / 1e6 / ceil["pt_madd_G_per_s"], 3)
                if sh["background"] and e["cu_share"] > 0:
                    e["frac_of_its_cus"] = round(e["frac"] / e["cu_share"], 3)
                # against the hardware's own multiply-add peak: a mixed addition is 7 F_p multiplications of 72 v_mad_u64_u32 each
                if ceil.get("v_mad_u64_u32_G_lane_ops_per_s"):
                    e["v_mad_frac"] = round(madds / lms / 1e6 * 7 * 72 / ceil["v_mad_u64_u32_G_lane_ops_per_s"], 3)
            if gath:   # one table gather per mixed addition: the pure-gather rate of this table set, same access pattern, no arithmetic
                gk = gath["b" if nm[2] else "a"]
                e["gather_ceiling_G_per_s"] = max(gk["inflight1"], gk["inflight2"])
                e["gather_frac"] = round(madds / lms / 1e6 / e["gather_ceiling_G_per_s"], 3)
            alu_shapes.append(e)
        roofline["alu"] = {"unit": "G mixed additions/s (7 F_p multiplications + 8 additions each)", "ceiling": ceil["pt_madd_G_per_s"] if ceil else None,
                           "ceiling_at_3_waves_per_simd": ceil.get("pt_madd_G_per_s_3_waves_per_simd") if ceil else None,
                           "ceiling_source": "bench/ubench_fpmul --json on this GPU, this run: dependent pt_madd chains at full occupancy, and at the row MSM's own occupancy (3 waves per SIMD)" if ceil else "bench/ubench_fpmul not built",
                           "v_mad_u64_u32_peak_G_lane_ops_per_s": ceil.get("v_mad_u64_u32_G_lane_ops_per_s") if ceil else None,
                           "gather": gath, "gather_source": "bench/gather_probe --json on this GPU, this run: the row MSM's gathers with no arithmetic behind them (one 96-byte entry per mixed addition)" if gath else "bench/gather_probe not built",
                           "note": "a launch needs BOTH resources at nearly the same rate (one gather per addition; the two ceilings are within 20 % of each other): `frac` (of the addition ceiling), `gather_frac` and `v_mad_frac` are reported side by side",
                           "additions_per_scalar": {"gens_r1cs_sat": nwin_of[False], "gens_r1cs_eval": nwin_of[True]}, "shapes": alu_shapes,
                           # `frac` IS the family figure: every mixed addition of the step over the CU-time it was given (the best
                           # launch shape is reported beside it as frac_best_shape, never as the headline)
                           "frac_sum_of_durations": (round(sum(e["mixed_additions"] for e in alu_shapes) / sum(e["launch_ms"] * e.get("cu_share", 1.0) for e in alu_shapes) / 1e6 / ceil["pt_madd_G_per_s"], 3)
                                                     if ceil and alu_shapes else None),
                           "frac_best_shape": max([e.get("frac", 0) for e in alu_shapes if "background" not in e["what"]] or [None])}
        # THE family figure: every mixed addition of the timed region over the time at least one row-MSM launch was in flight (the union of
        # the launches' intervals on one clock, sp_prof_read_spans): launches overlap — the column half of `derefs` is queued while the row half
        # still holds the CUs — so durations summed count the same milliseconds twice (kept beside it as frac_sum_of_durations)
        fam_adds = 0.0
        for r_, c_, bg_, _t0, _t1, _iss in spans:
            nm = named.get((r_, c_, bg_))
            if nm:
                fam_adds += nm[1] * nwin_of[nm[2]]
        busy = union_ms([(t0_, t1_) for _r, _c, _b, t0_, t1_, _i in spans])
        roofline["alu"]["busy_ms_per_step"] = round(busy / args.steps, 4) if busy else None
        roofline["alu"]["frac"] = round(fam_adds / busy / 1e6 / ceil["pt_madd_G_per_s"], 3) if ceil and busy else roofline["alu"]["frac_sum_of_durations"]
        roofline["alu"]["frac_note"] = "mixed additions of every row-MSM launch of the timed region / union of the launches' intervals / the pt_madd ceiling; the latency chain of the proof runs on the same CUs meanwhile"
        roofline["alu"]["frac_family"] = roofline["alu"]["frac"]
        # F_q streaming kernels (the HBM-shaped part): multiplications/s against the fq_mul chain ceiling, bytes/s against HBM
        fq = {}
        for name in ("sumcheck_eval", "sumcheck_bind_eval", "vecmat", "dot", "spark", "sparse", "eq_expand"):
            if name in breakdown and breakdown[name]["ms"] > 0:
                b_ = breakdown[name]
                def part(ms, n, by, ops):
                    if not n or ms <= 0: return None
                    e_ = {"ms_per_step": round(ms, 4), "launches": int(n), "GB_per_s": round(by / ms / 1e6, 1), "frac_of_hbm_peak": round(by / ms / 1e6 / HBM_PEAK_GBS, 4)}
                    if ops:
                        e_["G_fq_mul_per_s"] = round(ops / ms / 1e6, 2)
                        if ceil: e_["frac_of_fq_mul_ceiling"] = round(e_["G_fq_mul_per_s"] / ceil["fq_mul_G_per_s"], 3)
                    return e_
                big = fq_big.get(name)
                bm, bn, bb, bo = (big["ms"], big["launches"], big["alg_bytes"], big.get("alg_ops", 0.0)) if big else (0.0, 0, 0.0, 0.0)
                fq[name] = {"throughput_sized": part(bm, bn, bb, bo), "launch_sized": part(b_["ms"] - bm, b_["launches"] - bn, b_["alg_bytes"] - bb, b_["alg_ops"] - bo)}
        roofline["fq_kernels"] = {"ceiling_G_fq_mul_per_s": ceil["fq_mul_G_per_s"] if ceil else None, "families": fq,
                                  "note": "each family split into its throughput-sized launches (>= 64 MB of algorithmic bytes: the first rounds of the large layers) and the launch-sized rest (most of a proof's ~400 rounds run on tables that halve every round: latency-bound, their GB/s says nothing); per kernel and launch size with PMC bytes: profiles/r6_fq_bandwidth_2p20.txt / _2p22.txt"}

    if rank == 0:
        gpu_ms_total = sum(v["ms"] for v in breakdown.values())
        value = (1 if sharded else world) * N * args.steps / dt
        out = {
            "metric": "R1CS constraints/sec in SNARK::prove (synthetic 2^%d); bit-exact proof" % s,
            "value": value,
            "unit": "constraints/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if sharded else "weak",
            "vs_baseline": None,
            "dtype": "u64x4 (F_q Montgomery / F_p 2^255-19 limbs)",
            "data": "synthetic",
            "config": {"workload": f"SNARK::prove, Instance::produce_synthetic_r1cs(2^{s}, 2^{s}, 10), nnz 2^{s} per matrix; MSM + sum-checks + IPA + SPARK on GPU",
                       "proof_bytes": len(proof), "proof_sha256": __import__("hashlib").sha256(proof).hexdigest(),
                       "parallelism": ("1 proof, row commitments sharded over %d GPUs + all-gather" % world) if sharded else ("1 proof per GPU, %d independent proofs (replicas: no data-path collective)" % world),
                       "inputs": "instance, generators and computation commitment resident in HBM; the satisfying assignment `vars` is a host buffer uploaded inside the timed SNARK::prove (SURVEY 8d); inputs io on the host",
                       "resident_assignment": {"ms_per_step": round(dt_host * 1e3, 3), "value": N * world / dt_host if not sharded else N / dt_host, "steps": k_host,
                                               "note": "same proof from a VarsAssignment already in HBM (uploaded once by its constructor, outside prove): reported beside the headline, never as `value`"},
                       "bit_exact_against": "the in-repo oracle (oracle/: CPU restatement of the reference prover, pinned to RFC 9496 / Merlin / reference F_q vectors; no libspartan run exists here)",
                       "fs_trips_per_proof": fs_trips,
                       "table_GB": {"gens_r1cs_sat": round(gens.table_bytes(0) / 1e9, 2), "gens_r1cs_eval": round(gens.table_bytes(1) / 1e9, 2),
                                    "window_bits": [gens.window_bits(0), gens.window_bits(1)], "windows": [gens.windows(0), gens.windows(1)]},
                       "host_cores_busy": world, "host_note": "each proving thread spins on its completion flag and runs Merlin, the 2..5-term Sigma-protocol commitments, ~150 point encodes (in pairs), the challenge inversions and the end of every inner-product argument: one host core per GPU, flat out; a helper thread computes the tape-only halves of the ZK sum-checks' commitments ahead of the rounds (~0.5 ms of a second core per proof), another issues the witness upload",
                       "host_keccak": P.keccak_variant()},
            "roofline": roofline,
            "kernel_ms_per_step": {n: round(v["ms"], 4) for n, v in sorted(breakdown.items(), key=lambda kv: -kv[1]["ms"])},
            "gpu_busy_ms_per_step": round(gpu_ms_total, 3),
            "kernel_ms_note": "per-family totals from one untimed fully-instrumented step; roofline from the timed steps. Families OVERLAP (the background row-MSM runs under the sum-checks and openings, sparse evaluations on a low-priority stream): the sum exceeds ms_per_step, and event times of kernels that share the chip include what they wait",
            "phases_ms": {k_: round(v * 1e3, 3) for k_, v in phase.items()},
            "us_per_scalar": {"polycommit": round(phase.get("polycommit", 0) / N * 1e6, 5), "commit_nondet_witness": round(phase.get("commit_nondet_witness", 0) / (8 * N) * 1e6, 5),
                              "readme_i7_1065G7_one_core": README_US_PER_SCALAR, "note": "wall time of the phase / committed scalars (2^s and 8 * 2^s, zero padding included as in the README)"},
        }
        if s in (16, 20, 22) and not sharded and world == 1:   # the committed oracle digest of this very proof (tests/golden/make_golden.py --big)
            try:
                g_ = json.load(open(os.path.join(ROOT, "tests", "golden", "proof_digests.json")))["big"]["snark"].get(f"s{s}_seed0")
                if g_ and args.tape_offset == 100:
                    out["config"]["matches_oracle_digest"] = g_["sha256"] == out["config"]["proof_sha256"]
            except (OSError, KeyError, ValueError):
                pass
        if sharded:
            st_ = shard_stats_timed
            out["config"]["all_gathers_per_proof"] = st_["gathers"] / args.steps
            out["config"]["all_gather_bytes_per_proof"] = st_["bytes"] / args.steps
            out["config"]["shard_transport"] = shard_transport
        out["n_ranks_seen"] = n_ranks_seen
        if args.concurrent > 1 and world == 1:
            out["throughput_concurrent"] = concurrent_throughput(P, local_rank, s, args.concurrent, max(2, args.steps))
        if world == 1 and not args.no_side_metrics:
            out.update(side_metrics(P, ctx, inst, gens, N, s, tape_seed, max(2, args.steps)))
        if not args.no_cpu_baseline and world == 1:
            # (1) the oracle at the METRIC'S OWN SIZE on this box, all cores (bounded: ~20-40 s on the GPU box's host): its proof is compared with
            # the GPU proof timed above — a live check, next to the committed digest
            nthr = min(os.cpu_count() or 1, args.cpu_threads)
            cb = cpu_baseline(s if s <= args.cpu_log2_cons else args.cpu_log2_cons, threads=nthr, want_digest=(s <= args.cpu_log2_cons))
            if "proof_sha256" in cb:
                out["config"]["matches_oracle_live"] = cb["proof_sha256"] == out["config"]["proof_sha256"]
                cb["sample"] += "; the oracle's proof of the same instance and tape, computed in this run, " + ("EQUALS" if out["config"]["matches_oracle_live"] else "DIFFERS FROM") + " the GPU proof"
            out["cpu_baseline"] = cb
            # (2) one core, bounded sample (2^17: ~13 s): the single-thread figure next to the README's
            one = cpu_baseline(min(s, 17), threads=1)
            out["cpu_baseline_one_core"] = one
            # when the restatement is slower than the published single-core figure, the published figure is the denominator
            denom = max(one["value"], PUBLISHED_CPU_CPS)
            out["speedup_vs_cpu_baseline"] = value / denom
            out["speedup_note"] = ("denominator = published 26 797 constraints/s (README.md:375, one i7-1065G7 core, dalek SIMD): the oracle on one core of this box is slower (%.0f c/s)" % one["value"]
                                   if denom == PUBLISHED_CPU_CPS else "denominator = the oracle on one core of this box")
            out["speedup_vs_oracle_one_core_this_box"] = value / one["value"]
            out["speedup_vs_oracle_all_cores_this_box_same_size"] = value / cb["value"]
        if args.phases:
            print("phases (s):", json.dumps({k_: round(v, 5) for k_, v in phase.items()}), file=sys.stderr)
    else:
        out = None
    assignment.free(); enc.free(); gens.free(); inst.free()   # the strong leg builds its own sets (up to 2^24: 177 GB of window tables) in the room this leaves
    if dist is not None and not sharded and not args.no_strong:
        # The same command also reports ONE proof's latency over all ranks (strong scaling), at every size of --strong-log2 (default
        # 2^20, 2^22, 2^24: BASELINE configs 4 and 5 and the size where most of a proof is throughput-bound). It runs after everything
        # above is measured; a watchdog guarantees the job ends (and rank 0 still prints the replica line with the sizes finished so
        # far) if a collective of this leg hangs.
        import threading
        sizes = [int(x) for x in args.strong_log2.split(",") if x.strip()]
        legs = []
        if rank == 0:
            out["strong"] = legs
            out["strong_note"] = ("one proof over %d ranks, per size; the replica line above (`value`, scaling weak) is N independent proofs. north_star's "
                                  ">= 6x at 8 GPUs is met by the replicas only: a single proof is a Fiat-Shamir-ordered chain and its Amdahl ceiling is printed per size" % world)

        def bail():
            if rank == 0:
                legs.append({"scaling": "strong", "error": "timed out after %d s" % args.strong_timeout})
                print(json.dumps(out), flush=True)
            os._exit(0)
        wd = threading.Timer(args.strong_timeout, bail)
        wd.daemon = True
        wd.start()
        for ss in sizes:
            try:
                leg = strong_scaling_leg(P, ctx, dist, rank, world, ss, max(2, min(args.steps, 10 if ss <= 22 else 5)), dev)
            except Exception as e:  # noqa: BLE001  (reported, not fatal: the headline is the replica throughput above)
                leg = {"scaling": "strong", "log2_cons": ss, "error": repr(e)[:300]}
                # ranks must stay in lock-step: an error on one rank ends the leg on all of them
                bad = dist_sum(dist, 1.0, dev if dist.get_backend() == "nccl" else "cpu")
                legs.append(leg)
                break
            bad = dist_sum(dist, 0.0, dev if dist.get_backend() == "nccl" else "cpu")
            legs.append(leg)
            if bad:
                break
        wd.cancel()
    if rank == 0:
        print(json.dumps(out))
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
