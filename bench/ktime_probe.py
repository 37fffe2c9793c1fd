"""Where does the time of a launch-sized kernel go? Runs the two latency-bound kernels that dominate the proof's GPU time —
k_cubic_bind2_eval (the two-rounds-per-trip batched sum-check kernel, spark.hip) and k_ipa_round (one inner-product round,
core.hip) — from the DIAGNOSTIC build of the library (make -C spartan_amd/csrc ktime: -DSP_KTIME adds wall-clock stamps of
the first workgroup at the phase boundaries) and prints, per phase, the time between stamps next to the host-side time of
the whole call. 100 MHz device wall clock: 10 ns resolution.
Run on the GPU box from the repo root:  python bench/ktime_probe.py   (sets the library option debug.ktime through SPARTAN_OPTIONS)"""
import ctypes, os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("SPARTAN_HIP_LIB", os.path.join(ROOT, "spartan_amd", "lib", "libspartan_hip_ktime.so"))
os.environ.setdefault("SPARTAN_OPTIONS", "testing.unlock=1,debug.ktime=1")
from spartan_amd import capi
from tests.helpers import mont_bulk, fast_scalars, sz, vp, Q, load_oracle, gens_bytes

lib = capi.lib
ctx = capi.Ctx(0)
rng = random.Random(1)


def stamps(n=16):
    buf = (ctypes.c_longlong * n)()
    assert lib.sp_debug_ktime(ctx.h, buf, ctypes.c_int(n)) == 0
    return list(buf)


def show(name, rows, labels):
    """rows: list of stamp lists; prints the mean microseconds between consecutive labelled stamps"""
    print(name)
    for (i, j, what) in labels:
        d = [(r[j] - r[i]) / 100.0 for r in rows if r[i] and r[j]]
        if d:
            d.sort()
            print("   %-58s %7.2f us (median of %d; min %.2f max %.2f)" % (what, d[len(d) // 2], len(d), d[0], d[-1]))


# ---------------------------------------------------------------- k_cubic_bind2_eval
fq1 = lambda x: mont_bulk([x])
for ni, n0 in ((16, 512), (16, 64), (4, 512)):
    ev = (ctypes.c_uint64 * 12)(); co = (ctypes.c_uint64 * 48)(); heads = (ctypes.c_uint64 * (8 * ni + 4 * ni))()
    w = mont_bulk(fast_scalars(rng, ni))
    rows0, rows2, host0, host2 = [], [], [], []
    for rep in range(30):
        tabs = [[capi.Table.upload(ctx, mont_bulk(fast_scalars(rng, n0)), n0) for _ in range(ni)] for _ in range(3)]
        hA, hB, hC = [(vp * ni)(*[t.h for t in T]) for T in tabs]
        t0 = time.perf_counter()
        assert lib.sp_sumcheck_eval_coeffs_batched(ctx.h, hA, hB, hC, sz(ni), w, ev, co) == 0
        host0.append((time.perf_counter() - t0) * 1e6)
        rows0.append(stamps())
        t0 = time.perf_counter()
        assert lib.sp_sumcheck_bind2_eval_batched(ctx.h, hA, hB, hC, sz(ni), fq1(rng.randrange(Q)), fq1(rng.randrange(Q)), w, ev, co, heads) == 0
        host2.append((time.perf_counter() - t0) * 1e6)
        rows2.append(stamps())
        for T in tabs:
            for t in T:
                t.free()
    host0.sort(); host2.sort()
    print("==== k_cubic_bind2_eval, %d instances, tables of %d entries" % (ni, n0))
    print("   host time of the call: no bind %.1f us, two binds %.1f us (medians)" % (host0[len(host0) // 2], host2[len(host2) // 2]))
    lab0 = [(0, 3, "start -> table entries loaded (Triple2 from the host page, then HBM)"), (3, 4, "18 triple products (lines, 2 multiplications)"),
            (4, 5, "weight (from the host page) x product"), (5, 6, "block reduction + stores to the host page"), (6, 7, "signal_done (fence, counter, flag)"), (0, 7, "whole workgroup")]
    show("  no bind (sp_sumcheck_eval_coeffs_batched):", rows0, lab0)
    lab2 = [(0, 1, "start -> first loads back (Triple2 from the host page, then HBM)"), (1, 2, "bind at r0 (1 multiplication)"), (2, 3, "sync + bind at r1 + stores"),
            (3, 4, "18 triple products"), (4, 5, "weight x product"), (5, 6, "block reduction + stores to the host page"), (6, 7, "signal_done"), (0, 7, "whole workgroup")]
    show("  two binds (sp_sumcheck_bind2_eval_batched):", rows2, lab2)

# ---------------------------------------------------------------- k_ipa_round
orc = load_oracle()
import numpy as np
g_big = capi.Gens(ctx, compressed=gens_bytes(orc, 4096 + 1, b"gens_r1cs_eval"))  # the 4096-generator set: also the background MSM's
rows_bg = 768
Zbg = np.random.default_rng(3).integers(0, 2**64, size=(rows_bg * 4096, 4), dtype=np.uint64); Zbg[:, 3] &= np.uint64((1 << 60) - 1)
tbg = capi.Table.upload(ctx, Zbg.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), rows_bg * 4096)
lab = [(0, 1, "first workgroup: start -> scalar a'[i] s'[p] (folds + product)"), (1, 2, "digit + table entry gathered (HBM)"), (2, 3, "entry -> extended point (1 F_p mult), radix-2^25.5"),
       (3, 4, "LDS tree over 256 points (8 levels)"), (4, 5, "partial sum out, fence, ticket"), (0, 5, "first workgroup, whole"),
       (8, 9, "reducing workgroup: fence + load of the row's partials"), (9, 10, "LDS tree over the partials"), (10, 11, "convert + store to the host page"),
       (0, 11, "first stamp -> row sum stored (critical path of the launch)")]
for n, under_bg in ((4096, False), (1024, False), (1024, True)):
    g = g_big if n == 4096 else capi.Gens(ctx, compressed=gens_bytes(orc, n + 1, b"gens_r1cs_sat"))  # n generators, Q base, h
    a = mont_bulk(fast_scalars(rng, n)); b = mont_bulk(fast_scalars(rng, n))
    ipa = vp()
    assert lib.sp_ipa_begin(ctx.h, g.h, sz(0), sz(n), sz(n), sz(n + 1), fq1(7), a, b, ctypes.byref(ipa)) == 0
    job = vp()
    L = (ctypes.c_uint8 * 32)(); R = (ctypes.c_uint8 * 32)()
    rows, host = [], []
    k = n
    while k >= 2:
        if k == n // 2 and under_bg:  # after the first round (it builds the host-side tables of Q and h: milliseconds, once per generator set):
            # the derefs row half as the prover starts it, 768 x 4096 on the background stream (5/8 of the CUs), ~5.5 ms
            assert lib.sp_commit_rows_dev_begin(ctx.h, g_big.h, sz(0), tbg.h, sz(0), sz(rows_bg), sz(4096), ctypes.byref(job)) == 0
            time.sleep(0.0003)
        t0 = time.perf_counter()
        assert lib.sp_ipa_round_lr(ipa, fq1(rng.randrange(Q)), fq1(rng.randrange(Q)), L, R) == 0
        host.append((time.perf_counter() - t0) * 1e6)
        rows.append(stamps())
        u = rng.randrange(Q)
        assert lib.sp_ipa_round_fold(ipa, fq1(u), fq1(pow(u, Q - 2, Q))) == 0
        k //= 2
    if under_bg:
        sink = (ctypes.c_uint8 * (32 * rows_bg))()
        assert lib.sp_job_wait(job, sink) == 0
    lib.sp_ipa_free(ipa)
    print("==== k_ipa_round, n = %d generators (%d-bit windows)%s, %d rounds; host time of sp_ipa_round_lr per round: %s us" %
          (n, g.window_bits(), " WHILE the background MSM (768 x 4096) runs" if under_bg else "", len(rows), " ".join("%.0f" % h for h in host)))
    show("  per round (medians over the rounds):", rows, lab)
    if g is not g_big: g.free()
tbg.free(); g_big.free()
ctx.close()
