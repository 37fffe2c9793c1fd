"""What a background row commitment (k_msm_rows_bg on bg.eighths/8 of the CUs) costs the latency-bound foreground:
microseconds per call of (a) a few-term table-lookup commitment on the device (sp_msm_indexed over 1024 generators: the shape of
an inner-product round), (b) a launch-sized pure-ALU kernel (sp_sumcheck_eval_coeffs_batched on 64-entry tables), (c) a
32 MB streaming pass (sp_evaluate of a 2^20 table), each alone and while background commits are in flight."""
import ctypes, hashlib, os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spartan_amd import capi
from tests.helpers import mont_bulk, fast_scalars, sz, vp

BASE = bytes.fromhex("e2f2ae0a6abc4e71a884a961c500515f58e30b6aa582dd8db6a65945e08d2d76")
ctx = capi.Ctx(0)
npts = int(os.environ.get("PROBE_POINTS", "4097"))
gens = capi.Gens(ctx, uniform=hashlib.shake_256(b"bg_probe" + BASE).digest(64 * npts))
# the foreground lookups use their own generator set (as the witness opening does: gens_r1cs_sat, while the background commit
# runs over gens_r1cs_eval) unless PROBE_SAME_TABLE is set
gens_fg = gens if os.environ.get("PROBE_SAME_TABLE") else capi.Gens(ctx, uniform=hashlib.shake_256(b"bg_probe_fg" + BASE).digest(64 * 1025))
S_dev = None
rng = random.Random(3)
rows, cols = 768, npts - 1
Z = capi.Table.upload(ctx, mont_bulk(fast_scalars(rng, rows * cols)), rows * cols)
S_tab = capi.Table.upload(ctx, mont_bulk(fast_scalars(rng, 2 * 1024)), 2 * 1024)
ni, n = 12, 64
tabs = [[capi.Table.upload(ctx, mont_bulk(fast_scalars(rng, n)), n) for _ in range(ni)] for _ in range(3)]
hA, hB, hC = [(vp * ni)(*[t.h for t in T]) for T in tabs]
ev = (ctypes.c_uint64 * (12 * ni))(); co = (ctypes.c_uint64 * (48 * ni))()
big = capi.Table.upload(ctx, mont_bulk(fast_scalars(rng, 1 << 20)), 1 << 20)
r20 = mont_bulk(fast_scalars(rng, 20))

def fg(kind):
    if kind == "lookup": gens_fg.commit_rows(S_tab, 2, 1024, None, g_off=0, h_idx=1024)
    elif kind == "alu": capi.lib.sp_sumcheck_eval_coeffs_batched(ctx.h, hA, hB, hC, sz(ni), None, ev, co)
    else: capi.evaluate(ctx, big, r20, 20)

def measure(kind, with_bg):
    jobs = [gens.commit_rows_begin(Z, rows, cols) for _ in range(3)] if with_bg else []
    time.sleep(0.0005)
    ts = []
    t_end = time.perf_counter() + (0.012 if with_bg else 0.006)
    while time.perf_counter() < t_end:
        t0 = time.perf_counter(); fg(kind); ts.append((time.perf_counter() - t0) * 1e6)
    t0 = time.perf_counter()
    for j in jobs: gens.commit_rows_wait(j)
    ts.sort()
    return ts[len(ts) // 2], ts[int(len(ts) * 0.9)], len(ts), (time.perf_counter() - t0) * 1e3

for kind in ("lookup", "alu", "stream"):
    for _ in range(20): fg(kind)
    a = measure(kind, False); b = measure(kind, True)
    print(f"{kind:7s} alone: median {a[0]:7.1f} us p90 {a[1]:7.1f} (n={a[2]})   under background commits: median {b[0]:7.1f} us p90 {b[1]:7.1f} (n={b[2]}; bg tail {b[3]:.1f} ms)")
