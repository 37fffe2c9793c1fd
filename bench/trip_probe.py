"""Round-trip cost of one launch-sized kernel: N calls of sp_sumcheck_eval_coeffs_batched on 12 instances of 64-entry tables
(k_cubic_bind2_eval without a bind: the tables are left as they are), microseconds per call. Run twice to compare the
completion signal raised by the kernel itself. (The flag kernel queued behind the last kernel, the form of rounds 1-2, measured 0.6 us slower per trip in round 2 and was retired in round 6.)"""
import ctypes, os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spartan_amd import capi
from tests.helpers import mont_bulk, fast_scalars, sz, vp

ctx = capi.Ctx(0)
rng = random.Random(1)
ni, n = 12, int(os.environ.get("TRIP_LEN", "64"))
tabs = [[capi.Table.upload(ctx, mont_bulk(fast_scalars(rng, n)), n) for _ in range(ni)] for _ in range(3)]
hA, hB, hC = [(vp * ni)(*[t.h for t in T]) for T in tabs]
ev = (ctypes.c_uint64 * (12 * ni))(); co = (ctypes.c_uint64 * (48 * ni))()
f = capi.lib.sp_sumcheck_eval_coeffs_batched
for _ in range(200):
    assert f(ctx.h, hA, hB, hC, sz(ni), None, ev, co) == 0
N = 5000
best = 1e9
for rep in range(5):
    t0 = time.perf_counter()
    for _ in range(N):
        f(ctx.h, hA, hB, hC, sz(ni), None, ev, co)
    best = min(best, (time.perf_counter() - t0) / N * 1e6)
print("len %d: %.2f us per trip (best of 5 x %d)%s" % (n, best, N, "  [signal in kernel]"))
