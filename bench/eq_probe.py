"""Scratch probe: per-call time of sp_eq_expand + download of 1 element for several ell (kernel + one round trip)."""
import ctypes, sys, time, os, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spartan_amd import capi
from tests.helpers import mont_array, Q
ctx = capi.Ctx(0)
rng = random.Random(1)
for ell in (2, 5, 8, 12, 16, 20):
    r = mont_array([rng.randrange(Q) for _ in range(ell)])
    ts = []
    for it in range(30):
        t0 = time.perf_counter()
        t = capi.Table.eq(ctx, r, ell)
        t.download(1)
        ts.append(time.perf_counter() - t0)
        t.free()
    ts.sort()
    print("ell", ell, "median us", round(ts[len(ts) // 2] * 1e6, 1))
