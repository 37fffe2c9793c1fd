"""Scratch probe: per-call latency floor of the C ABI (launch + completion wait) and small-commit latency."""
import ctypes, hashlib, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spartan_amd import capi
import numpy as np
ctx = capi.Ctx(0)
def rand_fq(n, seed):
    rng = np.random.default_rng(seed); a = rng.integers(0, 2**64, size=(n, 4), dtype=np.uint64); a[:, 3] &= np.uint64((1 << 60) - 1); return a
P = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))
t4 = [capi.Table.upload(ctx, P(rand_fq(4, k)), 4) for k in range(4)]
for name, fn, n in [("heads(4 tables): 1 tiny launch + wait", lambda: capi.heads(ctx, t4), 2000)]:
    fn(); t0 = time.perf_counter()
    for _ in range(n): fn()
    print("%-50s %.1f us/call" % (name, (time.perf_counter() - t0) / n * 1e6))
B = bytes.fromhex("e2f2ae0a6abc4e71a884a961c500515f58e30b6aa582dd8db6a65945e08d2d76")
g = capi.Gens(ctx, uniform=hashlib.shake_256(b"gens_r1cs_sat" + B).digest(64 * 8))
S = rand_fq(14, 3)
for rows, cols in [(1, 2), (2, 7), (3, 7)]:
    idx = list(range(cols))
    f = lambda: g.msm_indexed(idx, P(S), rows)
    f(); t0 = time.perf_counter()
    for _ in range(300): f()
    print("msm_indexed rows=%d cols=%d %38s %.1f us/call" % (rows, cols, "", (time.perf_counter() - t0) / 300 * 1e6))
ctx.prof_enable(True)
for _ in range(100): g.msm_indexed(list(range(7)), P(S), 2)
for k, v in ctx.prof_read().items():
    if v["launches"]: print("%-22s avg %.1f us" % (k, 1e3 * v["ms"] / v["launches"]))
tabs = [capi.Table.upload(ctx, P(rand_fq(1 << 12, 10 + k)), 1 << 12) for k in range(4)]
r = rand_fq(1, 9)
t0 = time.perf_counter()
for _ in range(200): capi.sumcheck_eval(ctx, 2, tabs)
print("sumcheck_eval 4x2^12 %31s %.1f us/call" % ("", (time.perf_counter() - t0) / 200 * 1e6))
