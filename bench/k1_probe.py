"""Scratch probe: kernel-level timings at BASELINE sizes (not the bench contract)."""
import ctypes, hashlib, sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spartan_amd import capi

def rand_fq(n, seed):
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 2**64, size=(n, 4), dtype=np.uint64)
    a[:, 3] &= np.uint64((1 << 60) - 1)
    return a

ctx = capi.Ctx(0)
ctx.prof_enable(True)
s = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rows = 1 << (s // 2); cols = 1 << (s - s // 2)
B = bytes.fromhex("e2f2ae0a6abc4e71a884a961c500515f58e30b6aa582dd8db6a65945e08d2d76")
t0 = time.time()
g = capi.Gens(ctx, uniform=hashlib.shake_256(b"gens_r1cs_sat" + B).digest(64 * (cols + 1)))
print("gens+tables", cols + 1, "points", time.time() - t0, "s")
Z = rand_fq(rows * cols, 1)
t = capi.Table.upload(ctx, Z.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), rows * cols)
bl = rand_fq(rows, 2)
for it in range(3):
    t0 = time.time()
    out = g.commit_rows(t, rows, cols, bl.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), 0, cols)
    print("commit_rows %dx%d" % (rows, cols), (time.time() - t0) * 1e3, "ms")
# sumcheck: 4 tables of 2^s
tabs = [capi.Table.upload(ctx, rand_fq(1 << s, 10 + k).ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), 1 << s) for k in range(4)]
r = rand_fq(1, 99)
t0 = time.time()
e = capi.sumcheck_eval(ctx, 2, tabs)
for j in range(s - 1):
    e = capi.sumcheck_bind_eval(ctx, 2, tabs, r.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)))
print("phase-1-like sumcheck rounds", (time.time() - t0) * 1e3, "ms")
t0 = time.time()
q = capi.Table.eq(ctx, rand_fq(s, 5).ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), s)
print("eq_expand", (time.time() - t0) * 1e3, "ms")
for k, v in ctx.prof_read().items():
    if v["launches"]:
        print("%-22s n=%4d total %9.3f ms avg %9.4f ms  alg %.1f MB  -> %.1f GB/s" % (k, v["launches"], v["ms"], v["ms"] / v["launches"], v["alg_bytes"] / 1e6, v["alg_bytes"] / max(v["ms"], 1e-9) / 1e6))
