"""Scratch probe: time a 1024x1024 commit on the background stream under SPARTAN_OPTIONS=bg.eighths=k (is the CU mask honoured?)."""
import ctypes, hashlib, sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spartan_amd import capi
rng = np.random.default_rng(1)
rows = cols = 1024
Z = rng.integers(0, 2**64, size=(rows * cols, 4), dtype=np.uint64); Z[:, 3] &= np.uint64((1 << 60) - 1)
ctx = capi.Ctx(0)
B = bytes.fromhex("e2f2ae0a6abc4e71a884a961c500515f58e30b6aa582dd8db6a65945e08d2d76")
g = capi.Gens(ctx, uniform=hashlib.shake_256(b"gens_r1cs_sat" + B).digest(64 * (cols + 1)))
t = capi.Table.upload(ctx, Z.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), rows * cols)
for it in range(3):
    t0 = time.perf_counter()
    j = g.commit_rows_begin(t, rows, cols)
    out = g.commit_rows_wait(j)
    print("OPTIONS", os.environ.get("SPARTAN_OPTIONS"), "bg commit ms", (time.perf_counter() - t0) * 1e3)
