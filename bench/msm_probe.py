"""Row-MSM alone: the derefs column half (1280 x 4096 scalars over the 4098-point gens_r1cs_eval stream) at the window width
option msm.wbits (SPARTAN_OPTIONS=msm.wbits=14) forces, three launches — the launch shape whose ALU fraction drops with the size of the table set. Run under
rocprofv3 --pmc to collect address-translation / fabric counters per width (profiles/collect_r3_msm_counters.sh)."""
import ctypes, hashlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spartan_amd import capi
rows, cols = 1280, 4096
B = bytes.fromhex("e2f2ae0a6abc4e71a884a961c500515f58e30b6aa582dd8db6a65945e08d2d76")
ctx = capi.Ctx(0)
g = capi.Gens(ctx, uniform=hashlib.shake_256(b"gens_r1cs_eval" + B).digest(64 * (cols + 2)))
rng = np.random.default_rng(1)
Z = rng.integers(0, 2**64, size=(rows * cols, 4), dtype=np.uint64); Z[:, 3] &= np.uint64((1 << 60) - 1)
t = capi.Table.upload(ctx, Z.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), rows * cols)
for it in range(3):
    t0 = time.time()
    g.commit_rows(t, rows, cols, None, 0, cols)
    print("bits %d: commit_rows %dx%d %.3f ms" % (g.window_bits(), rows, cols, (time.time() - t0) * 1e3), flush=True)
