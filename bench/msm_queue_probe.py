"""The queue form of the row MSM (msm_queue.hip, the default) against the strip / balanced forms (msm.form = 3) on the launch shapes of a 2^s
proof, same process, same generator set: ms per launch (best of N, host clock, reduction and encode included) and G mixed additions/s, for a list
of queue configurations "waves/depth/units". The commitments of every form must be equal (each is also checked against the oracle by
tests/msm_forms_worker.py). usage: python bench/msm_queue_probe.py [log2_cons] [configs, e.g. 12/2/64,8/3/64] [shapes: w,h,d]"""
import ctypes, hashlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spartan_amd import capi
s = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cfgs = [tuple(int(x) for x in c.split("/")) for c in (sys.argv[2] if len(sys.argv) > 2 else "12/2/64,8/3/64,8/2/64,12/2/32,12/2/128").split(",")]
which = sys.argv[3] if len(sys.argv) > 3 else "w,h,d"
B = bytes.fromhex("e2f2ae0a6abc4e71a884a961c500515f58e30b6aa582dd8db6a65945e08d2d76")
ctx = capi.Ctx(0)
ctx.set_option("testing.unlock", 1)
rng = np.random.default_rng(1)
wit_rows, wit_cols = 1 << (s // 2), 1 << (s - s // 2)
der_cols = 1 << ((s + 3) - (s + 3) // 2)
der_rows = (6 << s) // der_cols
shapes = [("w", "witness", b"gens_r1cs_sat", wit_rows, wit_cols, True), ("c", "witness chunk", b"gens_r1cs_sat", wit_rows // 4, wit_cols, True),
          ("h", "derefs half", b"gens_r1cs_eval", der_rows // 2, der_cols, False), ("d", "derefs whole", b"gens_r1cs_eval", der_rows, der_cols, False)]
REPS = 5 if s <= 22 else 3
for key, name, label, rows, cols, blind in shapes:
    if key not in which.split(","):
        continue
    g = capi.Gens(ctx, uniform=hashlib.shake_256(label + B).digest(64 * (cols + 2)))
    Z = rng.integers(0, 2**64, size=(rows * cols, 4), dtype=np.uint64); Z[:, 3] &= np.uint64((1 << 60) - 1)
    t = capi.Table.upload(ctx, Z.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), rows * cols)
    del Z
    bl = None
    if blind:
        bz = rng.integers(0, 2**64, size=(rows, 4), dtype=np.uint64); bz[:, 3] &= np.uint64((1 << 60) - 1)
        bl = bz.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))
    nwin = g.windows()
    madds = rows * (cols + (1 if blind else 0)) * nwin
    ref = None
    def run(tag):
        global ref
        best = 1e9
        for it in range(REPS):
            t0 = time.time()
            out = g.commit_rows(t, rows, cols, bl, 0, cols)
            best = min(best, time.time() - t0)
        if ref is None:
            ref = out
        assert out == ref or os.environ.get("PROBE_NOCHECK"), "forms disagree on %s (%s)" % (name, tag)   # PROBE_NOCHECK: the -DSP_Q_DIAG timing variants
        print("2^%d %-13s %5d x %5d  %-14s %2d adds/scalar  %8.3f ms  %6.2f G madd/s  (%d-bit windows)" % (s, name, rows, cols, tag, nwin, best * 1e3, madds / best / 1e9, g.window_bits()), flush=True)
    ctx.set_option("msm.form", 3); run("strip/balanced")
    ctx.set_option("msm.form", 0)
    for wv, dp, un in cfgs:
        ctx.set_option("msm.q_waves", wv); ctx.set_option("msm.q_units", un)  # (the ring depth is fixed at 2 since the depth-3 variant was retired)
        run("queue %d/%d/%d" % (wv, dp, un))
    t.free(); g.free()
print("MSM_QUEUE_PROBE_OK")
