"""A/B of the row MSM's forms on the launch shapes of a 2^s proof: the wide-window gathered forms (core.hip) against the LDS-staged
small-window form (msm_lds.hip), same process, same generator set (built with both table kinds: option msm.lds_bits), the form chosen per
launch by option msm.form. Prints ms per launch (best of 6) and G mixed additions/s; the two forms' commitments must be equal (each is
also checked against the oracle by tests/msm_forms_worker.py).
usage: python bench/msm_lds_probe.py [log2_cons] [lds_bits]"""
import ctypes, hashlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spartan_amd import capi
s = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B = bytes.fromhex("e2f2ae0a6abc4e71a884a961c500515f58e30b6aa582dd8db6a65945e08d2d76")
LB = int(sys.argv[2]) if len(sys.argv) > 2 else 10
ctx = capi.Ctx(0)
ctx.set_option("msm.lds_bits", LB)
rng = np.random.default_rng(1)
wit_rows, wit_cols = 1 << (s // 2), 1 << (s - s // 2)
der_cols = 1 << ((s + 3) - (s + 3) // 2)
der_rows = (6 << s) // der_cols
shapes = [("witness", b"gens_r1cs_sat", wit_rows, wit_cols, True), ("derefs half", b"gens_r1cs_eval", der_rows // 2, der_cols, False),
          ("derefs whole", b"gens_r1cs_eval", der_rows, der_cols, False)]
for name, label, rows, cols, blind in shapes:
    t0 = time.time()
    g = capi.Gens(ctx, uniform=hashlib.shake_256(label + B).digest(64 * (cols + 2)))
    t_build = time.time() - t0
    Z = rng.integers(0, 2**64, size=(rows * cols, 4), dtype=np.uint64); Z[:, 3] &= np.uint64((1 << 60) - 1)
    t = capi.Table.upload(ctx, Z.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), rows * cols)
    bl = None
    if blind:
        bz = rng.integers(0, 2**64, size=(rows, 4), dtype=np.uint64); bz[:, 3] &= np.uint64((1 << 60) - 1)
        bl = bz.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))
    outs = {}
    for form in ("wide", "lds", "queue"):
        ctx.set_option("msm.form", {"wide": 3, "lds": 1, "queue": 2}[form])
        best = 1e9
        for it in range(6):
            t0 = time.time()
            out = g.commit_rows(t, rows, cols, bl, 0, cols)
            best = min(best, time.time() - t0)
        outs[form] = out
        nwin = -(-254 // LB) if form == "lds" else g.windows()
        madds = rows * (cols + (1 if blind else 0)) * nwin
        print("2^%d %-13s %5d x %5d  %-4s  %2d adds/scalar  %.3f ms  %.2f G madd/s  (set built in %.2f s, wide %d bits)" %
              (s, name, rows, cols, form, nwin, best * 1e3, madds / best / 1e9, t_build, g.window_bits()), flush=True)
    assert outs["wide"] == outs["lds"] == outs["queue"], "forms disagree on " + name
    t.free(); g.free()
print("MSM_LDS_PROBE_OK")
