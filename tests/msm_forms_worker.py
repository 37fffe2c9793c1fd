"""Worker of tests/test_gpu_kernels.py::test_row_msm_forms_match_oracle: the row MSM's launch form is chosen once per process
(msm.form = 0, the default: the queue form of msm_queue.hip for launches of >= 256 rows; msm.form = 3: the strip form and the balanced form with two entries in flight, chosen per launch; msm.form = 1 with msm.lds_bits = 10: the LDS-staged small-window form — all through SPARTAN_OPTIONS), so every form runs in a process of its own. Shapes that only these plans select, each against the oracle's orc_commit_rows:
blinds (an extra column that starts or ends a run in the middle of a scalar), rows of zeros, short scalars (the early exit of the strip form
and the ballot skip of the balanced form: SNARK::encode's addresses and timestamps, src/sparse_mlpoly.rs:483-503), scalars with only high
bits set (carries into the top window), a run boundary inside the signed recoding's carry chain, and the background kernel."""
import ctypes, os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from spartan_amd import capi
from tests import helpers as H

orc = H.load_oracle()
ctx = capi.Ctx(0)
rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
Q = H.Q
checked = 0


def scalars(kind, n):
    if kind == "uniform":
        return H.fast_scalars(rng, n)
    if kind == "short":      # addresses / timestamps: a few bits
        return [rng.randrange(1 << rng.choice((1, 7, 20, 33))) for _ in range(n)]
    if kind == "zero_rows":  # handled by the caller
        return H.fast_scalars(rng, n)
    if kind == "high":       # only the top windows are non-zero
        return [(rng.randrange(1, 1 << 13) << 239) % Q for _ in range(n)]
    if kind == "carry":      # long runs of ones: the signed recoding carries through many windows
        return [((1 << rng.randrange(200, 252)) - 1 - rng.randrange(4)) % Q for _ in range(n)]
    if kind == "mixed":
        return [rng.choice((0, 1, Q - 1, rng.randrange(Q), rng.randrange(1 << 16), (1 << 252) + rng.randrange(1 << 30))) % Q for _ in range(n)]
    raise ValueError(kind)


def check(rows, cols, kind, blinds, background=False):
    global checked
    g = capi.Gens(ctx, compressed=H.gens_bytes(orc, cols))
    Z = scalars(kind, rows * cols)
    if kind == "zero_rows":
        for r in range(rows):
            if (r // 64) % 2:   # whole wavefronts of zero rows next to live ones
                Z[r * cols:(r + 1) * cols] = [0] * cols
    Zm = H.mont_bulk(Z)
    bl = H.mont_bulk(H.fast_scalars(rng, rows)) if blinds else None
    t = capi.Table.upload(ctx, Zm, rows * cols)
    want = (ctypes.c_uint8 * (32 * rows))()
    comp = g.compressed
    assert orc.orc_commit_rows(comp[:32 * cols], H.sz(cols), comp[32 * cols:], Zm, H.sz(rows), H.sz(cols), bl, want) == 0
    if background:
        got = g.commit_rows_wait(g.commit_rows_begin(t, rows, cols, g_off=0))
    else:
        got = g.commit_rows(t, rows, cols, bl, g_off=0, h_idx=cols)
    assert got == bytes(want), "row MSM mismatch: %dx%d %s blinds=%s bg=%s FLAT=%s" % (rows, cols, kind, blinds, background, os.environ.get("SPARTAN_OPTIONS"))
    t.free(); g.free()
    checked += 1


for kind in ("uniform", "short", "zero_rows", "high", "carry", "mixed"):
    check(256, 160, kind, True)          # one row-block: the balanced form (when selected) with the blind as the last column
check(512, 96, "uniform", False)         # two row-blocks, no blinds
check(1024, 48, "mixed", True)           # four row-blocks
check(2048, 40, "short", False)          # eight row-blocks: strip form in every mode (heterogeneous rows)
check(768, 128, "uniform", False, background=True)
check(768, 128, "zero_rows", False, background=True)
if ctx.get_option("msm.form") != 3:
    # the LDS-staged small-window form (msm_lds.hip; msm.form = 1) takes every launch of >= 512 rows, the queue form (msm_queue.hip: the
    # default, msm.form = 0) every launch of >= 256 rows: row-blocks that are not a multiple of a wavefront,
    # two and three row-blocks, runs that start inside a scalar, the blind as the last column, every scalar kind, the persistent background form
    for kind in ("uniform", "short", "zero_rows", "high", "carry", "mixed"):
        check(576, 40, kind, True)
    check(1000, 33, "uniform", True)
    check(1536, 24, "mixed", False)
    check(1100, 20, "carry", True)
    check(2112, 10, "uniform", True)
    check(768, 128, "mixed", False, background=True)
    check(1536, 300, "uniform", False, background=True)
if ctx.get_option("msm.form") in (0, 2):
    # the queue form: a last row group that is not a whole wavefront, the smallest launch it takes, items of four units (runs that start
    # and end inside every scalar: msm.q_units = 4 in one of the settings), whole wavefronts of zero rows in the background
    check(300, 200, "mixed", True)
    check(256, 900, "carry", False)
    check(420, 300, "short", True)
    check(832, 96, "zero_rows", False, background=True)
print("MSM_FORMS_OK %d" % checked)
ctx.close()
