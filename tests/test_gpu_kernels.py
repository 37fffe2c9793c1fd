"""Parity of every HIP kernel behind the C ABI against the oracle, on seeded inputs (bit-exact)."""
import ctypes, random
import pytest
from tests.helpers import *

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from spartan_amd import capi
    c = capi.Ctx(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def gens40(ctx, orc):
    from spartan_amd import capi
    g = capi.Gens(ctx, compressed=gens_bytes(orc, 39))  # 40 points: G[0..39), h = P[39]
    yield g
    g.free()


def test_gens_from_uniform_matches_oracle(ctx, orc):
    import hashlib
    from spartan_amd import capi
    from tests.test_oracle_pins import BASEPOINT
    label = b"gens_r1cs_sat"
    n = 9
    stream = hashlib.shake_256(label + bytes.fromhex(BASEPOINT)).digest(64 * n)
    g = capi.Gens(ctx, uniform=stream)
    assert g.compressed == gens_bytes(orc, n - 1, label)
    g.free()


def test_generator_tables_are_shared_between_contexts(orc):
    """window tables are built once per (device, generator bytes) and reference-counted across contexts (core.hip gens_build)"""
    import hashlib, time
    from spartan_amd import capi
    from tests.test_oracle_pins import BASEPOINT
    n = 129
    stream = hashlib.shake_256(b"gens_cache_test" + bytes.fromhex(BASEPOINT)).digest(64 * n)
    a, b = capi.Ctx(0), capi.Ctx(0)
    t0 = time.perf_counter(); ga = capi.Gens(a, uniform=stream); t_build = time.perf_counter() - t0
    t0 = time.perf_counter(); gb = capi.Gens(b, uniform=stream); t_hit = time.perf_counter() - t0
    assert ga.compressed == gb.compressed == gens_bytes(orc, n - 1, b"gens_cache_test")
    assert t_hit < 0.5 * t_build, (t_build, t_hit)
    rng = random.Random(77)
    Z = rand_scalars(rng, 4 * 128)
    want = (ctypes.c_uint8 * 128)()
    g = ga.compressed
    assert orc.orc_commit_rows(g[:32 * 128], sz(128), g[32 * 128:], mont_array(Z), sz(4), sz(128), None, want) == 0
    ga.free()                                   # the table must survive its first owner
    assert gb.commit_rows(mont_array(Z), 4, 128, None, g_off=0, h_idx=128) == bytes(want)
    gb.free(); a.close()
    gc = capi.Gens(b, uniform=stream)           # last handle gone: rebuilt from scratch, same bytes
    assert gc.commit_rows(mont_array(Z), 4, 128, None, g_off=0, h_idx=128) == bytes(want)
    gc.free(); b.close()


def test_gens_upload_rejects_bad_point(ctx, orc):
    from spartan_amd import capi
    from tests.test_oracle_pins import RFC_BAD
    good = gens_bytes(orc, 3)
    for enc in RFC_BAD:   # every class of RFC 9496 A.2 invalid encodings is refused by the device decoder (k_points_load)
        bad = good[:32] + bytes.fromhex(enc) + good[64:]
        with pytest.raises(capi.SpartanHipError, match="-4"):
            capi.Gens(ctx, compressed=bad)


@pytest.mark.parametrize("rows,cols,kind,blind", [(1, 1, "uniform", False), (1, 5, "edge", True), (4, 8, "uniform", True), (32, 32, "uniform", True),
                                                  (3, 39, "sparse", False), (64, 16, "small", True), (130, 39, "uniform", True)])
def test_commit_rows_matches_oracle(ctx, orc, gens40, rows, cols, kind, blind):
    rng = random.Random(rows * 1000 + cols)
    Z = rand_scalars(rng, rows * cols, kind)
    bl = rand_scalars(rng, rows, "uniform") if blind else None
    g = gens40.compressed
    got = gens40.commit_rows(mont_array(Z), rows, cols, mont_array(bl) if blind else None, g_off=0, h_idx=39)
    want = (ctypes.c_uint8 * (32 * rows))()
    rc = orc.orc_commit_rows(g[:32 * cols], sz(cols), g[32 * 39:32 * 40], mont_array(Z), sz(rows), sz(cols), mont_array(bl) if blind else None, want)
    assert rc == 0
    assert got == bytes(want)


@pytest.fixture(scope="module")
def gens301(ctx, orc):
    from spartan_amd import capi
    g = capi.Gens(ctx, compressed=gens_bytes(orc, 300))  # 301 points: G[0..300), h = P[300]
    yield g
    g.free()


# every launch plan of msm_launch (core.hip): lookups+tree in one launch / + second reduce (rows <= 8, host encode);
# windowed one-pass and two-pass trees with per-block and one-lane-per-row encodes; row-strip kernel with and without the
# XCD-aware tile order
@pytest.mark.parametrize("rows,cols,blind", [(1, 300, True), (8, 300, False), (16, 128, True), (100, 128, False), (256, 128, True),
                                             (70, 300, True), (512, 64, False)])
def test_commit_rows_launch_plans_match_oracle(ctx, orc, gens301, rows, cols, blind):
    rng = random.Random(rows * 7919 + cols)
    Z = rand_scalars(rng, rows * cols, "uniform")
    bl = rand_scalars(rng, rows, "uniform") if blind else None
    g = gens301.compressed
    got = gens301.commit_rows(mont_array(Z), rows, cols, mont_array(bl) if blind else None, g_off=0, h_idx=300)
    want = (ctypes.c_uint8 * (32 * rows))()
    rc = orc.orc_commit_rows(g[:32 * cols], sz(cols), g[32 * 300:32 * 301], mont_array(Z), sz(rows), sz(cols), mont_array(bl) if blind else None, want)
    assert rc == 0
    assert got == bytes(want)


@pytest.mark.parametrize("wbits", [5, 6, 8, 10, 12, 13, 14, 15, -17, -18, -19, -20, -21, -22, -24, -26, -32])
def test_commit_rows_at_every_window_width(ctx, orc, wbits, monkeypatch):
    """the window geometry is a property of the generator set, chosen at upload (core.hip choose_geom): every uniform width (option
    msm.wbits) and every number of MIXED-width windows (option msm.windows, the negative parameters: 17 = 1 x 14 + 16 x 15 bits, 18 = 16 x
    14 + 2 x 15, ... 32 = 2 x 7 + 30 x 8: msm.hpp) gives the same commitments through every launch plan (one-launch small, windowed trees,
    row strips, the queue form, the indexed lookups of the inner-product argument)"""
    from spartan_amd import capi
    if wbits > 0:
        ctx.set_option("msm.wbits", wbits)   # read when a generator set is built
    else:
        ctx.set_option("msm.wbits", 0); ctx.set_option("msm.windows", -wbits)
    label = b"gens_width_%d" % (wbits & 0xff)      # fresh points per geometry: a resident table set would be reused whatever its width
    g = capi.Gens(ctx, compressed=gens_bytes(orc, 130, label))
    if wbits > 0:
        assert g.window_bits() == wbits and g.windows() == -(-254 // wbits)
    else:
        nw = -wbits
        assert g.windows() == nw and g.window_bits() == 254 // nw
        assert capi.lib.sp_gens_table_bytes(g.h) == 131 * (nw + 254 - nw * (254 // nw)) * (1 << (254 // nw - 1)) * 128
    gb = g.compressed
    rng = random.Random(wbits)
    for rows, cols, blind, kind in [(1, 130, True, "uniform"), (6, 3, False, "edge"), (40, 128, True, "uniform"), (300, 64, True, "uniform"), (9, 17, False, "sparse"),
                                    (256, 128, True, "edge"), (2304, 20, False, "uniform")]:   # (the balanced form; nine row-blocks: the strip form / the queue form)
        Z = rand_scalars(rng, rows * cols, kind)
        bl = rand_scalars(rng, rows, "uniform") if blind else None
        got = g.commit_rows(mont_array(Z), rows, cols, mont_array(bl) if blind else None, g_off=0, h_idx=130)
        want = (ctypes.c_uint8 * (32 * rows))()
        assert orc.orc_commit_rows(gb[:32 * cols], sz(cols), gb[32 * 130:], mont_array(Z), sz(rows), sz(cols), mont_array(bl) if blind else None, want) == 0
        assert got == bytes(want), (wbits, rows, cols)
    idx = [130, 0, 7, 7, 99]
    S = rand_scalars(rng, 2 * len(idx))
    got = g.msm_indexed(idx, mont_array(S), rows=2)
    pts = b"".join(gb[32 * i:32 * i + 32] for i in idx)
    out = (ctypes.c_uint8 * 32)()
    for r in range(2):
        assert orc.orc_pt_msm(mont_array(S[r * 5:(r + 1) * 5]), pts, sz(5), out) == 1
        assert got[32 * r:32 * r + 32] == bytes(out)
    g.free()
    ctx.set_option("msm.wbits", 0); ctx.set_option("msm.windows", 0)


def test_narrow_wide_tables_switch_the_set_to_the_lds_staged_form(ctx, orc):
    """The default per generator set (core.hip gens_build; profiles/r5_ab_msm_forms.txt): a set whose gathered tables come out at <= 10 bits — HBM
    was short (here: a 1 GB table budget) — also gets the packed 10-bit tables and its row commitments of >= 512 rows run the LDS-staged form
    (msm_lds.hip); msm.form = 3 keeps the gathered forms. Same commitments either way, equal to the oracle's. A width FORCED with msm.wbits
    gets exactly the tables it asks for and no second set."""
    from spartan_amd import capi
    ctx.set_option("msm.wbits", 10)
    gf = capi.Gens(ctx, compressed=gens_bytes(orc, 513, b"gens_forced_10"))
    assert capi.lib.sp_gens_table_bytes(gf.h) == 514 * 26 * 512 * 128 and gf.window_bits() == 10
    gf.free()
    ctx.set_option("msm.wbits", 0); ctx.set_option("msm.table_gb", 1); ctx.set_option("msm.wide_gb", 1)
    try:
        n = 520
        g = capi.Gens(ctx, compressed=gens_bytes(orc, n, b"gens_auto_lds"))
        # 26 windows is what 1 GB holds for 521 points (6 x 9 + 20 x 10 bits: 46 x 256 entries); the packed tables are uniform 10-bit ones
        assert g.windows() == 26 and capi.lib.sp_gens_table_bytes(g.h) == (n + 1) * (46 * 256 * 128 + 26 * 512 * 96)    # both table kinds were built
        rows, cols = 640, n
        rng = random.Random(77)
        Z = rand_scalars(rng, rows * cols, "uniform")
        bl = rand_scalars(rng, rows, "uniform")
        gb = g.compressed
        want = (ctypes.c_uint8 * (32 * rows))()
        assert orc.orc_commit_rows(gb[:32 * cols], sz(cols), gb[32 * n:], mont_array(Z), sz(rows), sz(cols), mont_array(bl), want) == 0
        got_auto = g.commit_rows(mont_array(Z), rows, cols, mont_array(bl), g_off=0, h_idx=n)
        ctx.set_option("msm.form", 3)
        got_gathered = g.commit_rows(mont_array(Z), rows, cols, mont_array(bl), g_off=0, h_idx=n)
        assert got_auto == got_gathered == bytes(want)
        g.free()
    finally:
        ctx.set_option("msm.form", 0); ctx.set_option("msm.wbits", 0); ctx.set_option("msm.table_gb", 180); ctx.set_option("msm.wide_gb", 80)


def test_commit_rows_dev_and_offset(ctx, orc, gens40):
    from spartan_amd import capi
    rng = random.Random(5)
    rows, cols = 8, 16
    Z = rand_scalars(rng, rows * cols)
    t = capi.Table.upload(ctx, mont_array(Z), rows * cols)
    got = gens40.commit_rows(t, rows, cols, None, g_off=3, h_idx=39)
    g = gens40.compressed
    want = (ctypes.c_uint8 * (32 * rows))()
    assert orc.orc_commit_rows(g[32 * 3:32 * (3 + cols)], sz(cols), g[32 * 39:], mont_array(Z), sz(rows), sz(cols), None, want) == 0
    assert got == bytes(want)
    t.free()


def test_msm_indexed_matches_oracle(ctx, orc, gens40):
    rng = random.Random(6)
    idx = [39, 0, 7, 7, 20]
    S = rand_scalars(rng, 2 * len(idx))
    got = gens40.msm_indexed(idx, mont_array(S), rows=2)
    g = gens40.compressed
    pts = b"".join(g[32 * i:32 * i + 32] for i in idx)
    out = (ctypes.c_uint8 * 32)()
    for r in range(2):
        assert orc.orc_pt_msm(mont_array(S[r * 5:(r + 1) * 5]), pts, sz(5), out) == 1
        assert got[32 * r:32 * r + 32] == bytes(out)


@pytest.mark.parametrize("ell", [1, 3, 4, 5, 11])
def test_eq_expand_matches_oracle(ctx, orc, ell):
    from spartan_amd import capi
    rng = random.Random(ell)
    r = rand_scalars(rng, ell)
    t = capi.Table.eq(ctx, mont_array(r), ell)
    want = (ctypes.c_uint64 * (4 << ell))()
    orc.orc_eq_evals(mont_array(r), sz(ell), want)
    assert list(t.download()) == list(want)
    t.free()


@pytest.mark.parametrize("kind,ntabs", [(0, 2), (1, 3), (2, 4)])
@pytest.mark.parametrize("ell", [1, 2, 6, 13])
def test_sumcheck_eval_bind_matches_oracle(ctx, orc, kind, ntabs, ell):
    from spartan_amd import capi
    rng = random.Random(kind * 100 + ell)
    n = 1 << ell
    vals = [rand_scalars(rng, n) for _ in range(ntabs)]
    tabs = [capi.Table.upload(ctx, mont_array(v), n) for v in vals]
    host = [mont_array(v) for v in vals] + [None] * (4 - ntabs)
    length = n
    first = True
    while length >= 2:
        want = (ctypes.c_uint64 * 12)()
        orc.orc_sumcheck_eval(ctypes.c_int(kind), host[0], host[1], host[2], host[3], sz(length), want)
        if first:
            got = capi.sumcheck_eval(ctx, kind, tabs)
        nv = 8 if kind == 0 else 12
        assert list(got)[:nv] == list(want)[:nv], f"len {length}"
        r = mont_array([rng.randrange(Q)])
        for k in range(ntabs):
            orc.orc_bound_top(host[k], sz(length), r)
        length //= 2
        if length >= 4 and ell == 13 and length % 3 == 2:
            capi.sumcheck_bind_eval_start(ctx, kind, tabs, r)   # the same round in two halves (the ZK sum-check commits in between)
            with pytest.raises(capi.SpartanHipError):
                capi.sumcheck_bind_eval_start(ctx, kind, tabs, r)   # one pending round per context
            got = capi.sumcheck_bind_eval_collect(ctx)
            first = False
        elif length >= 2 and (ell % 2 == 0 or length > 4):
            got = capi.sumcheck_bind_eval(ctx, kind, tabs, r)   # fused path
            first = False
        else:
            capi.bind_top(ctx, tabs, r)                         # unfused path
            first = True
        for k in range(ntabs):
            assert len(tabs[k]) == length
            assert list(tabs[k].download(length)) == list(host[k])[:4 * length]
    hd = capi.heads(ctx, tabs)
    for k in range(ntabs):
        assert list(hd)[4 * k:4 * k + 4] == list(host[k])[:4]
    for t in tabs:
        t.free()


@pytest.mark.parametrize("v", [2, 5, 10, 13])
def test_vecmat_dot_evaluate_match_oracle(ctx, orc, v):
    from spartan_amd import capi
    rng = random.Random(v)
    n = 1 << v
    Z = rand_scalars(rng, n)
    Ls = 1 << (v // 2)
    Lv = rand_scalars(rng, Ls)
    t = capi.Table.upload(ctx, mont_array(Z), n)
    got = capi.vecmat(ctx, mont_array(Lv), Ls, t)
    want = (ctypes.c_uint64 * (4 * (n // Ls)))()
    orc.orc_bound_vecmat(mont_array(Z), sz(v), mont_array(Lv), want)
    assert list(got) == list(want)
    B = rand_scalars(rng, n)
    tb = capi.Table.upload(ctx, mont_array(B), n)
    d = capi.dot(ctx, t, tb, n)
    assert from_mont_limbs(d) == sum(a * b for a, b in zip(Z, B)) % Q
    r = rand_scalars(rng, v)
    e = capi.evaluate(ctx, t, mont_array(r), v)
    chi = (ctypes.c_uint64 * (4 * n))()
    orc.orc_eq_evals(mont_array(r), sz(v), chi)
    w = u64x4()
    orc.orc_dot(mont_array(Z), chi, sz(n), w)
    assert list(e) == list(w)
    t.free(); tb.free()


@pytest.mark.parametrize("per,cells,nlists,dist", [(8, 4, 3, "uniform"), (256, 16, 3, "uniform"), (1000, 1 << 10, 2, "uniform"), (4096, 8192, 3, "uniform"),
                                                   (4096, 64, 3, "one"), (1 << 16, 1 << 17, 3, "sorted")])
def test_addr_timestamps_match_reference_scan(ctx, per, cells, nlists, dist):
    """sp_addr_timestamps against the loop of AddrTimestamps::new (sparse_mlpoly.rs:221-254) restated in Python, with many
    operations per cell (the synthetic R1CS has almost none), a single hot cell, and the increasing addresses of a real instance"""
    from spartan_amd import capi
    rng = random.Random(per * 31 + cells)
    if dist == "uniform":
        lists = [[rng.randrange(cells) for _ in range(per)] for _ in range(nlists)]
    elif dist == "one":
        lists = [[7 if rng.random() < 0.9 else rng.randrange(cells) for _ in range(per)] for _ in range(nlists)]
    else:
        lists = [[(i + k) % cells for i in range(per)] for k in range(nlists)]
    audit = [0] * cells
    want_ts = []
    for ops in lists:            # one counter array across all lists, lists walked in order
        ts = []
        for a in ops:
            ts.append(audit[a]); audit[a] += 1
        want_ts.append(ts)
    ix = []
    for ops in lists:
        h = vp()
        assert capi.lib.sp_index_upload(ctx.h, (ctypes.c_uint64 * per)(*ops), sz(per), ctypes.byref(h)) == 0
        ix.append(h)
    dst = capi.Table.alloc(ctx, nlists * per + 5)
    aud = capi.Table.alloc(ctx, cells + 3)
    offs = (sz * nlists)(*[k * per + 5 for k in range(nlists)])
    rc = capi.lib.sp_addr_timestamps(ctx.h, (vp * nlists)(*ix), sz(nlists), sz(cells), dst.h, offs, aud.h, sz(3))
    assert rc == 0
    got = from_mont_array(dst.download(), nlists * per + 5)
    for k in range(nlists):
        assert got[5 + k * per:5 + (k + 1) * per] == want_ts[k]
    assert from_mont_array(aud.download(), cells + 3)[3:] == audit
    for h in ix:
        capi.lib.sp_index_free(h)
    dst.free(); aud.free()


def test_bind_top_heads_product_tree_many_and_round_body(ctx, orc, gens40):
    """the calls that issue several pieces of one protocol step together: sp_table_bind_top_heads (last sum-check round),
    sp_product_tree_many (ProductCircuit::new for circuits of one size), sp_sumcheck_bind_eval_commit (a ZK round body)"""
    from spartan_amd import capi
    rng = random.Random(2024)
    # --- last round: 37 tables of length 2 bound at r, remaining entries returned (Python ground truth)
    r = rng.randrange(Q)
    vals = [[rng.randrange(Q), rng.randrange(Q)] for _ in range(37)]
    tabs = [capi.Table.upload(ctx, mont_array(v), 2) for v in vals]
    out = (ctypes.c_uint64 * (4 * 37))()
    assert capi.lib.sp_table_bind_top_heads(ctx.h, (vp * 37)(*[t.h for t in tabs]), sz(37), mont_array([r]), out) == 0
    want = [(a + r * (b - a)) % Q for a, b in vals]
    assert from_mont_array(out, 37) == want
    assert [from_mont_array(t.download(1), 1)[0] for t in tabs] == want and all(len(t) == 1 for t in tabs)
    dup = (vp * 2)(tabs[0].h, tabs[0].h)   # a table listed twice would be bound twice: refused
    assert capi.lib.sp_table_bind_top_heads(ctx.h, dup, sz(2), mont_array([r]), out) != 0
    for t in tabs:
        t.free()
    # --- product trees of 5 circuits of 64 leaves in one go == one at a time (sp_product_tree is checked against the oracle by the proofs)
    n = 64
    leaves = [rand_scalars(rng, n) for _ in range(5)]
    many = [capi.Table.upload(ctx, mont_array(l + [0] * n), 2 * n) for l in leaves]
    assert capi.lib.sp_product_tree_many(ctx.h, (vp * 5)(*[t.h for t in many]), sz(5), sz(n)) == 0
    for l, t in zip(leaves, many):
        layer, off, got = l, 0, from_mont_array(t.download(), 2 * n)
        while len(layer) > 2:   # layer k+1[i] = left[i] * right[i] (product_tree.rs:36-56), stored behind layer k
            half = len(layer) // 2
            nxt = [layer[i] * layer[half + i] % Q for i in range(half)]
            off += len(layer)
            assert got[off:off + half] == nxt
            layer = nxt
        t.free()
    # --- hash layer + first multiplication layer in one pass, read and write set of a matrix together (sp_hash_layer_first), then the rest of
    # the tree (sp_product_tree_many_from .. 1): Layers::build_hash_layer (sparse_mlpoly.rs:529-604) and ProductCircuit::new in Python
    for n in (4, 64, 8192, 65536):   # 8192: above the one-launch tail (2048), so layer 2 comes from the per-layer kernel; 65536: two layers per launch (k_prod_layer2_many)
        addr, val, ts = [rng.randrange(n) for _ in range(n)], rand_scalars(rng, n), [rng.randrange(50) for _ in range(n)]
        rh, rm = rng.randrange(Q), rng.randrange(Q)
        hashed = lambda a, v, t: (t * rh * rh + v * rh + a - rm) % Q
        want_r = [hashed(addr[i], val[i], ts[i]) for i in range(n)]
        want_w = [hashed(addr[i], val[i], ts[i] + 1) for i in range(n)]
        want_i = [hashed(i, val[i], 0) for i in range(n)]   # init: addr = identity, ts = 0 (:572-584)
        ta, tv, tt = capi.Table.upload(ctx, mont_array(addr), n), capi.Table.upload(ctx, mont_array(val), n), capi.Table.upload(ctx, mont_array(ts), n)
        st = [capi.Table.upload(ctx, mont_array([0] * (2 * n)), 2 * n) for _ in range(3)]
        assert capi.lib.sp_hash_layer_first(ctx.h, ta.h, tv.h, tt.h, 0, sz(n), mont_array([rh]), mont_array([rm]), st[0].h, st[1].h) == 0
        assert capi.lib.sp_hash_layer_first(ctx.h, None, tv.h, None, 0, sz(n), mont_array([rh]), mont_array([rm]), st[2].h, None) == 0
        assert capi.lib.sp_hash_layer_first(ctx.h, ta.h, tv.h, tt.h, 1, sz(n), mont_array([rh]), mont_array([rm]), st[0].h, st[1].h) != 0   # a pair is ts and ts + 1
        assert capi.lib.sp_product_tree_many_from(ctx.h, (vp * 3)(*[t.h for t in st]), sz(3), sz(n), sz(1)) == 0
        for leaves_, t in zip((want_r, want_w, want_i), st):
            got = from_mont_array(t.download(), 2 * n)
            assert got[:n] == leaves_
            layer, off = leaves_, 0
            while len(layer) > 2:
                half = len(layer) // 2
                nxt = [layer[i] * layer[half + i] % Q for i in range(half)]
                off += len(layer)
                assert got[off:off + half] == nxt, (n, off)
                layer = nxt
            t.free()
        for t in (ta, tv, tt):
            t.free()
    # --- ZK round body: bind+evaluate and two small commitments in one call == the two calls made separately
    ell = 9
    A = [rand_scalars(rng, 1 << ell) for _ in range(4)]
    t1 = [capi.Table.upload(ctx, mont_array(a), 1 << ell) for a in A]
    t2 = [capi.Table.upload(ctx, mont_array(a), 1 << ell) for a in A]
    rr = mont_array([rng.randrange(Q)])
    idx = [3, 4, 5, 6, 39, 0, 39]
    S = rand_scalars(rng, 2 * len(idx))
    want_ev = capi.sumcheck_bind_eval(ctx, 2, t1, rr)
    want_pts = gens40.msm_indexed(idx, mont_array(S), rows=2)
    ev = (ctypes.c_uint64 * 12)(); pts = (ctypes.c_uint8 * 64)()
    rc = capi.lib.sp_sumcheck_bind_eval_commit(ctx.h, ctypes.c_int(2), (vp * 4)(*[t.h for t in t2]), sz(4), rr, ev, gens40.h,
                                               (ctypes.c_uint32 * len(idx))(*idx), sz(len(idx)), mont_array(S), sz(2), pts)
    assert rc == 0 and list(ev) == list(want_ev) and bytes(pts) == want_pts
    assert all(list(a.download()) == list(b.download()) for a, b in zip(t1, t2))
    for t in t1 + t2:
        t.free()


def test_polynomial_evaluation_known_answer_on_device(ctx):
    """dense_mlpoly.rs:433-452 check_polynomial_evaluation: Z = [1,2,1,4], r = [4,3] -> 28, via sp_evaluate and via the
    L/R factorisation (sp_vecmat + host dot) the PolyEvalProof uses"""
    from spartan_amd import capi
    t = capi.Table.upload(ctx, mont_array([1, 2, 1, 4]), 4)
    assert from_mont_limbs(capi.evaluate(ctx, t, mont_array([4, 3]), 2)) == 28
    L = [(1 - 4) % Q, 4]; R_ = [(1 - 3) % Q, 3]           # EqPolynomial::compute_factored_evals (dense_mlpoly.rs:86-98)
    LZ = from_mont_array(capi.vecmat(ctx, mont_array(L), 2, t), 2)
    assert sum(a * b for a, b in zip(LZ, R_)) % Q == 28
    chi = capi.Table.eq(ctx, mont_array([4, 3]), 2)
    assert from_mont_array(chi.download(), 4) == [(a * b) % Q for a in L for b in R_]  # check_memoized_factored_chis
    t.free(); chi.free()


def test_background_commit_overlaps_and_matches(ctx, orc, gens40):
    """sp_commit_rows_dev_begin / sp_job_wait: the commit runs on the background stream while main-stream calls proceed;
    both results are bit-exact (the job must observe Z as written before begin, and its scratch must not alias)."""
    from spartan_amd import capi
    rng = random.Random(77)
    rows, cols = 512, 32   # large enough to take the strip kernel path (not the windowed one)
    Z = rand_scalars(rng, rows * cols)
    t = capi.Table.upload(ctx, mont_array(Z), rows * cols)
    job = gens40.commit_rows_begin(t, rows, cols, g_off=2)
    # main-stream work issued while the job is in flight
    a = rand_scalars(rng, 1 << 12); b = rand_scalars(rng, 1 << 12)
    ta = capi.Table.upload(ctx, mont_array(a), 1 << 12); tb = capi.Table.upload(ctx, mont_array(b), 1 << 12)
    for _ in range(5):
        d = capi.dot(ctx, ta, tb, 1 << 12)
        assert from_mont_limbs(d) == sum(x * y for x, y in zip(a, b)) % Q
    sync = gens40.commit_rows(t, rows, cols, None, g_off=2, h_idx=39)
    got = gens40.commit_rows_wait(job)
    assert got == sync
    g = gens40.compressed
    want = (ctypes.c_uint8 * (32 * 4))()
    assert orc.orc_commit_rows(g[32 * 2:32 * (2 + cols)], sz(cols), g[32 * 39:], mont_array(Z[:4 * cols]), sz(4), sz(cols), None, want) == 0
    assert got[:128] == bytes(want)
    t.free(); ta.free(); tb.free()


def test_dot3_many_matches_reference_arithmetic(ctx):
    """sp_dot3_many: the six DotProductCircuit::evaluate of ProductLayerProof::prove (product_tree.rs:84-88) in one launch"""
    from spartan_amd import capi
    rng = random.Random(606)
    nt, n = 6, 3000
    vals = [[rand_scalars(rng, n) for _ in range(3)] for _ in range(nt)]
    tabs = [[capi.Table.upload(ctx, mont_array(v), n) for v in trip] for trip in vals]
    L = (vp * nt)(*[t[0].h for t in tabs]); R = (vp * nt)(*[t[1].h for t in tabs]); W = (vp * nt)(*[t[2].h for t in tabs])
    out = (ctypes.c_uint64 * (4 * nt))()
    assert capi.lib.sp_dot3_many(ctx.h, L, R, W, sz(nt), sz(n), out) == 0
    got = from_mont_array(out, nt)
    for k in range(nt):
        assert got[k] == sum(a * b * c for a, b, c in zip(*vals[k])) % Q, k
    for trip in tabs:
        for t in trip:
            t.free()


@pytest.mark.gpu
@pytest.mark.parametrize("opts,count", [({"msm__form": 3}, 11), ({"msm__form": 3, "msm__wbits": 13, "bg__eighths": 3}, 11),
                                        ({"msm__lds_bits": 10, "msm__form": 1}, 23), ({"msm__lds_bits": 8, "msm__form": 1, "bg__eighths": 3}, 23),
                                        ({"msm__lds_bits": 6, "msm__form": 1}, 23), ({"msm__lds_bits": 7, "msm__form": 1, "msm__wbits": 12}, 23),
                                        ({}, 27), ({"msm__form": 2, "msm__q_waves": 8, "msm__q_bg_waves": 4, "msm__q_units": 4, "msm__wbits": 12, "bg__eighths": 6}, 27),
                                        ({"msm__q_waves": 4, "msm__q_bg_waves": 12, "msm__q_units": 100, "msm__wbits": 15}, 27)])
def test_row_msm_forms_match_oracle(opts, count):
    """Every launch form of the fixed-base row MSM (DensePolynomial::commit_inner, src/dense_mlpoly.rs:164-177) against the oracle: the strip
    form with its short-scalar early exit (many row-blocks, the persistent background launch), the balanced (column, window) form with two
    entries in flight, another window width and background share, and the LDS-staged small-window form (msm_lds.hip: 10-bit sub-tables
    double-buffered in LDS; 8-, 7- and 6-bit ones — every width the option accepts since ADVICE r5) with shapes of its own — row-blocks that are not a multiple of a
    wavefront, two and three row-blocks, runs that start inside a scalar, the persistent background form — and the queue form (k_msm_q, the
    default since round 6: self-contained wavefronts with private LDS rings pulling runs from per-row-group queues) on the same shapes plus four
    of its own, co-resident and on a fenced share of the CUs, with items of 4, 32 and 100 units. The form is an option of the
    context; tests/msm_forms_worker.py runs in a process of its own per setting (11 shapes each, 12 more for the LDS form: blinds, zero
    rows, short / high-bit / carry-chain scalars, 1..8 row-blocks)."""
    import subprocess, sys
    from tests.helpers import options_env
    e = dict(os.environ, SPARTAN_OPTIONS=options_env(**opts))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "msm_forms_worker.py"), "7"], env=e, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0 and "MSM_FORMS_OK %d" % count in r.stdout, (r.stdout[-500:], r.stderr[-3000:])


@pytest.mark.gpu
def test_tables_pack_and_unpack_residues_round_trip():
    """sp_table_residue_split -> sp_tables_pack -> sp_tables_unpack_residues is the identity (the hand-over of the residue-sharded batched
    cubic sum-checks): W sub-tables of each of several tables, packed per shard, concatenated in shard order, scattered back."""
    from spartan_amd import capi
    ctx = capi.Ctx(0)
    rng = random.Random(3)
    W, n, nt = 4, 1 << 12, 5
    vals = [fast_scalars(rng, n) for _ in range(nt)]
    tabs = [capi.Table.upload(ctx, mont_bulk(v), n) for v in vals]
    sub = n // W
    buf = (ctypes.c_uint64 * (4 * W * nt * sub))()
    for g in range(W):
        subs = []
        for t in tabs:
            o = ctypes.c_void_p()
            assert capi.lib.sp_table_residue_split(ctx.h, t.h, sz(W), sz(g), ctypes.byref(o)) == 0
            subs.append(o)
        arr = (ctypes.c_void_p * nt)(*[x.value for x in subs])
        dst = ctypes.cast(ctypes.addressof(buf) + 32 * g * nt * sub, ctypes.POINTER(ctypes.c_uint64))
        assert capi.lib.sp_tables_pack(ctx.h, arr, sz(nt), sz(sub), dst) == 0
        for x in subs:
            capi.lib.sp_table_free(x)
    fresh = [capi.Table.upload(ctx, mont_bulk([0] * n), n) for _ in range(nt)]
    arr = (ctypes.c_void_p * nt)(*[t.h.value if hasattr(t.h, "value") else t.h for t in fresh])
    assert capi.lib.sp_tables_unpack_residues(ctx.h, arr, sz(nt), sz(W), sz(sub), buf) == 0
    for t, v in zip(fresh, vals):
        assert from_mont_bulk(t.download(), n) == v
    bad = (ctypes.c_void_p * 2)(arr[0], arr[0])
    assert capi.lib.sp_tables_unpack_residues(ctx.h, bad, sz(2), sz(W), sz(sub), buf) != 0   # every table once
    for t in tabs + fresh:
        t.free()
    ctx.close()
