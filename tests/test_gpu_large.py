"""Direct parity tests of the C-ABI entry points at the sizes where the THROUGHPUT launch plans are selected (the
small-shape tests in test_gpu_kernels.py reach only the latency forms): >= 2^17 elements for the F_q streaming kernels
(k_eq_outer, k_cubic_*_batched, k_sc_bind_eval, k_dot_many, k_dot3, k_spmv, k_eval_table, k_sparse_eval, k_hash_layer),
n = 4096 for the inner-product argument, and the full 1024 x 1024 witness-sized commit on every row (k_msm_rows with the
XCD tile order, and the persistent half-chip k_msm_rows_bg).

Ground truth: the oracle where it exports the operation (orc_eq_evals, orc_sumcheck_eval, orc_bound_top, orc_commit_rows,
orc_pt_msm), Python big integers restating the reference line by line elsewhere (each function cites it). Bit-exact."""
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


def up(ctx, vals):
    from spartan_amd import capi
    return capi.Table.upload(ctx, mont_bulk(vals), len(vals))


def fq1(x):
    return mont_bulk([x])


def test_eq_expand_long_table_matches_oracle(ctx, orc):
    """EqPolynomial::evals (dense_mlpoly.rs:68-84) at ell = 17 and 18: the outer-product path (k_eq_outer)."""
    from spartan_amd import capi
    for ell in (14, 17, 18):
        rng = random.Random(ell)
        r = fast_scalars(rng, ell)
        t = capi.Table.eq(ctx, mont_bulk(r), ell)
        want = (ctypes.c_uint64 * (4 << ell))()
        orc.orc_eq_evals(mont_bulk(r), sz(ell), want)
        assert bytes(t.download()) == bytes(want)
        t.free()


@pytest.mark.parametrize("kind,ntabs", [(0, 2), (2, 4)])
def test_zk_sumcheck_rounds_large_match_oracle(ctx, orc, kind, ntabs):
    """sumcheck.rs:460-469 / 624-652 + bound_poly_var_top at 2^17: the streaming k_sc_eval / k_sc_bind_eval (quarter > 8192)
    and the hand-over to the lane-parallel form as the tables shrink."""
    from spartan_amd import capi
    ell = 17
    n = 1 << ell
    rng = random.Random(900 + kind)
    vals = [fast_scalars(rng, n) for _ in range(ntabs)]
    tabs = [up(ctx, v) for v in vals]
    host = [mont_bulk(v) for v in vals] + [None] * (4 - ntabs)
    nv = 8 if kind == 0 else 12
    want = (ctypes.c_uint64 * 12)()
    orc.orc_sumcheck_eval(ctypes.c_int(kind), host[0], host[1], host[2], host[3], sz(n), want)
    got = capi.sumcheck_eval(ctx, kind, tabs)
    assert list(got)[:nv] == list(want)[:nv]
    length = n
    for _ in range(6):   # quarters 2^15 .. 2^10: three streaming rounds, three lane-parallel ones
        r = fq1(rng.getrandbits(250))
        for k in range(ntabs):
            orc.orc_bound_top(host[k], sz(length), r)
        length //= 2
        orc.orc_sumcheck_eval(ctypes.c_int(kind), host[0], host[1], host[2], host[3], sz(length), want)
        got = capi.sumcheck_bind_eval(ctx, kind, tabs, r)
        assert list(got)[:nv] == list(want)[:nv], length
    for k in range(ntabs):
        assert bytes(tabs[k].download(length)) == bytes(host[k])[:32 * length]
    for t in tabs:
        t.free()


def cubic_evals(A, B, C):
    """sumcheck.rs:290-357: evaluations of sum_i A(t) B(t) C(t) at t = 0, 2, 3 over the top-variable pairs"""
    h = len(A) // 2
    e0 = e2 = e3 = 0
    for i in range(h):
        a0, a1, b0, b1, c0, c1 = A[i], A[h + i], B[i], B[h + i], C[i], C[h + i]
        e0 += a0 * b0 * c0
        a2, b2, c2 = 2 * a1 - a0, 2 * b1 - b0, 2 * c1 - c0
        e2 += a2 * b2 * c2
        e3 += (a2 + a1 - a0) * (b2 + b1 - b0) * (c2 + c1 - c0)
    return [e0 % Q, e2 % Q, e3 % Q]


def bind(T, r):
    h = len(T) // 2
    return [(T[i] + r * (T[h + i] - T[i])) % Q for i in range(h)]


def test_batched_cubic_sumcheck_large_matches_reference_arithmetic(ctx):
    """prove_cubic_batched's round body (sumcheck.rs:287-393) at 2^17: 3 'par' instances sharing poly_C_par (bound once, out
    of place) and 2 'seq' instances with their own C; k_cubic_eval_batched, then k_cubic_bind_eval_batched for the rounds
    with quarter > 8192, then the lane-parallel form."""
    from spartan_amd import capi
    n = 1 << 17
    rng = random.Random(4242)
    npar, nseq = 3, 2
    A = [fast_scalars(rng, n) for _ in range(npar + nseq)]
    B = [fast_scalars(rng, n) for _ in range(npar + nseq)]
    Cpar = fast_scalars(rng, n)
    Cseq = [fast_scalars(rng, n) for _ in range(nseq)]
    tA, tB = [up(ctx, a) for a in A], [up(ctx, b) for b in B]
    tCpar, tCseq = up(ctx, Cpar), [up(ctx, c) for c in Cseq]
    ni = npar + nseq
    hA = (vp * ni)(*[t.h for t in tA]); hB = (vp * ni)(*[t.h for t in tB])
    hC = (vp * ni)(*([tCpar.h] * npar + [t.h for t in tCseq]))
    C = [Cpar] * npar + Cseq
    out = (ctypes.c_uint64 * (12 * ni))()
    assert capi.lib.sp_sumcheck_eval_batched(ctx.h, hA, hB, hC, sz(ni), out) == 0
    got = from_mont_bulk(out, 3 * ni)
    for k in range(ni):
        assert got[3 * k:3 * k + 3] == cubic_evals(A[k], B[k], C[k]), k
    for rnd in range(4):  # lengths 2^17 -> 2^13: quarters 2^15, 2^14 streaming; 2^13, 2^12 lane-parallel
        r = rng.getrandbits(251)
        A = [bind(a, r) for a in A]; B = [bind(b, r) for b in B]
        Cpar = bind(Cpar, r); Cseq = [bind(c, r) for c in Cseq]
        C = [Cpar] * npar + Cseq
        assert capi.lib.sp_sumcheck_bind_eval_batched(ctx.h, hA, hB, hC, sz(ni), fq1(r), out) == 0
        got = from_mont_bulk(out, 3 * ni)
        for k in range(ni):
            assert got[3 * k:3 * k + 3] == cubic_evals(A[k], B[k], C[k]), (rnd, k)
    ln = len(Cpar)
    assert from_mont_bulk(tCpar.download(ln), ln) == Cpar and len(tCpar) == ln
    assert from_mont_bulk(tA[4].download(ln), ln) == A[4] and from_mont_bulk(tCseq[1].download(ln), ln) == Cseq[1]
    for t in tA + tB + tCseq + [tCpar]:
        t.free()


def test_batched_cubic_sumcheck_with_the_eq_table_as_a_factor(ctx):
    """k_cubic_eval_batched_eq / k_cubic_bind_eval_batched_eq (spartan_hip.h: sp_sumcheck_*_batched_eq) at 2^17: 3 product-circuit
    instances that share an eq table and 2 generic instances. Per product-circuit instance the device must return
    q(t) = sum_x A(t,x) B(t,x) C_original[x] at t = 0, 2 over the ORIGINAL eq table's leading entries (the table is never bound, and must
    come back untouched), per generic instance the evaluations at t = 0, 1, 2, 3 of sumcheck.rs:290-357 (plus t = 1); A, B and the generic C
    tables are bound as by the generic kernel. Then the hand-over: sp_table_scale_prefix. Python integers are the reference."""
    from spartan_amd import capi
    n = 1 << 17
    rng = random.Random(777)
    npar, nseq = 3, 2
    ni = npar + nseq
    A = [fast_scalars(rng, n) for _ in range(ni)]
    B = [fast_scalars(rng, n) for _ in range(ni)]
    Ceq = fast_scalars(rng, n)   # any table will do for the kernel's arithmetic: it only ever reads the leading entries
    Cseq = [fast_scalars(rng, n) for _ in range(nseq)]
    tA, tB = [up(ctx, a) for a in A], [up(ctx, b) for b in B]
    tCeq, tCseq = up(ctx, Ceq), [up(ctx, c) for c in Cseq]
    hA = (vp * ni)(*[t.h for t in tA]); hB = (vp * ni)(*[t.h for t in tB])
    hC = (vp * ni)(*([tCeq.h] * npar + [t.h for t in tCseq]))
    line = lambda u, v, t: (u + t * (v - u)) % Q

    def want(A, B, Cseq):
        h = len(A[0]) // 2
        w = []
        for k in range(ni):
            if k < npar:
                w += [sum(line(A[k][z], A[k][h + z], t) * line(B[k][z], B[k][h + z], t) * Ceq[z] for z in range(h)) % Q for t in (0, 2)] + [0, 0]
            else:
                c = Cseq[k - npar]
                w += [sum(line(A[k][z], A[k][h + z], t) * line(B[k][z], B[k][h + z], t) * line(c[z], c[h + z], t) for z in range(h)) % Q for t in range(4)]
        return w
    out = (ctypes.c_uint64 * (16 * ni))()
    assert capi.lib.sp_sumcheck_eval_batched_eq(ctx.h, hA, hB, hC, sz(ni), sz(npar), out) == 0
    assert from_mont_bulk(out, 4 * ni) == want(A, B, Cseq)
    r = rng.getrandbits(251)
    A = [bind(a, r) for a in A]; B = [bind(b, r) for b in B]; Cseq = [bind(c, r) for c in Cseq]
    assert capi.lib.sp_sumcheck_bind_eval_batched_eq(ctx.h, hA, hB, hC, sz(ni), sz(npar), fq1(r), out) == 0   # 2^17 -> 2^16
    assert from_mont_bulk(out, 4 * ni) == want(A, B, Cseq)
    ln = n // 2
    assert len(tA[0]) == ln and len(tCseq[0]) == ln and len(tCeq) == n
    assert from_mont_bulk(tA[1].download(ln), ln) == A[1] and from_mont_bulk(tB[4].download(ln), ln) == B[4]
    assert from_mont_bulk(tCseq[1].download(ln), ln) == Cseq[1]
    assert from_mont_bulk(tCeq.download(n), n) == Ceq          # read, never written
    r2 = rng.getrandbits(250)
    A = [bind(a, r2) for a in A]; B = [bind(b, r2) for b in B]; Cseq = [bind(c, r2) for c in Cseq]
    assert capi.lib.sp_sumcheck_bind_eval_batched_eq(ctx.h, hA, hB, hC, sz(ni), sz(npar), fq1(r2), out) == 0  # 2^16 -> 2^15: the last factored length
    assert from_mont_bulk(out, 4 * ni) == want(A, B, Cseq)
    assert capi.lib.sp_sumcheck_bind_eval_batched_eq(ctx.h, hA, hB, hC, sz(ni), sz(npar), fq1(r2), out) != 0  # below 65536 entries: no factored form
    k = rng.getrandbits(249)
    assert capi.lib.sp_table_scale_prefix(ctx.h, tCeq.h, sz(n // 4), fq1(k)) == 0
    assert len(tCeq) == n // 4 and from_mont_bulk(tCeq.download(n // 4), n // 4) == [x * k % Q for x in Ceq[:n // 4]]
    for t in tA + tB + tCseq + [tCeq]:
        t.free()


def test_dot_many_dot3_hash_layer_gather_large(ctx):
    """HashLayerProof::prove's shared-chi evaluations (sparse_mlpoly.rs:722-835 -> k_dot_many), DotProductCircuit::evaluate
    (product_tree.rs:84-88 -> k_dot3), Layers::build_hash_layer (sparse_mlpoly.rs:529-604 -> k_hash_layer) and
    AddrTimestamps::deref_mem (:256-265 -> k_gather) at 2^17 elements."""
    from spartan_amd import capi
    n = 1 << 17
    rng = random.Random(1717)
    chi = fast_scalars(rng, n)
    T = [fast_scalars(rng, n) for _ in range(5)]
    tchi, tT = up(ctx, chi), [up(ctx, t) for t in T]
    out = (ctypes.c_uint64 * 20)()
    assert capi.lib.sp_dot_many(ctx.h, tchi.h, (vp * 5)(*[t.h for t in tT]), sz(5), out) == 0
    assert from_mont_bulk(out, 5) == [sum(a * b for a, b in zip(chi, t)) % Q for t in T]
    o4 = (ctypes.c_uint64 * 4)()
    off = 1 << 16   # the right half of a split DotProductCircuit (product_tree.rs:90-109)
    assert capi.lib.sp_dot3(ctx.h, tT[0].h, tT[1].h, tT[2].h, sz(off), sz(n - off), o4) == 0
    assert from_mont_bulk(o4, 1)[0] == sum(T[0][i] * T[1][i] * T[2][i] for i in range(off, n)) % Q
    # hash layers: (ts + inc) * r^2 + val * r + addr - r_multiset, with addr = identity / ts = 0 variants (init, audit, read, write)
    rh, rm = rng.getrandbits(250), rng.getrandbits(249)
    dst = capi.Table.alloc(ctx, n + 7)
    for addr, ts, inc in ((None, None, 0), (None, 3, 0), (4, 3, 0), (4, 3, 1)):
        rc = capi.lib.sp_hash_layer(ctx.h, tT[addr].h if addr is not None else None, tT[1].h, tT[ts].h if ts is not None else None, ctypes.c_int(inc),
                                    sz(n), fq1(rh), fq1(rm), dst.h, sz(7))
        assert rc == 0
        want = [(((T[ts][i] if ts is not None else 0) + inc) * rh * rh + T[1][i] * rh + (T[addr][i] if addr is not None else i) - rm) % Q for i in range(n)]
        assert from_mont_bulk(dst.download(n, 7), n) == want, (addr, ts, inc)
    # deref_mem: dst[off + i] = mem[addr[i]]
    addrs = [rng.randrange(n) for _ in range(n)]
    ix = vp()
    assert capi.lib.sp_index_upload(ctx.h, (ctypes.c_uint64 * n)(*addrs), sz(n), ctypes.byref(ix)) == 0
    g = capi.Table.alloc(ctx, n + 3)
    assert capi.lib.sp_gather(ctx.h, tT[2].h, ix, g.h, sz(3)) == 0
    assert from_mont_bulk(g.download(n, 3), n) == [T[2][a] for a in addrs]
    f = capi.Table.alloc(ctx, n)   # DensePolynomial::from_usize (dense_mlpoly.rs:274-280)
    assert capi.lib.sp_table_from_index(ctx.h, ix, f.h, sz(0)) == 0
    assert from_mont_bulk(f.download(), n) == addrs
    capi.lib.sp_index_free(ix)
    for t in tT + [tchi, dst, g, f]:
        t.free()


def test_sparse_matrix_products_large(ctx):
    """SparseMatPolynomial::multiply_vec (sparse_mlpoly.rs:454-464), compute_eval_table_sparse x3 combined as
    r1csproof.rs:275-283, evaluate_with_tables (:429-438) on matrices with 2^17 rows/columns, several entries per row and
    per column, empty rows, and duplicate (row, col) pairs (the synthetic instances have exactly one entry per row)."""
    from spartan_amd import capi
    nr = nc = 1 << 17
    rng = random.Random(31337)
    mats = []
    for k in range(3):
        nnz = (1 << 17) + 1000 * k
        rows = [rng.randrange(nr) if rng.random() < 0.7 else rng.randrange(64) for _ in range(nnz)]   # a few crowded rows
        cols = [rng.randrange(nc) if rng.random() < 0.7 else nc - 1 - rng.randrange(32) for _ in range(nnz)]
        vals = fast_scalars(rng, nnz)
        rows[5], cols[5] = rows[4], cols[4]   # duplicate coordinate: both entries count
        mats.append((rows, cols, vals))
    hs = []
    for rows, cols, vals in mats:
        h = vp()
        n = len(rows)
        rc = capi.lib.sp_sparse_upload(ctx.h, (ctypes.c_uint64 * n)(*rows), (ctypes.c_uint64 * n)(*cols), mont_bulk(vals), sz(n), sz(nr), sz(nc), ctypes.byref(h))
        assert rc == 0
        hs.append(h)
    z = fast_scalars(rng, nc)
    tz = up(ctx, z)
    for (rows, cols, vals), h in zip(mats, hs):
        o = vp()
        assert capi.lib.sp_sparse_mulvec(ctx.h, h, tz.h, ctypes.byref(o)) == 0
        t = capi.Table(ctx, o)
        want = [0] * nr
        for r_, c_, v_ in zip(rows, cols, vals):
            want[r_] += v_ * z[c_]
        assert from_mont_bulk(t.download(), nr) == [w % Q for w in want]
        t.free()
    rx = fast_scalars(rng, nr)
    trx = up(ctx, rx)
    w = fast_scalars(rng, 3)
    o = vp()
    assert capi.lib.sp_sparse_eval_table(ctx.h, (vp * 3)(*hs), mont_bulk(w), sz(3), trx.h, ctypes.byref(o)) == 0
    t = capi.Table(ctx, o)
    want = [0] * nc
    for k, (rows, cols, vals) in enumerate(mats):
        for r_, c_, v_ in zip(rows, cols, vals):
            want[c_] += w[k] * rx[r_] * v_
    assert from_mont_bulk(t.download(), nc) == [x % Q for x in want]
    t.free()
    ty = up(ctx, z)
    for (rows, cols, vals), h in zip(mats, hs):
        o4 = (ctypes.c_uint64 * 4)()
        assert capi.lib.sp_sparse_evaluate(ctx.h, h, trx.h, ty.h, o4) == 0
        assert from_mont_bulk(o4, 1)[0] == sum(rx[r_] * z[c_] * v_ for r_, c_, v_ in zip(rows, cols, vals)) % Q
    # the three evaluations as one queued job on the low-priority stream (what SNARK::prove starts when ry is known): same values
    job = vp()
    assert capi.lib.sp_sparse_evaluate_begin(ctx.h, (vp * 3)(*hs), sz(3), trx.h, ty.h, ctypes.byref(job)) == 0
    o12 = (ctypes.c_uint64 * 12)()
    assert capi.lib.sp_job_wait(job, ctypes.cast(o12, ctypes.POINTER(ctypes.c_uint8))) == 0
    assert from_mont_bulk(o12, 3) == [sum(rx[r_] * z[c_] * v_ for r_, c_, v_ in zip(rows, cols, vals)) % Q for rows, cols, vals in mats]
    for h in hs:
        capi.lib.sp_sparse_free(h)
    for t in (tz, trx, ty):
        t.free()


def test_vecmat_and_evaluate_large_match_oracle(ctx, orc):
    """DensePolynomial::bound (dense_mlpoly.rs:206-213) and ::evaluate (:236-242) at 2^18 (512 x 512 view)."""
    from spartan_amd import capi
    v = 18
    n = 1 << v
    rng = random.Random(1818)
    Z = fast_scalars(rng, n)
    Ls = 1 << (v // 2)
    Lv = fast_scalars(rng, Ls)
    t = up(ctx, Z)
    got = capi.vecmat(ctx, mont_bulk(Lv), Ls, t)
    want = (ctypes.c_uint64 * (4 * (n // Ls)))()
    orc.orc_bound_vecmat(mont_bulk(Z), sz(v), mont_bulk(Lv), want)
    assert bytes(got) == bytes(want)
    r = fast_scalars(rng, v)
    e = capi.evaluate(ctx, t, mont_bulk(r), v)
    chi = (ctypes.c_uint64 * (4 * n))()
    orc.orc_eq_evals(mont_bulk(r), sz(v), chi)
    w = u64x4()
    orc.orc_dot(mont_bulk(Z), chi, sz(n), w)
    assert list(e) == list(w)
    t.free()


def _ipa_against_folded_generators(ctx, orc, n, comp, a, b, rng):
    """BulletReductionProof::prove (nizk/bullet.rs:32-132) as the reference computes it — G folded every round,
    L = <a_L, G_R> + c_L Q + blind_L H over the FOLDED generators — through the oracle's point arithmetic only
    (orc_pt_msm), against sp_ipa_*, which never folds G (fixed-base rows over the original generators).
    comp: n + 2 compressed points, G[0..n), Q = P[n], H = P[n+1]."""
    from spartan_amd import capi
    g = capi.Gens(ctx, compressed=comp)
    P = [comp[32 * i:32 * i + 32] for i in range(n + 2)]
    a, b = list(a), list(b)
    qs = rng.getrandbits(250)
    ipa = vp()
    assert capi.lib.sp_ipa_begin(ctx.h, g.h, sz(0), sz(n), sz(n), sz(n + 1), fq1(qs), mont_bulk(a), mont_bulk(b), ctypes.byref(ipa)) == 0
    out = (ctypes.c_uint8 * 32)()

    def msm(scalars, points):
        assert orc.orc_pt_msm(mont_bulk(scalars), b"".join(points), sz(len(points)), out) == 1
        return bytes(out)
    Qp = msm([qs], [P[n]])   # gens_1.scale(r) (nizk/mod.rs:479-480)
    G = P[:n]
    cur = n
    while cur > 1:
        h = cur // 2
        bl, br = rng.getrandbits(250), rng.getrandbits(249)
        cL = sum(a[i] * b[h + i] for i in range(h)) % Q
        cR = sum(a[h + i] * b[i] for i in range(h)) % Q
        wantL = msm(a[:h] + [cL, bl], G[h:] + [Qp, P[n + 1]])     # bullet.rs:83-89
        wantR = msm(a[h:] + [cR, br], G[:h] + [Qp, P[n + 1]])     # :91-97
        L = (ctypes.c_uint8 * 32)(); Rr = (ctypes.c_uint8 * 32)()
        assert capi.lib.sp_ipa_round_lr(ipa, fq1(bl), fq1(br), L, Rr) == 0
        assert bytes(L) == wantL and bytes(Rr) == wantR, cur
        u = rng.getrandbits(251) | 1
        ui = pow(u, Q - 2, Q)
        assert capi.lib.sp_ipa_round_fold(ipa, fq1(u), fq1(ui)) == 0
        a = [(a[i] * u + ui * a[h + i]) % Q for i in range(h)]     # :105-106
        b = [(b[i] * ui + u * b[h + i]) % Q for i in range(h)]
        G = [msm([ui, u], [G[i], G[h + i]]) for i in range(h)]     # :108
        cur = h
    ah = (ctypes.c_uint64 * 4)(); bh = (ctypes.c_uint64 * 4)(); gh = (ctypes.c_uint8 * 32)()
    d, r = rng.getrandbits(250), rng.getrandbits(250)
    want_delta = msm([d, r], [G[0], P[n + 1]])                     # nizk/mod.rs:496-501
    # the end of the argument in one call: on the calling thread's core from the last round's row sums (ipa.hip, ipa_finish_on_host) ...
    assert capi.lib.sp_ipa_finish_commit(ipa, fq1(d), fq1(r), ah, bh, out) == 0
    assert from_mont_bulk(ah, 1) == a and from_mont_bulk(bh, 1) == b and bytes(out) == want_delta
    # ... and on the device (the fold applied, g_hat as a fixed-base row over all n generators)
    assert capi.lib.sp_ipa_finish(ipa, ah, bh, gh) == 0
    assert from_mont_bulk(ah, 1) == a and from_mont_bulk(bh, 1) == b and bytes(gh) == G[0]
    assert capi.lib.sp_ipa_commit_ghat(ipa, fq1(d), fq1(r), out) == 0
    assert bytes(out) == want_delta
    capi.lib.sp_ipa_free(ipa)
    g.free()


def test_inner_product_argument_n4096_matches_folded_generators(ctx, orc):
    """n = 4096: the opening size of a 2^20 proof (two trees of ~8 levels per row and round, core.hip k_ipa_round)."""
    n = 4096
    rng = random.Random(4096)
    comp = gens_bytes(orc, n + 1, b"gens_ipa_test")   # G[0..n), Q = P[n], H = P[n+1]
    _ipa_against_folded_generators(ctx, orc, n, comp, fast_scalars(rng, n), fast_scalars(rng, n), rng)


@pytest.mark.parametrize("case", ["repeated_generators_equal_scalars", "repeated_generators_opposite_scalars", "zero_vector", "sparse_vector", "plain_small"])
def test_inner_product_argument_exceptional_sums(ctx, orc, case):
    """The tree of k_ipa_round adds with the dedicated (two-multiplication) formula, which is not complete (core.hip,
    pt10_tree_quad_ded): the neutral element is kept away from it by marks, P = Q collapses to the all-zero quadruple and the host
    re-runs the round with the unified formula. The cases that reach those paths, against the reference's folded-generator algorithm:
    a generator list that repeats points under equal scalars (two sub-trees with limb-identical sums: the re-run), under opposite
    scalars (a computed neutral element as an operand), an all-zero vector (every leaf is the marked neutral element), a vector with a
    few non-zero entries (marked and unmarked operands mixed), and an ordinary small argument. Caller-supplied generator lists (sp_gens_upload,
    as here) get the complete tree by default since round 5 — the dedicated tree cannot detect every exceptional pair — so the test turns the
    dedicated tree on for them (option ipa.dedicated_uploaded) to keep driving its marks and its re-run; the last case runs the default."""
    n = 16
    rng = random.Random(sum(map(ord, case)))
    ctx.set_option("testing.unlock", 1); ctx.set_option("ipa.dedicated_uploaded", 0 if case == "plain_small" else 1)
    base = gens_bytes(orc, n + 1, b"gens_ipa_exc")
    P = [base[32 * i:32 * i + 32] for i in range(n + 2)]
    a, b = fast_scalars(rng, n), fast_scalars(rng, n)
    if case.startswith("repeated_generators"):
        for i in range(0, n, 2):
            P[i + 1] = P[i]                      # G[2k+1] = G[2k]
            a[i + 1] = a[i] if case.endswith("equal_scalars") else (Q - a[i]) % Q
    elif case == "zero_vector":
        a = [0] * n
    elif case == "sparse_vector":
        a = [a[i] if i in (3, 12) else 0 for i in range(n)]
    _ipa_against_folded_generators(ctx, orc, n, b"".join(P), a, b, rng)
    if case == "repeated_generators_equal_scalars":
        # the re-run really happened: with it switched off (test-only switch) the same argument must fail in its first round
        import os
        from spartan_amd import capi
        ctx.set_option("ipa.rerun_exceptional", 0)
        try:
            g = capi.Gens(ctx, compressed=b"".join(P))
            ipa = vp()
            assert capi.lib.sp_ipa_begin(ctx.h, g.h, sz(0), sz(n), sz(n), sz(n + 1), fq1(5), mont_bulk(a), mont_bulk(b), ctypes.byref(ipa)) == 0
            L = (ctypes.c_uint8 * 32)(); Rr = (ctypes.c_uint8 * 32)()
            assert capi.lib.sp_ipa_round_lr(ipa, fq1(1), fq1(2), L, Rr) != 0
            capi.lib.sp_ipa_free(ipa)
            g.free()
        finally:
            ctx.set_option("ipa.rerun_exceptional", 1)
    ctx.set_option("ipa.dedicated_uploaded", 0); ctx.set_option("testing.unlock", 0)


def test_witness_sized_commit_every_row_matches_oracle(ctx, orc):
    """DensePolynomial::commit_inner (dense_mlpoly.rs:164-177) at the 2^20 headline shape, 1024 rows x 1024 columns with
    blinds: every one of the 1024 commitments of k_msm_rows (XCD tile order, two-pass reduce, one-lane-per-row encode) against
    orc_commit_rows; then the same rows through the persistent half-chip background kernel (k_msm_rows_bg, no blinds)."""
    from spartan_amd import capi
    rows = cols = 1024
    rng = random.Random(20)
    comp = gens_bytes(orc, cols, b"gens_r1cs_sat")
    g = capi.Gens(ctx, compressed=comp)
    Z = fast_scalars(rng, rows * cols)
    bl = fast_scalars(rng, rows)
    Zm, blm = mont_bulk(Z), mont_bulk(bl)
    t = capi.Table.upload(ctx, Zm, rows * cols)
    got = g.commit_rows(t, rows, cols, blm, g_off=0, h_idx=cols)
    orc.orc_set_threads(ctypes.c_int(min(32, __import__("os").cpu_count() or 1)))
    want = (ctypes.c_uint8 * (32 * rows))()
    assert orc.orc_commit_rows(comp[:32 * cols], sz(cols), comp[32 * cols:], Zm, sz(rows), sz(cols), blm, want) == 0
    assert got == bytes(want)
    job = g.commit_rows_begin(t, rows, cols, g_off=0)
    got_bg = g.commit_rows_wait(job)
    assert orc.orc_commit_rows(comp[:32 * cols], sz(cols), comp[32 * cols:], Zm, sz(rows), sz(cols), None, want) == 0
    assert got_bg == bytes(want)
    orc.orc_set_threads(ctypes.c_int(1))
    t.free(); g.free()


@pytest.mark.parametrize("ell", [2, 3, 4, 5, 9, 13, 15])
def test_two_rounds_per_launch_match_reference_arithmetic(ctx, ell):
    """sp_sumcheck_eval_coeffs_batched / sp_sumcheck_bind2_eval_batched (spark.hip k_cubic_bind2_eval): two rounds of
    prove_cubic_batched (sumcheck.rs:287-393) per device round trip. Checked against the reference arithmetic in Python:
    the evaluations of each round, the cubic (M0, M3, T1, T2) that predicts the round after the next bind, the doubly bound
    tables, and the final claims; 3 instances share their C table, 2 own theirs. 2^15 = the longest tables the driver
    sends down this path."""
    from spartan_amd import capi
    n = 1 << ell
    rng = random.Random(9100 + ell)
    npar, nseq = 3, 2
    ni = npar + nseq
    A = [fast_scalars(rng, n) for _ in range(ni)]
    B = [fast_scalars(rng, n) for _ in range(ni)]
    Cpar = fast_scalars(rng, n)
    Cseq = [fast_scalars(rng, n) for _ in range(nseq)]
    tA, tB = [up(ctx, a) for a in A], [up(ctx, b) for b in B]
    tCpar, tCseq = up(ctx, Cpar), [up(ctx, c) for c in Cseq]
    hA = (vp * ni)(*[t.h for t in tA]); hB = (vp * ni)(*[t.h for t in tB])
    hC = (vp * ni)(*([tCpar.h] * npar + [t.h for t in tCseq]))
    Cs = lambda: [Cpar] * npar + Cseq
    ev = (ctypes.c_uint64 * (12 * ni))(); co = (ctypes.c_uint64 * (48 * ni))()
    heads = (ctypes.c_uint64 * (4 * (2 * ni + 1 + nseq)))()
    half = pow(2, Q - 2, Q)

    def predicted(coeffs, r):   # E(t; r) from (M0, M3, T1, T2), as the host driver evaluates it
        out = []
        om = (1 - r) % Q
        for k in range(3 * ni):
            M0, M3, T1, T2 = coeffs[4 * k:4 * k + 4]
            M1 = ((T1 - T2) * half - M3) % Q; M2 = ((T1 + T2) * half - M0) % Q
            out.append((M0 * om ** 3 + M1 * om ** 2 * r + M2 * om * r ** 2 + M3 * r ** 3) % Q)
        return out
    assert capi.lib.sp_sumcheck_eval_coeffs_batched(ctx.h, hA, hB, hC, sz(ni), None, ev, co if n >= 4 else None) == 0
    got = from_mont_bulk(ev, 3 * ni)
    for k in range(ni):
        assert got[3 * k:3 * k + 3] == cubic_evals(A[k], B[k], Cs()[k]), k
    length = n
    while length >= 4:
        coeffs = from_mont_bulk(co, 12 * ni)
        r0, r1 = rng.getrandbits(251), rng.getrandbits(250)
        A = [bind(a, r0) for a in A]; B = [bind(b, r0) for b in B]; Cpar = bind(Cpar, r0); Cseq = [bind(c, r0) for c in Cseq]
        want_mid = [x for k in range(ni) for x in cubic_evals(A[k], B[k], Cs()[k])] if length >= 4 else None
        assert predicted(coeffs, r0) == want_mid, length          # the cubic predicts the round after the bind at r0
        A = [bind(a, r1) for a in A]; B = [bind(b, r1) for b in B]; Cpar = bind(Cpar, r1); Cseq = [bind(c, r1) for c in Cseq]
        length //= 4
        rc = capi.lib.sp_sumcheck_bind2_eval_batched(ctx.h, hA, hB, hC, sz(ni), fq1(r0), fq1(r1), None, ev if length >= 2 else None, co if length >= 4 else None,
                                                     heads if length == 1 else None)
        assert rc == 0
        assert len(tA[0]) == length and len(tCpar) == length and len(tCseq[0]) == length
        if length >= 2:
            got = from_mont_bulk(ev, 3 * ni)
            for k in range(ni):
                assert got[3 * k:3 * k + 3] == cubic_evals(A[k], B[k], Cs()[k]), (length, k)
        assert from_mont_bulk(tA[2].download(length), length) == A[2] and from_mont_bulk(tCpar.download(length), length) == Cpar
        assert from_mont_bulk(tB[4].download(length), length) == B[4] and from_mont_bulk(tCseq[1].download(length), length) == Cseq[1]
    if length == 1:
        want = []
        for k in range(ni):
            want += [A[k][0], B[k][0]]
        want += [Cpar[0]] + [c[0] for c in Cseq]
        assert from_mont_bulk(heads, 2 * ni + 1 + nseq) == want
    for t in tA + tB + tCseq + [tCpar]:
        t.free()
    if ell < 4:
        return
    # weighted form (the `coeffs` of sumcheck.rs:359-369 applied on the device, instances summed) and the single-bind form
    # (r1 = NULL: the round at which the tables become short)
    A = [fast_scalars(rng, n) for _ in range(ni)]; B = [fast_scalars(rng, n) for _ in range(ni)]
    Cpar = fast_scalars(rng, n); Cseq = [fast_scalars(rng, n) for _ in range(nseq)]
    w = fast_scalars(rng, ni)
    tA, tB = [up(ctx, a) for a in A], [up(ctx, b) for b in B]
    tCpar, tCseq = up(ctx, Cpar), [up(ctx, c) for c in Cseq]
    hA = (vp * ni)(*[t.h for t in tA]); hB = (vp * ni)(*[t.h for t in tB])
    hC = (vp * ni)(*([tCpar.h] * npar + [t.h for t in tCseq]))
    ev1 = (ctypes.c_uint64 * 12)(); co1 = (ctypes.c_uint64 * 48)()
    comb = lambda: [sum(w[k] * cubic_evals(A[k], B[k], Cs()[k])[t_] for k in range(ni)) % Q for t_ in range(3)]
    assert capi.lib.sp_sumcheck_eval_coeffs_batched(ctx.h, hA, hB, hC, sz(ni), mont_bulk(w), ev1, co1) == 0
    assert from_mont_bulk(ev1, 3) == comb()
    c12 = from_mont_bulk(co1, 12)
    r0 = rng.getrandbits(251)
    A = [bind(a, r0) for a in A]; B = [bind(b, r0) for b in B]; Cpar = bind(Cpar, r0); Cseq = [bind(c, r0) for c in Cseq]
    om = (1 - r0) % Q
    pred = []
    for t_ in range(3):
        M0, M3, T1, T2 = c12[4 * t_:4 * t_ + 4]
        M1 = ((T1 - T2) * half - M3) % Q; M2 = ((T1 + T2) * half - M0) % Q
        pred.append((M0 * om ** 3 + M1 * om ** 2 * r0 + M2 * om * r0 ** 2 + M3 * r0 ** 3) % Q)
    assert pred == comb()
    assert capi.lib.sp_sumcheck_bind2_eval_batched(ctx.h, hA, hB, hC, sz(ni), fq1(r0), None, mont_bulk(w), ev1, co1, None) == 0   # one bind
    assert from_mont_bulk(ev1, 3) == comb() and len(tA[0]) == n // 2 and len(tCpar) == n // 2
    assert from_mont_bulk(tA[1].download(n // 2), n // 2) == A[1] and from_mont_bulk(tCpar.download(n // 2), n // 2) == Cpar
    for t in tA + tB + tCseq + [tCpar]:
        t.free()


@pytest.mark.parametrize("ell,ni", [(13, 5), (12, 20), (9, 24), (6, 3)])
def test_trips_launched_ahead_of_their_challenges_give_the_same_rounds(ell, ni):
    """Round 6: while one two-rounds trip of prove_cubic_batched runs, the kernel of the next is already enqueued and waits for its challenges on
    a bell in host memory (internal.hpp AheadArm; option sumcheck.launch_ahead). An uninterrupted chain of trips — nothing else touches the context
    between them, as in the prover — must return the same sums, coefficients, final claims and tables whether each kernel is launched with its
    challenges (0), launched ahead and rung (1), or launched ahead and never rung (2: the test hook — every such kernel gives up after its 20 ms
    and the trip is repeated the ordinary way). ell = 13 / 12 with 20 instances: the grids whose partial sums need a second kernel."""
    import time
    from spartan_amd import capi
    n = 1 << ell
    rng = random.Random(77000 + ell)
    A0 = [fast_scalars(rng, n) for _ in range(ni)]; B0 = [fast_scalars(rng, n) for _ in range(ni)]
    C0 = [fast_scalars(rng, n) for _ in range(2)]
    w = fast_scalars(rng, ni)
    chal = [(rng.getrandbits(251), rng.getrandbits(250)) for _ in range(ell)]
    results = {}
    for mode in (0, 1, 2):
        ctx = capi.Ctx(0)
        ctx.set_option("testing.unlock", 1); ctx.set_option("sumcheck.launch_ahead", mode)
        tA, tB, tC = [up(ctx, a) for a in A0], [up(ctx, b) for b in B0], [up(ctx, c) for c in C0]
        hA = (vp * ni)(*[t.h for t in tA]); hB = (vp * ni)(*[t.h for t in tB])
        hC = (vp * ni)(*[tC[0].h if k < ni - 1 else tC[1].h for k in range(ni)])   # all but the last instance share their C
        ev = (ctypes.c_uint64 * 12)(); co = (ctypes.c_uint64 * 48)(); heads = (ctypes.c_uint64 * (4 * (2 * ni + 2)))()
        out = []
        assert capi.lib.sp_sumcheck_eval_coeffs_batched(ctx.h, hA, hB, hC, sz(ni), mont_bulk(w), ev, co) == 0
        out.append((bytes(ev), bytes(co)))
        length, k = n, 0
        t0 = time.time()
        while length >= 4:
            r0, r1 = chal[k]; k += 1
            length //= 4
            assert capi.lib.sp_sumcheck_bind2_eval_batched(ctx.h, hA, hB, hC, sz(ni), fq1(r0), fq1(r1), mont_bulk(w), ev if length >= 2 else None,
                                                           co if length >= 4 else None, heads if length == 1 else None) == 0
            out.append((bytes(ev) if length >= 2 else b"", bytes(co) if length >= 4 else b"", bytes(heads) if length == 1 else b""))
        dt = time.time() - t0
        if length >= 2:  # an odd number of variables: the last round binds once
            assert capi.lib.sp_sumcheck_bind2_eval_batched(ctx.h, hA, hB, hC, sz(ni), fq1(chal[k][0]), None, mont_bulk(w), None, None, heads) == 0
            out.append(bytes(heads))
        out.append(bytes(tA[0].download(1)) + bytes(tB[ni - 1].download(1)) + bytes(tC[0].download(1)) + bytes(tC[1].download(1)))
        results[mode] = out
        if mode == 1:
            assert dt < 0.018, "a trip launched ahead waited for its 20 ms time-out: %.1f ms for %d trips" % (dt * 1e3, k)
        for t in tA + tB + tC:
            t.free()
        ctx.close()
    assert results[1] == results[0] and results[2] == results[0]


@pytest.mark.parametrize("ell,nbind", [(5, 2), (4, 2), (3, 2), (4, 1), (3, 1), (2, 1), (3, 0), (2, 0), (1, 0), (6, 2)])
def test_short_tables_are_handed_over_with_the_round(ctx, ell, nbind):
    """sp_sumcheck_bind2_eval_tables_batched: the call of the two-rounds-per-trip path that also returns the tables once they have
    at most 8 entries, so that the host driver finishes the last <= 3 rounds of prove_cubic_batched on its own core
    (spark.inc, finish_on_host). The tables handed over are the bound tables of the reference arithmetic; with longer tables
    (ell = 6 -> 16 entries) nothing is handed over and the first word says so."""
    from spartan_amd import capi
    n = 1 << ell
    rng = random.Random(9900 + 10 * ell + nbind)
    npar, nseq = 3, 2
    ni = npar + nseq
    A = [fast_scalars(rng, n) for _ in range(ni)]
    B = [fast_scalars(rng, n) for _ in range(ni)]
    Cpar = fast_scalars(rng, n)
    Cseq = [fast_scalars(rng, n) for _ in range(nseq)]
    tA, tB = [up(ctx, a) for a in A], [up(ctx, b) for b in B]
    tCpar, tCseq = up(ctx, Cpar), [up(ctx, c) for c in Cseq]
    hA = (vp * ni)(*[t.h for t in tA]); hB = (vp * ni)(*[t.h for t in tB])
    hC = (vp * ni)(*([tCpar.h] * npar + [t.h for t in tCseq]))
    w = fast_scalars(rng, ni)
    rs = [rng.getrandbits(251) for _ in range(nbind)]
    for r in rs:
        A = [bind(a, r) for a in A]; B = [bind(b, r) for b in B]; Cpar = bind(Cpar, r); Cseq = [bind(c, r) for c in Cseq]
    m = n >> nbind
    Cs = [Cpar] * npar + Cseq
    ev = (ctypes.c_uint64 * 12)(); co = (ctypes.c_uint64 * 48)(); heads = (ctypes.c_uint64 * (4 * (2 * ni + 1 + nseq)))()
    tables = (ctypes.c_uint64 * (4 * ni * 3 * 8))()
    rc = capi.lib.sp_sumcheck_bind2_eval_tables_batched(ctx.h, hA, hB, hC, sz(ni), fq1(rs[0]) if nbind >= 1 else None, fq1(rs[1]) if nbind == 2 else None,
                                                        mont_bulk(w), ev if m >= 2 else None, co if m >= 4 else None, heads if m == 1 else None, tables)
    assert rc == 0
    if m >= 2:   # the weighted sums of the round, as without the hand-over
        want = [sum(w[k] * cubic_evals(A[k], B[k], Cs[k])[t] for k in range(ni)) % Q for t in range(3)]
        assert from_mont_bulk(ev, 3) == want
    if 2 <= m <= 8:
        got = from_mont_bulk(tables, ni * 3 * m)
        for k in range(ni):
            assert got[(3 * k) * m:(3 * k + 1) * m] == A[k] and got[(3 * k + 1) * m:(3 * k + 2) * m] == B[k] and got[(3 * k + 2) * m:(3 * k + 3) * m] == Cs[k], k
    else:
        assert tables[0] == 0xFFFFFFFFFFFFFFFF
    for t in tA + tB + [tCpar] + tCseq:
        t.free()
    for t in tA + tB + tCseq + [tCpar]:
        t.free()


def test_commit_rows_upload_start_equals_commit_of_the_uploaded_table(ctx):
    """sp_commit_rows_upload_start (the host assignment of SNARK::prove: row chunks committed behind their PCIe copies, one
    reduction at the end) returns the commitments sp_commit_rows_dev returns for the same rows already in HBM, with and without
    blinds, and leaves the rows in the destination table; 1024 x 256 takes the chunked path, 40 x 256 the copy-then-commit path."""
    import hashlib
    import numpy as np
    from spartan_amd import capi
    cols = 256
    g = capi.Gens(ctx, uniform=hashlib.shake_256(b"upload_commit").digest(64 * (cols + 1)))
    rng = np.random.default_rng(5)
    for rows, with_blinds in ((1024, True), (1024, False), (40, True)):
        Z = rng.integers(0, 2**64, size=(rows * cols, 4), dtype=np.uint64); Z[:, 3] &= np.uint64((1 << 60) - 1)
        Z[7 * cols:8 * cols] = 0                                                    # an all-zero row
        B = rng.integers(0, 2**64, size=(rows, 4), dtype=np.uint64); B[:, 3] &= np.uint64((1 << 60) - 1)
        zp = Z.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)); bp = B.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)) if with_blinds else None
        t_ref = capi.Table.upload(ctx, zp, rows * cols)
        want = (ctypes.c_uint8 * (32 * rows))()
        assert capi.lib.sp_commit_rows_dev(ctx.h, g.h, sz(0), sz(cols), t_ref.h, sz(0), sz(rows), sz(cols), bp, want) == 0
        dst = capi.Table.alloc(ctx, rows * cols)
        job = vp()
        assert capi.lib.sp_commit_rows_upload_start(ctx.h, g.h, sz(0), sz(cols), dst.h, sz(0), zp, sz(rows), sz(cols), bp, ctypes.byref(job)) == 0
        got = (ctypes.c_uint8 * (32 * rows))()
        assert capi.lib.sp_job_wait(job, got) == 0
        assert bytes(got) == bytes(want), (rows, with_blinds)
        assert bytes(dst.download()) == Z.tobytes()
        t_ref.free(); dst.free()
    g.free()
