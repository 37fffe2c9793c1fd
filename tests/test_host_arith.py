"""The product's field/curve/MSM source (spartan_amd/csrc/*.hpp) compiled for the host and checked against
Python big ints and the oracle. This validates the arithmetic that the gfx950 kernels execute without a GPU."""
import ctypes, random
import pytest
from tests.helpers import *

u8x32 = ctypes.c_uint8 * 32


def test_fq_matches_python(hc):
    rng = random.Random(11)
    out = u64x4()
    vals = [0, 1, Q - 1, Q - 2, 2, 2**252, (Q - 1) // 2] + [rng.randrange(Q) for _ in range(300)]
    for i in range(len(vals) - 1):
        a, b = vals[i], vals[i + 1]
        hc.hc_fq_mul(to_mont_limbs(a), to_mont_limbs(b), out); assert from_mont_limbs(out) == a * b % Q
        hc.hc_fq_add(to_mont_limbs(a), to_mont_limbs(b), out); assert from_mont_limbs(out) == (a + b) % Q
        hc.hc_fq_sub(to_mont_limbs(a), to_mont_limbs(b), out); assert from_mont_limbs(out) == (a - b) % Q
        hc.hc_fq_neg(to_mont_limbs(a), out); assert from_mont_limbs(out) == (-a) % Q
        hc.hc_fq_from_mont(to_mont_limbs(a), out); assert sum(int(out[k]) << (64 * k) for k in range(4)) == a
    for a in vals[1:30]:
        hc.hc_fq_invert(to_mont_limbs(a), out); assert from_mont_limbs(out) == pow(a, Q - 2, Q)
    w = (ctypes.c_uint64 * 8)(*[2**64 - 1] * 8)
    hc.hc_fq_from_u512(w, out); assert from_mont_limbs(out) == (2**512 - 1) % Q
    hc.hc_fq_from_u64(ctypes.c_uint64(12345), out); assert from_mont_limbs(out) == 12345


def test_fq_limbs_identical_to_oracle(hc, orc):
    rng = random.Random(12)
    o1 = u64x4(); o2 = u64x4()
    for _ in range(200):
        a, b = to_mont_limbs(rng.randrange(Q)), to_mont_limbs(rng.randrange(Q))
        hc.hc_fq_mul(a, b, o1); orc.orc_fq_mul(a, b, o2); assert list(o1) == list(o2)
        hc.hc_fq_sub(a, b, o1); orc.orc_fq_sub(a, b, o2); assert list(o1) == list(o2)


def test_fp_matches_python(hc):
    rng = random.Random(13)
    out = u8x32()
    edge = [0, 1, 19, P - 1, P, P + 1, 2**255 - 1, 2**255, 2**256 - 1, 2**256 - 38, 2**256 - 39, 38, 37]
    vals = edge + [rng.randrange(2**256) for _ in range(300)]

    def raw(x): return u64x4(*[(x >> (64 * i)) & (2**64 - 1) for i in range(4)])
    for i in range(len(vals)):
        for j in (i, (i + 1) % len(vals), (i * 7 + 3) % len(vals)):
            a, b = vals[i], vals[j]
            hc.hc_fp_mul_raw(raw(a), raw(b), out); assert int.from_bytes(bytes(out), "little") == a * b % P
            hc.hc_fp_sqr_raw(raw(a), out); assert int.from_bytes(bytes(out), "little") == a * a % P
            hc.hc_fp_add_raw(raw(a), raw(b), out); assert int.from_bytes(bytes(out), "little") == (a + b) % P
            hc.hc_fp_sub_raw(raw(a), raw(b), out); assert int.from_bytes(bytes(out), "little") == (a - b) % P
    for a in vals[1:40]:
        a %= 2**255
        if a % P == 0:
            continue
        hc.hc_fp_invert(a.to_bytes(32, "little"), out); assert int.from_bytes(bytes(out), "little") == pow(a, P - 2, P)


def test_points_match_oracle(hc, orc):
    rng = random.Random(14)
    o1 = u8x32(); o2 = u8x32()
    pts = []
    for _ in range(40):
        u = bytes(rng.randrange(256) for _ in range(64))
        hc.hc_pt_from_uniform(u, o1); orc.orc_pt_from_uniform_bytes(u, o2)
        assert bytes(o1) == bytes(o2)
        pts.append(bytes(o1))
    pts.append(bytes(32))  # identity
    for i in range(len(pts)):
        a, b = pts[i], pts[(i * 5 + 1) % len(pts)]
        assert hc.hc_pt_recompress(a, o1) == 1 and bytes(o1) == a
        assert hc.hc_pt_add(a, b, o1) == 1 and orc.orc_pt_add(a, b, o2) == 1 and bytes(o1) == bytes(o2)
        assert hc.hc_pt_add(a, a, o1) == 1 and hc.hc_pt_dbl(a, o2) == 1 and bytes(o1) == bytes(o2)
        orc.orc_pt_dbl(a, o1); assert bytes(o1) == bytes(o2)
    from tests.test_oracle_pins import RFC_BAD
    for bad in RFC_BAD:
        assert hc.hc_pt_recompress(bytes.fromhex(bad), o1) == 0


@pytest.mark.parametrize("kind", ["uniform", "edge", "small", "sparse"])
def test_fixed_base_table_msm_matches_oracle(hc, orc, kind):
    rng = random.Random(15)
    n = 6
    g = gens_bytes(orc, n - 1)  # n points
    sc = rand_scalars(rng, n, kind)
    o1 = u8x32(); o2 = u8x32()
    assert hc.hc_msm_fixed(g, sz(n), mont_array(sc), o1) == 1
    assert orc.orc_pt_msm(mont_array(sc), g, sz(n), o2) == 1
    assert bytes(o1) == bytes(o2)


@pytest.mark.parametrize("nwin,wbits", [(32, 0), (26, 0), (43, 0), (0, 6), (0, 7)])
def test_mixed_width_window_geometry_matches_oracle(hc, orc, nwin, wbits):
    """The window geometries of round 6 (spartan_amd/csrc/msm.hpp: nwin windows over exactly 254 bits, the top ones one bit wider; the
    policy uses 17 .. 32, here the cheap-to-build end and one past it) on the host build of the device arithmetic: tables laid out by
    msm_tidx / msm_woff, the strip form's shifting digit stream with one and with two entries in flight, and the per-window msm_digit
    lookups of the latency kernels all give the oracle's multi-scalar multiplication — edge scalars (0, 1, q - 1, all-ones runs that carry
    through every window, values just below 2^252 that fill the top window) included."""
    rng = random.Random(90 + nwin + wbits)
    n = 6
    g = gens_bytes(orc, n - 1)
    for kind in ("uniform", "edge", "sparse"):
        sc = rand_scalars(rng, n, kind)
        if kind == "edge":
            sc = [0, 1, Q - 1, (1 << 252) - 1, (1 << 252) + 27742317777372353535851937790883648492, (1 << 200) - 1][:n]
        o1, o2, o3, want = u8x32(), u8x32(), u8x32(), u8x32()
        assert hc.hc_msm_fixed_geom(g, sz(n), mont_array(sc), ctypes.c_int(nwin), ctypes.c_int(wbits), o1, o2, o3) == 1
        assert orc.orc_pt_msm(mont_array(sc), g, sz(n), want) == 1
        assert bytes(o1) == bytes(o2) == bytes(o3) == bytes(want), (nwin, wbits, kind)


def test_fe10_serial_chain_matches_python(hc):
    """radix-2^25.5 arithmetic used by the lone-wave exponentiation ladder (spartan_amd/csrc/fe10.hpp)."""
    rng = random.Random(16)
    out = u8x32()
    def raw(x): return u64x4(*[(x >> (64 * i)) & (2**64 - 1) for i in range(4)])
    edge = [0, 1, 2, 19, P - 1, P - 2, P, P + 5, 2**255 - 1, 2**255, 2**256 - 1, 2**26 - 1, 2**26, (2**255 - 19) // 2]
    vals = edge + [rng.randrange(2**256) for _ in range(200)]
    for i, a in enumerate(vals):
        b = vals[(i * 7 + 1) % len(vals)]
        hc.hc_fe10_mul_raw(raw(a), raw(b), out); assert int.from_bytes(bytes(out), "little") == a * b % P
        for k in (1, 2, 5, 50):
            hc.hc_fe10_sqr_chain_raw(raw(a), ctypes.c_int(k), out); assert int.from_bytes(bytes(out), "little") == pow(a, 2**k, P)
        hc.hc_fp_pow_p58_serial(raw(a), out); assert int.from_bytes(bytes(out), "little") == pow(a, (P - 5) // 8, P)
        hc.hc_fp_invert_serial(raw(a), out); assert int.from_bytes(bytes(out), "little") == pow(a, P - 2, P)


def test_pt10_add_and_encode_match_oracle(hc, orc):
    rng = random.Random(17)
    o1 = u8x32(); o2 = u8x32()
    pts = []
    for _ in range(30):
        u = bytes(rng.randrange(256) for _ in range(64))
        orc.orc_pt_from_uniform_bytes(u, o2); pts.append(bytes(o2))
    pts.append(bytes(32))  # identity
    for i in range(len(pts)):
        a, b = pts[i], pts[(i * 3 + 1) % len(pts)]
        assert hc.hc_pt10_add_compress(a, b, o1) == 1 and orc.orc_pt_add(a, b, o2) == 1 and bytes(o1) == bytes(o2)
        assert hc.hc_pt10_add_compress(a, a, o1) == 1 and orc.orc_pt_dbl(a, o2) == 1 and bytes(o1) == bytes(o2)
    # long accumulation chain (limb bounds must hold across many dependent additions)
    acc = bytes(32)
    for p_ in pts:
        orc.orc_pt_add(acc, p_, o2); acc = bytes(o2)
    assert hc.hc_pt10_sum_compress(b"".join(pts), sz(len(pts)), o1) == 1 and bytes(o1) == acc


def test_small_commitments_on_the_host_core_match_oracle(orc):
    """sp_host_commit_* (csrc/host_commit.hip): the few-term commitments of the Sigma protocols, computed on the calling
    thread's core from signed 10-bit window tables — same recoding, same point arithmetic as the device tables (no device
    needed: this is the device-free entry point of the same code)"""
    import ctypes, random
    from spartan_amd import capi
    rng = random.Random(2024)
    g = gens_bytes(orc, 4)  # 5 points
    for rows, kind in [(3, "uniform"), (2, "edge"), (4, "sparse"), (2, "small")]:
        S = rand_scalars(rng, rows * 5, kind)
        got = (ctypes.c_uint8 * (32 * rows))()
        assert capi.lib.sp_host_commit_probe(g, sz(5), mont_array(S), sz(rows), got) == 0
        want = (ctypes.c_uint8 * 32)()
        for r in range(rows):
            assert orc.orc_pt_msm(mont_array(S[r * 5:(r + 1) * 5]), g, sz(5), want) == 1
            assert bytes(got[32 * r:32 * r + 32]) == bytes(want), (kind, r)


def _cubic_evals(A, B, C):
    h = len(A) // 2
    out = []
    for t in (0, 2, 3):
        acc = 0
        for z in range(h):
            a = (A[z] + t * (A[h + z] - A[z])) % Q; b = (B[z] + t * (B[h + z] - B[z])) % Q; c = (C[z] + t * (C[h + z] - C[z])) % Q
            acc += a * b * c
        out.append(acc % Q)
    return out


def _bind(T, r):
    h = len(T) // 2
    return [(T[z] + r * (T[h + z] - T[z])) % Q for z in range(h)]


def test_last_rounds_of_the_batched_sumcheck_on_the_host():
    """spark.inc cubic_tail_rounds: the <= 3 last rounds of prove_cubic_batched (sumcheck.rs:287-419) that the driver runs on its
    own core once the device has handed over the short tables — evaluations at t = 0, 2, 3 combined with the batching
    coefficients, binds, final values — against the reference arithmetic, for every hand-over length."""
    import ctypes, random
    from spartan_amd import prover
    rng = random.Random(4242)
    for m in (8, 4, 2):
        ni = 5
        tabs = [[[rng.randrange(Q) for _ in range(m)] for _ in range(3)] for _ in range(ni)]
        coeffs = [rng.randrange(Q) for _ in range(ni)]
        rounds = m.bit_length() - 1
        ch = [rng.randrange(Q) for _ in range(rounds)]
        flat = [x for inst in tabs for t in inst for x in t]
        buf = mont_array(flat); evs = (ctypes.c_uint64 * (4 * 3 * rounds))()
        assert prover.H.spz_cubic_tail_probe(buf, sz(ni), sz(m), mont_array(coeffs), mont_array(ch), evs) == rounds
        got_ev = from_mont_array(evs, 3 * rounds)
        cur = tabs
        for k in range(rounds):
            want = [sum(coeffs[i] * _cubic_evals(*cur[i])[t] for i in range(ni)) % Q for t in range(3)]
            assert got_ev[3 * k:3 * k + 3] == want, (m, k)
            cur = [[_bind(T, ch[k]) for T in inst] for inst in cur]
        out = from_mont_array(buf, ni * 3 * m)
        for i in range(ni):
            for t in range(3):
                assert out[(i * 3 + t) * m] == cur[i][t][0], (m, i, t)


def test_cubic_in_the_next_challenge_on_the_host():
    """spark.inc evals_from_coeffs: E(t; r) = (1-r)^3 M0 + (1-r)^2 r M1 + (1-r) r^2 M2 + r^3 M3 with M1 = (T1-T2)/2 - M3,
    M2 = (T1+T2)/2 - M0 is the evaluation of the round after a bind at r (two rounds per trip): checked by building
    M0, M3, T1, T2 from random tables in Python and comparing with the evaluations of the tables actually bound at r."""
    import ctypes, random
    from spartan_amd import prover
    rng = random.Random(77)
    n = 16
    A, B, C = ([rng.randrange(Q) for _ in range(n)] for _ in range(3))
    q = n // 4
    line = lambda u, v, t: (u + t * (v - u)) % Q
    S = []
    for t in (0, 2, 3):
        M0 = M3 = T1 = T2 = 0
        for i in range(q):
            P = [line(T[i], T[i + q], t) for T in (A, B, C)]            # entries (i, i+q) of the low half: the pair the bind mixes with ...
            U = [line(T[i + 2 * q], T[i + 3 * q], t) for T in (A, B, C)]  # ... the pair of the high half
            M0 += P[0] * P[1] * P[2]; M3 += U[0] * U[1] * U[2]
            T1 += (P[0] + U[0]) * (P[1] + U[1]) * (P[2] + U[2]); T2 += (P[0] - U[0]) * (P[1] - U[1]) * (P[2] - U[2])
        S += [M0 % Q, M3 % Q, T1 % Q, T2 % Q]
    r = rng.randrange(Q)
    ev = (ctypes.c_uint64 * 12)()
    prover.H.spz_cubic_coeffs_probe(mont_array(S), mont_array([r]), ev)
    assert from_mont_array(ev, 3) == _cubic_evals(_bind(A, r), _bind(B, r), _bind(C, r))



def test_challenge_inversion_by_division_steps():
    """fq_inv.hpp: the inner-product rounds invert their (public) challenge with Bernstein-Yang division steps instead of the
    reference's a^(q-2) chain (scalar/ristretto255.rs:541-595). Same value for every input: edge cases, powers of two (long runs of
    zero bits exercise the skip-ahead), short and full-length scalars against Python's pow, and 0 -> 0 as the chain gives."""
    import ctypes, random
    from spartan_amd import prover
    rng = random.Random(2024)
    out = u64x4()
    vals = [0, 1, 2, 3, Q - 1, Q - 2, (Q - 1) // 2, (Q + 1) // 2, 2**252, 2**252 - 1, 2**126, 2**62, 2**62 - 1, 2**124 + 1]
    vals += [2**k for k in range(0, 252, 7)] + [Q - 2**k for k in range(1, 252, 11)]
    vals += [rng.randrange(Q) for _ in range(3000)] + [rng.randrange(2**k) for k in (8, 31, 62, 63, 64, 65, 124, 130, 200) for _ in range(30)]
    for a in vals:
        prover.H.spz_fq_invert_vartime(to_mont_limbs(a), out)
        assert from_mont_limbs(out) == pow(a, Q - 2, Q), hex(a)


def test_several_encodes_at_once_equal_single_encodes(hc, orc):
    """curve.hpp pt_compress_many: the proving thread encodes L and R of an inner-product round, the rows of a few-term commitment
    call and the <= 8 row sums of a small commitment several at a time (interleaved inverse-square-root chains). Every group size
    must give the bytes of pt_compress, which test_points_match_oracle ties to the oracle; the running sums have Z != 1 and include
    the identity and points whose encoding takes the rotated branch."""
    rng = random.Random(15)
    o = u8x32()
    pts = []
    for _ in range(23):
        hc.hc_pt_from_uniform(bytes(rng.randrange(256) for _ in range(64)), o)
        pts.append(bytes(o))
    pts.insert(3, bytes(32))  # identity as an addend: two equal running sums
    for n in list(range(1, 12)) + [24]:
        blob = b"".join(pts[:n])
        many = (ctypes.c_uint8 * (32 * n))(); single = (ctypes.c_uint8 * (32 * n))()
        assert hc.hc_pt_running_sums_compress_many(blob, sz(n), many, single) == 1
        assert bytes(many) == bytes(single), n
    acc = pts[0]
    o2 = u8x32()
    for i in range(1, 6):  # and the running sums themselves against the oracle's additions
        assert orc.orc_pt_add(acc, pts[i], o2) == 1
        acc = bytes(o2)
        assert bytes(many)[32 * i:32 * i + 32] == acc


def test_eq_table_as_a_factor_host_arithmetic():
    """spark.inc EqFactor: in the throughput-sized rounds of prove_cubic_batched (sumcheck.rs:287-393) under
    ProductCircuitEvalProofBatched::prove the device returns, per product-circuit instance, q(0) and q(2) of the QUADRATIC
    q(t) = sum_x A(t,x) B(t,x) C_original[x] (the eq table never bound) and four evaluations of each generic instance; the host
    rebuilds the round's E(0), E(2), E(3) from them, the round's claim and the eq point. Here the "device" is Python: every round of
    a whole sum-check on random tables is computed both ways — the generic sums over tables actually bound, as the reference does,
    and the factored sums pushed through the host arithmetic — and must agree; the hand-over scalar times the original table must
    be the bound eq table."""
    import ctypes, random
    from spartan_amd import prover
    rng = random.Random(99)
    nv, np_, ns = 5, 3, 2
    ni = np_ + ns
    n = 1 << nv
    rho = [rng.randrange(2, Q) for _ in range(nv)]
    C = [1]
    for r in rho:  # EqPolynomial::evals (dense_mlpoly.rs:68-84): r[0] is the most significant index bit
        C = [x * y % Q for x in C for y in ((1 - r) % Q, r)]
    A = [[rng.randrange(Q) for _ in range(n)] for _ in range(ni)]
    B = [[rng.randrange(Q) for _ in range(n)] for _ in range(ni)]
    Cg = [[rng.randrange(Q) for _ in range(n)] for _ in range(ns)]  # the generic instances' own third tables
    coeffs = [rng.randrange(Q) for _ in range(ni)]
    line = lambda u, v, t: (u + t * (v - u)) % Q
    def generic_evals(a, b, c):  # sumcheck.rs:290-357 for one instance: t = 0, 1, 2, 3
        h = len(a) // 2
        return [sum(line(a[z], a[h + z], t) * line(b[z], b[h + z], t) * line(c[z], c[h + z], t) for z in range(h)) % Q for t in range(4)]
    bind = lambda T, r: [(T[z] + r * (T[len(T) // 2 + z] - T[z])) % Q for z in range(len(T) // 2)]
    Cb = list(C)
    claim = sum(coeffs[i] * sum(A[i][x] * B[i][x] * (C[x] if i < np_ else Cg[i - np_][x]) for x in range(n)) for i in range(ni)) % Q
    claims, ev4, chal, want = [], [], [], []
    for j in range(nv):
        h = len(A[0]) // 2
        per = []
        E = [0, 0, 0, 0]
        for i in range(ni):
            g = generic_evals(A[i], B[i], Cb if i < np_ else Cg[i - np_])
            for t in range(4):
                E[t] = (E[t] + coeffs[i] * g[t]) % Q
            if i < np_:  # what the factored kernel returns: sums against the ORIGINAL eq table's leading entries
                q = [sum(line(A[i][z], A[i][h + z], t) * line(B[i][z], B[i][h + z], t) * C[z] for z in range(h)) % Q for t in (0, 2)]
                per += [q[0], q[1], 0, 0]
            else:
                per += g
        assert (E[0] + E[1]) % Q == claim
        claims.append(claim); ev4 += per; want += [E[0], E[2], E[3]]
        r = rng.randrange(Q)
        chal.append(r)
        # the next claim: the round's cubic at r (UniPoly::evaluate): Lagrange over t = 0, 1, 2, 3
        inv = lambda x: pow(x, Q - 2, Q)
        L = [(-(r - 1) * (r - 2) * (r - 3) * inv(6)), (r * (r - 2) * (r - 3) * inv(2)), (-r * (r - 1) * (r - 3) * inv(2)), (r * (r - 1) * (r - 2) * inv(6))]
        claim = sum(L[t] * E[t] for t in range(4)) % Q
        A = [bind(T, r) for T in A]; B = [bind(T, r) for T in B]; Cg = [bind(T, r) for T in Cg]; Cb = bind(Cb, r)
        want_K = None
    evc = (ctypes.c_uint64 * (4 * 3 * nv))(); Ks = (ctypes.c_uint64 * (4 * nv))()
    assert prover.H.spz_eq_factor_probe(mont_array(rho), sz(nv), sz(np_), sz(ni), mont_array(coeffs), sz(nv), mont_array(claims), mont_array(ev4),
                                        mont_array(chal), evc, Ks) == 1
    assert from_mont_array(evc, 3 * nv) == want
    # hand-over: after j binds, K_j * C_original[x] = the eq table bound at the first j challenges, for every x below the bound length
    K = from_mont_array(Ks, nv)
    Cb = list(C)
    for j in range(nv):
        Cb = bind(Cb, chal[j])
        assert all(K[j] * C[x] % Q == Cb[x] for x in range(len(Cb))), j
    # a coordinate equal to 0 or 1 has no factored form: the driver must fall back to the generic kernels
    assert prover.H.spz_eq_factor_probe(mont_array([1] + rho[1:]), sz(nv), sz(np_), sz(ni), mont_array(coeffs), sz(nv), mont_array(claims), mont_array(ev4),
                                        mont_array(chal), evc, Ks) == 0


def test_two_point_variable_base_multiplication_on_the_host(orc):
    """ipa.hip pt_var_msm2 (through its device-free entry point): the end of every inner-product argument is k1 P1 + k2 P2 over the last
    round's two row sums, computed by the proving thread with 4-bit windows (sp_ipa_finish_commit). Against the oracle's multi-scalar
    multiplication for random, small, zero and maximal scalars and for equal and opposite points."""
    import ctypes, random
    from spartan_amd import capi
    rng = random.Random(808)
    g = gens_bytes(orc, 5)
    P = [g[32 * i:32 * i + 32] for i in range(6)]
    neg = (ctypes.c_uint8 * 32)()
    assert orc.orc_pt_msm(mont_array([Q - 1]), P[0], sz(1), neg) == 1   # -P0
    got = (ctypes.c_uint8 * 32)(); want = (ctypes.c_uint8 * 32)()
    cases = [(P[0], rng.randrange(Q), P[1], rng.randrange(Q)) for _ in range(20)]
    cases += [(P[2], 0, P[3], rng.randrange(Q)), (P[2], rng.randrange(Q), P[3], 0), (P[2], 0, P[3], 0), (P[4], 1, P[5], Q - 1), (P[4], Q - 1, P[5], Q - 1),
              (P[0], 7, P[0], 9), (P[0], 5, bytes(neg), 5), (P[1], 2**252, P[2], 15), (P[1], 16, P[2], 2**248 + 1), (bytes(32), rng.randrange(Q), P[3], 3)]
    for p1, k1, p2, k2 in cases:
        assert capi.lib.sp_host_msm2_probe(p1, mont_array([k1]), p2, mont_array([k2]), got) == 0
        assert orc.orc_pt_msm(mont_array([k1, k2]), p1 + p2, sz(2), want) == 1
        assert bytes(got) == bytes(want), (k1, k2)
    assert capi.lib.sp_host_msm2_probe(bytes([1] + [0] * 31), mont_array([1]), P[0], mont_array([1]), got) != 0   # not a valid encoding
