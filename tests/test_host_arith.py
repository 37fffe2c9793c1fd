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
    """small_msm.cc: the few-term commitments of the Sigma protocols, computed on the proving thread's core from signed
    10-bit window tables — same recoding, same point arithmetic as the device tables (no device needed)"""
    import ctypes, random
    from spartan_amd import prover
    rng = random.Random(2024)
    g = gens_bytes(orc, 4)  # 5 points
    for rows, kind in [(3, "uniform"), (2, "edge"), (4, "sparse"), (2, "small")]:
        S = rand_scalars(rng, rows * 5, kind)
        got = (ctypes.c_uint8 * (32 * rows))()
        assert prover.H.spz_small_msm_probe(g, sz(5), mont_array(S), sz(rows), got) == 0
        want = (ctypes.c_uint8 * 32)()
        for r in range(rows):
            assert orc.orc_pt_msm(mont_array(S[r * 5:(r + 1) * 5]), g, sz(5), want) == 1
            assert bytes(got[32 * r:32 * r + 32]) == bytes(want), (kind, r)
