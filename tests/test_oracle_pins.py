"""Pins the ORACLE (oracle/) before anything is compared against it.
 - F_q: the known answers of the reference's own tests (src/scalar/ristretto255.rs:772-1202) + Python ints
 - group: RFC 9496 vectors and libsodium 1.0.18 (an independent implementation) when available
 - transcript: Merlin test vector, SHAKE256 vs hashlib
 - wire format: proof lengths derived from the serde structs (README.md:362 gives 47 024 B at s=20)
 - protocol: the restated verifier accepts the oracle's proofs and rejects tampered ones
"""
import ctypes, hashlib, os, random
import pytest
from tests.helpers import *

R2_BYTES = bytes([29, 149, 152, 141, 116, 49, 236, 214, 112, 207, 125, 115, 244, 91, 239, 198, 254] + [255] * 14 + [15])
NEG1_BYTES = bytes([236, 211, 245, 92, 26, 99, 18, 88, 214, 156, 247, 162, 222, 249, 222, 20] + [0] * 15 + [16])
R_LIMBS = [0xd6ec31748d98951d, 0xc6ef5bf4737dcf70, 0xfffffffffffffffe, 0x0fffffffffffffff]
R2_LIMBS = [0xa40611e3449c0f01, 0xd00e1ba768859347, 0xceec73d217f5be65, 0x0399411b7c309a3d]
R3_LIMBS = [0x2a9e49687b83a2db, 0x278324e6aef7f3ec, 0x8065dc6c04ec5b65, 0x0e530b773599cec7]


def tb(orc, limbs):
    out = (ctypes.c_uint8 * 32)()
    orc.orc_fq_to_bytes(u64x4(*limbs), out)
    return bytes(out)


def test_fq_constants_and_to_bytes(orc):
    assert sum(l << (64 * i) for i, l in enumerate(R_LIMBS)) == 2**256 % Q
    assert sum(l << (64 * i) for i, l in enumerate(R2_LIMBS)) == 2**512 % Q
    assert sum(l << (64 * i) for i, l in enumerate(R3_LIMBS)) == 2**768 % Q
    assert (0xd2b51da312547e1b * Q) % 2**64 == 2**64 - 1  # INV (ristretto255.rs:777-789)
    assert tb(orc, [0, 0, 0, 0]) == bytes(32)                     # :819-851
    assert tb(orc, R_LIMBS) == bytes([1] + [0] * 31)
    assert tb(orc, R2_LIMBS) == R2_BYTES
    neg1 = u64x4()
    orc.orc_fq_neg(u64x4(*R_LIMBS), neg1)
    assert tb(orc, list(neg1)) == NEG1_BYTES


def test_fq_from_bytes(orc):  # ristretto255.rs:854-932
    out = u64x4()
    assert orc.orc_fq_from_bytes(bytes(32), out) == 1 and list(out) == [0, 0, 0, 0]
    assert orc.orc_fq_from_bytes(bytes([1] + [0] * 31), out) == 1 and list(out) == R_LIMBS
    assert orc.orc_fq_from_bytes(R2_BYTES, out) == 1 and list(out) == R2_LIMBS
    assert orc.orc_fq_from_bytes(NEG1_BYTES, out) == 1
    for bad in ([1, 0, 0, 0, 255, 255, 255, 255, 254, 91, 254, 255, 2, 164, 189, 83, 5, 216, 161, 9, 8, 216, 57, 51, 72, 125, 157, 41, 83, 167, 237, 115],
                [2, 0, 0, 0, 255, 255, 255, 255, 254, 91, 254, 255, 2, 164, 189, 83, 5, 216, 161, 9, 8, 216, 57, 51, 72, 125, 157, 41, 83, 167, 237, 115]):
        assert orc.orc_fq_from_bytes(bytes(bad), out) == 0
    assert orc.orc_fq_from_bytes(Q.to_bytes(32, "little"), out) == 0
    assert orc.orc_fq_from_bytes((Q - 1).to_bytes(32, "little"), out) == 1
    # lib.rs:658-661 "larger_than_mod"
    assert orc.orc_fq_from_bytes(bytes([3, 0, 0, 0, 255, 255, 255, 255, 254, 91, 254, 255, 2, 164, 189, 83, 5, 216, 161, 9, 8, 216, 57, 51, 72, 125, 157, 41, 83, 167, 237, 115]), out) == 0


def test_fq_from_bytes_wide(orc):  # ristretto255.rs:935-1005
    out = u64x4()
    orc.orc_fq_from_bytes_wide(R2_BYTES + bytes(32), out)
    assert list(out) == R2_LIMBS
    orc.orc_fq_from_bytes_wide(NEG1_BYTES + bytes(32), out)
    assert tb(orc, list(out)) == NEG1_BYTES
    orc.orc_fq_from_bytes_wide(bytes([0xff] * 64), out)
    raw = sum(l << (64 * i) for i, l in enumerate([0xa40611e3449c0f00, 0xd00e1ba768859347, 0xceec73d217f5be65, 0x0399411b7c309a3d]))
    assert from_mont_limbs(out) == raw == (2**512 - 1) % Q  # Scalar::from_raw takes the plain integer (:1175-1190)
    orc.orc_fq_from_bytes_wide(Q.to_bytes(32, "little") + bytes(32), out)
    assert list(out) == [0, 0, 0, 0]
    rng = random.Random(1)
    for _ in range(50):
        b = bytes(rng.randrange(256) for _ in range(64))
        orc.orc_fq_from_bytes_wide(b, out)
        assert from_mont_limbs(out) == int.from_bytes(b, "little") % Q


def test_fq_arith_vs_python(orc):  # ristretto255.rs:1023-1172 (identities) against big ints
    rng = random.Random(2)
    out = u64x4()
    vals = [0, 1, Q - 1, Q - 2, 2, (Q - 1) // 2] + [rng.randrange(Q) for _ in range(200)]
    for i in range(0, len(vals) - 1):
        a, b = vals[i], vals[i + 1]
        orc.orc_fq_mul(to_mont_limbs(a), to_mont_limbs(b), out); assert from_mont_limbs(out) == a * b % Q
        orc.orc_fq_add(to_mont_limbs(a), to_mont_limbs(b), out); assert from_mont_limbs(out) == (a + b) % Q
        orc.orc_fq_sub(to_mont_limbs(a), to_mont_limbs(b), out); assert from_mont_limbs(out) == (a - b) % Q
        orc.orc_fq_neg(to_mont_limbs(a), out); assert from_mont_limbs(out) == (-a) % Q
    for a in vals[1:40]:
        orc.orc_fq_invert(to_mont_limbs(a), out); assert from_mont_limbs(out) == pow(a, Q - 2, Q)


# ---------------- group
BASEPOINT = "e2f2ae0a6abc4e71a884a961c500515f58e30b6aa582dd8db6a65945e08d2d76"
RFC_MULTIPLES = [  # RFC 9496 A.1: k*B for k = 0..15
    "0000000000000000000000000000000000000000000000000000000000000000",
    "e2f2ae0a6abc4e71a884a961c500515f58e30b6aa582dd8db6a65945e08d2d76",
    "6a493210f7499cd17fecb510ae0cea23a110e8d5b901f8acadd3095c73a3b919",
    "94741f5d5d52755ece4f23f044ee27d5d1ea1e2bd196b462166b16152a9d0259",
    "da80862773358b466ffadfe0b3293ab3d9fd53c5ea6c955358f568322daf6a57",
    "e882b131016b52c1d3337080187cf768423efccbb517bb495ab812c4160ff44e",
    "f64746d3c92b13050ed8d80236a7f0007c3b3f962f5ba793d19a601ebb1df403",
    "44f53520926ec81fbd5a387845beb7df85a96a24ece18738bdcfa6a7822a176d",
    "903293d8f2287ebe10e2374dc1a53e0bc887e592699f02d077d5263cdd55601c",
    "02622ace8f7303a31cafc63f8fc48fdc16e1c8c8d234b2f0d6685282a9076031",
    "20706fd788b2720a1ed2a5dad4952b01f413bcf0e7564de8cdc816689e2db95f",
    "bce83f8ba5dd2fa572864c24ba1810f9522bc6004afe95877ac73241cafdab42",
    "e4549ee16b9aa03099ca208c67adafcafa4c3f3e4e5303de6026e3ca8ff84460",
    "aa52e000df2e16f55fb1032fc33bc42742dad6bd5a8fc0be0167436c5948501f",
    "46376b80f409b29dc2b5f6f0c52591990896e5716f41477cd30085ab7f10301e",
    "e0c418f7c8d9c4cdd7395b93ea124f3ad99021bb681dfc3302a9d99a2e53e64e",
]
RFC_BAD = [  # RFC 9496 A.2 "Invalid Encodings", all five classes (curve25519-dalek's `bad_encodings` holds the same list).
    # Every entry is also required to be rejected by libsodium below, so a mistyped vector cannot pass unnoticed.
    # non-canonical field encodings
    "00ffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffff",
    "ffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffff7f",
    "f3ffffffffffffffffffffffffffffffffffffffffffffffffffffffffffff7f",
    "edffffffffffffffffffffffffffffffffffffffffffffffffffffffffffff7f",
    # negative field elements
    "0100000000000000000000000000000000000000000000000000000000000000",
    "01ffffffffffffffffffffffffffffffffffffffffffffffffffffffffffff7f",
    "ed57ffd8c914fb201471d1c3d245ce3c746fcbe63a3679d51b6a516ebebe0e20",
    "c34c4e1826e5d403b78e246e88aa051c36ccf0aafebffe137d148a2bf9104562",
    "c940e5a4404157cfb1628b108db051a8d439e1a421394ec4ebccb9ec92a8ac78",
    "47cfc5497c53dc8e61c91d17fd626ffb1c49e2bca94eed052281b510b1117a24",
    "f1c6165d33367351b0da8f6e4511010c68174a03b6581212c71c0e1d026c3c72",
    "87260f7a2f12495118360f02c26a470f450dadf34a413d21042b43b9d93e1309",
    # non-square x^2
    "26948d35ca62e643e26a83177332e6b6afeb9d08e4268b650f1f5bbd8d81d371",
    "4eac077a713c57b4f4397629a4145982c661f48044dd3f96427d40b147d9742f",
    "de6a7b00deadbeefdeadbeefdeadbeefdeadbeefdeadbeefdeadbeefdeadbeef",
    "bcab477be20861e01e4a0e295284146a510150d9817763caf1a6f4b422d67042",
    "2a292df7e32cababbd9de088d1d1abec9fc0440f637ed2fba145094dc14bea08",
    "f4a9e534fc0d216c44b218fa0c42d99635a0127ee2e53c712f70609649fdff22",
    "8268436f8c4126196cf64b3c7ddbda90746a378625f9813dd9b8457077256731",
    "2810e5cbc2cc4d4eece54f61c6f69758e289aa7ab440b3cbeaa21995c2f4232b",
    # negative x*y
    "3eb858e78f5a7254d8c9731174a94f76755fd3941c0ac93735c07ba14579630e",
    "a45fdc55c76448c049a1ab33f17023edfb2be3581e9c7aade8a6125215e04220",
    "d483fe813c6ba647ebbfd3ec41adca1c6130c2beeee9d9bf065c8d151c5f396e",
    "8a2e1d30050198c65a54483123960ccc38aef6848e1ec8f5f780e8523769ba32",
    "32888462f8b486c68ad7dd9610be5192bbeaf3b443951ac1a8118419d9fa097b",
    "227142501b9d4355ccba290404bde41575b037693cef1f438c47f8fbf35d1165",
    "5c37cc491da847cfeb9281d407efc41e15144c876e0170b499a96a22ed31e01e",
    "445425117cb8c90edcbc7c1cc0e74f747f2c1efa5630a967c64f287792a48a4b",
    # s = -1 (y = 0)
    "ecffffffffffffffffffffffffffffffffffffffffffffffffffffffffffff7f",
]
RFC_MAP_IN = "5d1be09e3d0c82fc538112490e35701979d99e06ca3e2b5b54bffe8b4dc772c14d98b696a1bbfb5ca32c436cc61c16563790306c79eaca7705668b47dffe5bb6"
RFC_MAP_OUT = "3066f82a1a747d45120d1740f14358531a8f04bbffe6a819f86dfe50f44a0a46"


def test_group_rfc9496_vectors(orc):
    out = (ctypes.c_uint8 * 32)()
    orc.orc_basepoint(out)
    assert bytes(out).hex() == BASEPOINT
    acc = bytes(32)
    B = bytes.fromhex(BASEPOINT)
    for k in range(16):
        assert acc.hex() == RFC_MULTIPLES[k]
        assert orc.orc_pt_recompress(acc, out) == 1 and bytes(out) == acc
        assert orc.orc_pt_mul_bytes(k.to_bytes(32, "little"), B, out) == 1 and bytes(out).hex() == RFC_MULTIPLES[k]
        assert orc.orc_pt_add(acc, B, out) == 1
        acc = bytes(out)
    sod = _sodium()
    for bad in RFC_BAD:
        assert orc.orc_pt_recompress(bytes.fromhex(bad), out) == 0, bad
        if sod is not None:
            assert sod.crypto_core_ristretto255_is_valid_point(bytes.fromhex(bad)) == 0, bad
    # one-way map: RFC 9496 A.3 first vector (input = SHA-512("Ristretto is traditionally a short shot of espresso coffee"))
    h = hashlib.sha512(b"Ristretto is traditionally a short shot of espresso coffee").digest()
    assert h.hex() == RFC_MAP_IN
    orc.orc_pt_from_uniform_bytes(h, out)
    assert bytes(out).hex() == RFC_MAP_OUT


def _sodium():
    for p in ("/opt/conda/lib/libsodium.so", "libsodium.so", "libsodium.so.23"):
        try:
            s = ctypes.CDLL(p)
            s.sodium_init()
            return s
        except OSError:
            continue
    return None


def test_group_vs_libsodium(orc):
    s = _sodium()
    if s is None:
        pytest.skip("libsodium not present on this machine")
    rng = random.Random(3)
    out = (ctypes.c_uint8 * 32)(); ref = (ctypes.c_uint8 * 32)()
    pts = []
    for _ in range(20):
        u = bytes(rng.randrange(256) for _ in range(64))
        orc.orc_pt_from_uniform_bytes(u, out)
        assert s.crypto_core_ristretto255_from_hash(ref, u) == 0
        assert bytes(out) == bytes(ref)
        pts.append(bytes(out))
    for i in range(len(pts) - 1):
        assert orc.orc_pt_add(pts[i], pts[i + 1], out) == 1
        assert s.crypto_core_ristretto255_add(ref, pts[i], pts[i + 1]) == 0
        assert bytes(out) == bytes(ref)
        k = rng.randrange(1, Q)
        assert orc.orc_pt_mul_bytes(k.to_bytes(32, "little"), pts[i], out) == 1
        assert s.crypto_scalarmult_ristretto255(ref, k.to_bytes(32, "little"), pts[i]) == 0
        assert bytes(out) == bytes(ref)
    # MSM (Straus and Pippenger branches) against sum of libsodium scalar mults
    for n in (3, 40, 200):
        P = [pts[i % len(pts)] for i in range(n)]
        sc = [rng.randrange(Q) for _ in range(n)]
        assert orc.orc_pt_msm(mont_array(sc), b"".join(P), sz(n), out) == 1
        acc = bytes(32)
        tmp = (ctypes.c_uint8 * 32)()
        for k, p in zip(sc, P):
            if k == 0:
                continue
            assert s.crypto_scalarmult_ristretto255(tmp, k.to_bytes(32, "little"), p) == 0
            s.crypto_core_ristretto255_add(ref, acc, bytes(tmp)); acc = bytes(ref)
        assert bytes(out) == acc


def test_generators_are_shake_stream(orc):
    # commitments.rs:15-33: SHAKE256(label || compressed basepoint), 64 bytes per point
    label = b"gens_r1cs_sat"
    g = gens_bytes(orc, 4, label)
    stream = hashlib.shake_256(label + bytes.fromhex(BASEPOINT)).digest(64 * 5)
    out = (ctypes.c_uint8 * 32)()
    for i in range(5):
        orc.orc_pt_from_uniform_bytes(stream[64 * i:64 * i + 64], out)
        assert bytes(out) == g[32 * i:32 * i + 32]
    # values recorded in SURVEY.md §8c (computed through libsodium)
    assert g[:32].hex() == "f8dad3b0fba18ec2a61684952cbfd51372cbdcca26b05e5b0b4637157c98ca43"
    assert g[32:64].hex() == "da819f7228eaa0de8b0112cc7520a7367292513556bd70d3f7b68cf86e962d23"


# ---------------- transcript
def test_shake256_vs_hashlib(orc):
    rng = random.Random(4)
    for n in (0, 1, 135, 136, 137, 500):
        m = bytes(rng.randrange(256) for _ in range(n))
        out = (ctypes.c_uint8 * 300)()
        orc.orc_shake256(m, sz(n), out, sz(300))
        assert bytes(out) == hashlib.shake_256(m).digest(300)


def test_merlin_kat(orc):
    # merlin crate tests::equivalence_simple
    out = (ctypes.c_uint8 * 32)()
    orc.orc_merlin_simple(b"test protocol", b"some label", b"some data", sz(9), b"challenge", out, sz(32))
    assert bytes(out).hex() == "d5a21972d0d5fe320c0d263fac7fffb8145aa640af6e9bca177c03c7efcf0615"


# ---------------- wire format + protocol self-check
def sat_proof_len(s):
    L = 1 << (s // 2); lgR = s - s // 2
    return (8 + L * 32) + (2 * (8 + s * 32) + 8 + s * 264) + 128 + 96 + 256 + 64 + (2 * (8 + (s + 1) * 32) + 8 + (s + 1) * 232) + 32 + \
        (2 * (8 + lgR * 32) + 128) + 64


def test_proof_length_formula_matches_readme():
    assert sat_proof_len(20) == 47024  # README.md:362


@pytest.mark.parametrize("s", [4, 7, 10])
def test_oracle_nizk_roundtrip_and_lengths(orc, s):
    N = 1 << s
    inst = vp(orc.orc_instance_synthetic(sz(N), sz(N), sz(10), ctypes.c_uint64(s)))
    assert orc.orc_instance_is_sat(inst) == 1
    g = vp(orc.orc_nizk_gens_new(sz(N), sz(N), sz(10)))
    seed = u64x4(); orc.orc_seed_scalar(b"tape", ctypes.c_uint64(0), seed)
    digest = b"opaque-digest"
    p = vp(orc.orc_nizk_prove(inst, g, digest, sz(len(digest)), b"nizk_example", seed, None))
    lens = (sz * 3)(); orc.orc_proof_part_lens(p, lens)
    assert lens[0] == sat_proof_len(s)
    assert orc.orc_nizk_verify(p, inst, g, digest, sz(len(digest)), b"nizk_example") == 1
    assert orc.orc_nizk_verify(p, inst, g, b"other", sz(5), b"nizk_example") == 0
    orc.orc_proof_tamper(p, 1)
    assert orc.orc_nizk_verify(p, inst, g, digest, sz(len(digest)), b"nizk_example") == 0
    # determinism: same seed -> same bytes ; different seed -> different bytes
    n1 = orc.orc_proof_bytes(p, None, sz(0))
    p2 = vp(orc.orc_nizk_prove(inst, g, digest, sz(len(digest)), b"nizk_example", seed, None))
    b1 = (ctypes.c_uint8 * n1)(); b2 = (ctypes.c_uint8 * n1)()
    orc.orc_proof_tamper(p, 1)
    orc.orc_proof_bytes(p2, b2, sz(n1))
    p3 = vp(orc.orc_nizk_prove(inst, g, digest, sz(len(digest)), b"nizk_example", seed, None))
    orc.orc_proof_bytes(p3, b1, sz(n1))
    assert bytes(b1) == bytes(b2)
    for h in (p, p2, p3): orc.orc_proof_free(h)
    orc.orc_nizk_gens_free(g); orc.orc_instance_free(inst)


@pytest.mark.parametrize("s", [5, 8])
def test_oracle_snark_roundtrip(orc, s):
    N = 1 << s
    inst = vp(orc.orc_instance_synthetic(sz(N), sz(N), sz(10), ctypes.c_uint64(1)))
    g = vp(orc.orc_snark_gens_new(sz(N), sz(N), sz(10), sz(N)))
    e = vp(orc.orc_snark_encode(inst, g))
    seed = u64x4(); orc.orc_seed_scalar(b"tape", ctypes.c_uint64(7), seed)
    p = vp(orc.orc_snark_prove(inst, g, e, b"snark_example", seed, None))
    assert orc.orc_snark_verify(p, inst, g, e, b"snark_example") == 1
    lens = (sz * 3)(); orc.orc_proof_part_lens(p, lens)
    assert lens[0] == sat_proof_len(s)
    for what in (0, 2, 3):
        orc.orc_proof_tamper(p, what)
        assert orc.orc_snark_verify(p, inst, g, e, b"snark_example") == 0
        # undo is not possible for additive tamper; re-prove
        orc.orc_proof_free(p)
        p = vp(orc.orc_snark_prove(inst, g, e, b"snark_example", seed, None))
    orc.orc_proof_free(p); orc.orc_encode_free(e); orc.orc_snark_gens_free(g); orc.orc_instance_free(inst)


def test_oracle_verifies_its_own_serialized_bytes(orc):
    """bincode round trip: ser_snark -> orc_snark_verify_bytes (the entry point used to check GPU proofs at sizes
    where the oracle prover is too slow)."""
    s = 6; N = 1 << s
    inst = vp(orc.orc_instance_synthetic(sz(N), sz(N), sz(10), ctypes.c_uint64(5)))
    g = vp(orc.orc_snark_gens_new(sz(N), sz(N), sz(10), sz(N)))
    e = vp(orc.orc_snark_encode(inst, g))
    seed = u64x4(); orc.orc_seed_scalar(b"tape", ctypes.c_uint64(1), seed)
    p = vp(orc.orc_snark_prove(inst, g, e, b"snark_example", seed, None))
    n = orc.orc_proof_bytes(p, None, sz(0)); pb = (ctypes.c_uint8 * n)(); orc.orc_proof_bytes(p, pb, sz(n))
    comms = []
    for which in (0, 1):
        k = orc.orc_encode_comm(e, ctypes.c_int(which), None, sz(0)); b = (ctypes.c_uint8 * (32 * k))()
        orc.orc_encode_comm(e, ctypes.c_int(which), b, sz(32 * k)); comms.append((bytes(b), k))
    tot = sum(orc.orc_instance_nnz(inst, ctypes.c_int(k)) for k in range(3))
    rows = (ctypes.c_uint64 * tot)(); cols = (ctypes.c_uint64 * tot)(); vals = (ctypes.c_uint64 * (4 * tot))()
    vars_ = (ctypes.c_uint64 * (4 * N))(); inputs = (ctypes.c_uint64 * 40)()
    orc.orc_instance_export(inst, rows, cols, vals, vars_, inputs)
    def verify(b):
        return orc.orc_snark_verify_bytes(bytes(b), sz(len(b)), g, sz(N), sz(N), sz(10), sz(N), sz(2 * N), comms[0][0], sz(comms[0][1]),
                                          comms[1][0], sz(comms[1][1]), inputs, b"snark_example")
    good = bytes(pb)
    assert verify(good) == 1
    assert verify(good[:-1]) == -1                      # truncated
    bad = bytearray(good); bad[len(bad) // 2] ^= 1
    assert verify(bytes(bad)) in (0, -1)                # a flipped bit is rejected (or is no longer a canonical scalar)
    bad = bytearray(good); bad[40] ^= 1                 # inside comm_vars
    assert verify(bytes(bad)) in (0, -1)


# ---- the reference's own known answers for the polynomial layer (unipoly.rs:127-183, dense_mlpoly.rs:433-452) ----
UNIPOLY_KATS = [  # (evaluations at 0,1,2[,3]) -> coefficients low..high, (point, value)
    ([1, 6, 15], [1, 3, 2], (3, 28)),          # 2x^2 + 3x + 1
    ([1, 7, 23, 55], [1, 3, 2, 1], (4, 109)),  # x^3 + 2x^2 + 3x + 1
]


def _unipoly(lib, fn, evals, r):
    n = len(evals)
    co = (ctypes.c_uint64 * (4 * n))(); cc = (ctypes.c_uint64 * (4 * (n - 1)))(); ev = (ctypes.c_uint64 * 4)()
    getattr(lib, fn)(mont_array(evals), sz(n), mont_array([r]), co, cc, ev)
    return from_mont_array(co, n), from_mont_array(cc, n - 1), from_mont_array(ev, 1)[0]


@pytest.mark.parametrize("evals,coeffs,pt", UNIPOLY_KATS)
def test_unipoly_known_answers_oracle(orc, evals, coeffs, pt):
    co, cc, ev = _unipoly(orc, "orc_unipoly_probe", evals, pt[0])
    assert co == coeffs and ev == pt[1]
    assert cc == [coeffs[0]] + coeffs[2:]           # CompressedUniPoly drops the linear term (unipoly.rs:82-88)
    assert coeffs[0] == evals[0] and sum(coeffs) % Q == evals[1]  # eval_at_zero / eval_at_one


@pytest.mark.parametrize("evals,coeffs,pt", UNIPOLY_KATS)
def test_unipoly_known_answers_host_driver(evals, coeffs, pt):
    from spartan_amd import prover
    co, cc, ev = _unipoly(prover.H, "spz_unipoly_probe", evals, pt[0])
    assert co == coeffs and ev == pt[1] and cc == [coeffs[0]] + coeffs[2:]


def test_polynomial_evaluation_known_answer(orc):
    """dense_mlpoly.rs:433-452: Z = [1,2,1,4], r = [4,3] -> 28, directly and through the L/R factorisation"""
    Z, r = [1, 2, 1, 4], [4, 3]
    chi = (ctypes.c_uint64 * 16)()
    orc.orc_eq_evals(mont_array(r), sz(2), chi)
    out = (ctypes.c_uint64 * 4)()
    orc.orc_dot(mont_array(Z), chi, sz(4), out)
    assert from_mont_array(out, 1)[0] == 28
    L = (ctypes.c_uint64 * 8)(); R = (ctypes.c_uint64 * 8)()
    orc.orc_eq_evals(mont_array(r[:1]), sz(1), L); orc.orc_eq_evals(mont_array(r[1:]), sz(1), R)
    LZ = (ctypes.c_uint64 * 8)()
    orc.orc_bound_vecmat(mont_array(Z), sz(2), L, LZ)
    orc.orc_dot(LZ, R, sz(2), out)
    assert from_mont_array(out, 1)[0] == 28
