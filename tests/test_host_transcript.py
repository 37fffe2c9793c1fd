"""CPU tests of the product's host-side Fiat–Shamir layer (spartan_amd/host/transcript.hpp): SHAKE256 vs hashlib,
Merlin vs the published test vector and vs the oracle on random scripts, RandomTape draws vs the oracle."""
import ctypes, hashlib, random
import pytest
from tests.helpers import *


@pytest.fixture(scope="module")
def H():
    from spartan_amd import prover
    return prover.H


def test_shake256(H):
    rng = random.Random(1)
    for n in (0, 1, 135, 136, 137, 1000):
        m = bytes(rng.randrange(256) for _ in range(n))
        out = (ctypes.c_uint8 * 500)()
        H.spz_shake256(m, sz(n), out, sz(500))
        assert bytes(out) == hashlib.shake_256(m).digest(500)


def test_keccak_variants_agree(H):
    """keccak.cc compiles the permutation three ways (plain, BMI2, AVX-512 planes) and picks one per CPU by timing them: every form this
    CPU can run must give the same state as the plain one, and the FIPS 202 zero-state vector."""
    H.spz_keccak_variant.restype = ctypes.c_char_p
    assert H.spz_keccak_variant() in (b"plain", b"bmi2", b"avx512")
    rng = random.Random(3)
    St = ctypes.c_uint64 * 25
    z = St()
    assert H.spz_keccak_run_variant(b"plain", z) == 1
    assert z[0] == 0xF1258F7940E1DDE7 and z[24] == 0xEAF1FF7B5CECA249  # Keccak-f[1600] of the all-zero state (KeccakF-1600-IntermediateValues)
    ran = 0
    for v in (b"bmi2", b"avx512"):
        for _ in range(200):
            words = [rng.randrange(2**64) for _ in range(25)]
            a, b = St(*words), St(*words)
            H.spz_keccak_run_variant(b"plain", a)
            if H.spz_keccak_run_variant(v, b) == 0:
                break  # this CPU cannot run the form
            assert list(a) == list(b), v
            ran += 1
    print("variants checked:", ran)


def _script(lib, fn, tlabel, ops):
    n = len(ops)
    kinds = (ctypes.c_int * n)(*[o[0] for o in ops])
    labels = (ctypes.c_char_p * n)(*[o[1] for o in ops])
    bufs = [ctypes.create_string_buffer(o[2], max(len(o[2]), 1)) if o[0] != 1 else None for o in ops]
    datas = (ctypes.POINTER(ctypes.c_uint8) * n)(*[ctypes.cast(b, ctypes.POINTER(ctypes.c_uint8)) if b is not None else None for b in bufs])
    lens = (sz * n)(*[len(o[2]) if o[0] != 1 else o[2] for o in ops])
    total = sum(o[2] for o in ops if o[0] == 1)
    out = (ctypes.c_uint8 * max(total, 1))()
    got = getattr(lib, fn)(tlabel, sz(n), kinds, labels, datas, lens, out)
    assert got == total
    return bytes(out)[:total]


def test_merlin_kat_and_random_scripts(H, orc):
    kat = _script(H, "spz_merlin_script", b"test protocol", [(0, b"some label", b"some data"), (1, b"challenge", 32)])
    assert kat.hex() == "d5a21972d0d5fe320c0d263fac7fffb8145aa640af6e9bca177c03c7efcf0615"
    rng = random.Random(2)
    for trial in range(20):
        ops = []
        for _ in range(rng.randrange(1, 40)):
            k = rng.randrange(3)
            label = bytes(rng.randrange(97, 123) for _ in range(rng.randrange(1, 20)))
            if k == 0:
                ops.append((0, label, bytes(rng.randrange(256) for _ in range(rng.choice([0, 1, 32, 165, 166, 167, 400])))))
            elif k == 1:
                ops.append((1, label, rng.choice([1, 32, 64, 200])))
            else:
                ops.append((2, label, rng.randrange(2**64).to_bytes(8, "little")))
        ops.append((1, b"final", 64))
        assert _script(H, "spz_merlin_script", b"proto", ops) == _script(orc, "orc_merlin_script", b"proto", ops)


def test_seed_scalar_matches_oracle(H, orc):
    a = u64x4(); b = u64x4()
    for seed in (0, 1, 2**63):
        H.spz_seed_scalar(b"tape", ctypes.c_uint64(seed), a)
        orc.orc_seed_scalar(b"tape", ctypes.c_uint64(seed), b)
        assert list(a) == list(b)


def _state(lib, fn, tlabel, ops):
    n = len(ops)
    kinds = (ctypes.c_int * n)(*[o[0] for o in ops])
    labels = (ctypes.c_char_p * n)(*[o[1] for o in ops])
    bufs = [ctypes.create_string_buffer(o[2], max(len(o[2]), 1)) if o[0] != 1 else None for o in ops]
    datas = (ctypes.POINTER(ctypes.c_uint8) * n)(*[ctypes.cast(b, ctypes.POINTER(ctypes.c_uint8)) if b is not None else None for b in bufs])
    lens = (sz * n)(*[len(o[2]) if o[0] != 1 else o[2] for o in ops])
    out = (ctypes.c_uint8 * 203)()
    getattr(lib, fn)(tlabel, sz(n), kinds, labels, datas, lens, out)
    return out


def test_transcript_state_export_import_matches_oracle(H, orc):
    """The 203 bytes that ARE a merlin transcript (Strobe128 state, pos, pos_begin, cur_flags): what spz_snark_prove_t /
    spz_nizk_prove_t take and return to continue a caller-owned `&mut Transcript` (src/lib.rs:339-347). Export after a script
    equals the oracle's; a transcript continued from the exported state draws what the uninterrupted transcript draws."""
    rng = random.Random(5)
    for trial in range(10):
        ops = []
        for _ in range(rng.randrange(1, 12)):
            label = bytes(rng.randrange(97, 123) for _ in range(rng.randrange(1, 12)))
            ops.append(rng.choice([(0, label, bytes(rng.randrange(256) for _ in range(rng.choice([0, 5, 32, 166, 300])))), (1, label, rng.choice([16, 64])),
                                   (2, label, rng.randrange(2**64).to_bytes(8, "little"))]))
        a, b = _state(H, "spz_merlin_state", b"caller protocol", ops), _state(orc, "orc_merlin_state", b"caller protocol", ops)
        assert bytes(a) == bytes(b)
        whole = _script(H, "spz_merlin_script", b"caller protocol", ops + [(1, b"next", 64)])
        tail = whole[-64:]
        oa, ob = (ctypes.c_uint8 * 64)(), (ctypes.c_uint8 * 64)()
        H.spz_merlin_challenge_from_state(a, b"next", oa, sz(64))
        orc.orc_merlin_challenge_from_state(b, b"next", ob, sz(64))
        assert bytes(oa) == tail and bytes(ob) == tail and bytes(a) == bytes(b)  # the states after the draw agree too


def test_coarse_binding_probe_is_the_script_it_says(H, orc):
    """spz_transcript_probe (host_capi.cc) is the C half of the coarse Rust binding's layout self-check (rust_shim/src/gpu_tail.rs.in,
    `layout_checked`: the binding reads merlin::Transcript's private state through the struct's memory and must not trust `repr(Rust)`):
    the state it returns is the state of Transcript::new(b"spartan_amd binding probe") + append_message(b"probe-message", 0..63) — equal to
    the oracle's independent transcript on the same script — and the challenge is what that transcript draws next. The Rust side panics,
    in release builds too, unless merlin gives the same 203 bytes through the raw copy and the same challenge through the raw write."""
    st, ch = (ctypes.c_uint8 * 203)(), (ctypes.c_uint8 * 32)()
    H.spz_transcript_probe(st, ch)
    ops = [(0, b"probe-message", bytes(range(64)))]
    want = _state(orc, "orc_merlin_state", b"spartan_amd binding probe", ops)
    assert bytes(st) == bytes(want) and bytes(st) == bytes(_state(H, "spz_merlin_state", b"spartan_amd binding probe", ops))
    oc = (ctypes.c_uint8 * 32)()
    orc.orc_merlin_challenge_from_state(want, b"probe-challenge", oc, sz(32))
    assert bytes(ch) == bytes(oc)
    assert st[200] != 0 or st[201] != 0 or st[202] != 0   # the three position / flag bytes are live in this state: a reordered struct cannot pass
    src = open(os.path.join(ROOT, "rust_shim", "src", "gpu_tail.rs.in")).read()
    assert "spz_transcript_probe" in src and "debug_assert" not in src and src.count("layout_checked();") == 2


def _zlib6(H, data, old=0):
    H.spz_zlib_level6.restype = sz
    cap = 2 * len(data) + 1024
    out = (ctypes.c_uint8 * cap)()
    n = H.spz_zlib_level6(data, sz(len(data)), ctypes.c_int(old), out, sz(cap))
    return bytes(out[:n])


def test_shape_digest_deflater_round_trips_and_keeps_miniz_parameters(H, orc):
    """R1CSShape::get_digest (src/r1cs.rs:154-158) = zlib(level 6)(bincode(shape)) through flate2's rust_backend, i.e. miniz's tdefl.
    spartan_amd/host/deflate.cc restates that path (byte equality with the real miniz: test_deflater_equals_real_miniz below); here: (1) every stream inflates with an
    independent implementation (Python zlib) to exactly its input — stored, static and dynamic blocks, multi-block inputs, the
    bincode of a synthetic R1CS shape as the ORACLE serialises it; (2) the stream shape miniz documents: header 0x78 0x9C (0x78 0x01
    with the old-header switch), Adler-32 trailer, a stored block for inputs a coded block would expand, blocks closed by the
    31 KiB / code-buffer rule (so a 150 KB input has several blocks). Byte equality with miniz_oxide needs the Rust run of
    scripts/compare_with_libspartan.sh."""
    import zlib
    rng = random.Random(11)
    shape = None
    N = 1 << 9
    oi = vp(orc.orc_instance_synthetic(sz(N), sz(N), sz(10), ctypes.c_uint64(3)))
    n = orc.orc_instance_shape_bincode(oi, None, sz(0)); b = (ctypes.c_uint8 * n)(); orc.orc_instance_shape_bincode(oi, b, sz(n))
    shape = bytes(b)
    orc.orc_instance_free(oi)
    assert len(shape) == 24 + 3 * (24 + 48 * N)
    cases = [b"", b"a", b"hello hello hello hello", bytes(5000), bytes(rng.randrange(256) for _ in range(40000)), b"abcdefghij" * 20000,
             bytes(rng.randrange(4) for _ in range(150000)), shape]
    for c in cases:
        z = _zlib6(H, c)
        assert zlib.decompress(z) == c
        assert z[:2] == b"\x78\x9c" and z[-4:] == zlib.adler32(c).to_bytes(4, "big")
        z1 = _zlib6(H, c, 1)
        assert z1[:2] == b"\x78\x01" and z1[2:] == z[2:]
    assert _zlib6(H, b"a")[2:8] == b"\x01\x01\x00\xfe\xff\x61"            # one byte: the coded block would not be smaller -> stored, final
    incompressible = cases[4]
    z = _zlib6(H, incompressible)
    assert len(z) == len(incompressible) + 2 + 4 + 5 * 2 and z[2] == 0        # two stored blocks (the 31 KiB rule), not final / final
    d = zlib.decompressobj(); d.decompress(_zlib6(H, shape)); assert d.eof


def _zlib_probes(H, data, probes, old=0):
    H.spz_zlib_probes.restype = sz
    cap = 2 * len(data) + 1024
    out = (ctypes.c_uint8 * cap)()
    n = H.spz_zlib_probes(data, sz(len(data)), ctypes.c_uint(probes), ctypes.c_int(old), out, sz(cap))
    return bytes(out[:n])


def test_deflater_equals_real_miniz(H, orc):
    """THE PIN of the compressed R1CSShapeDigest (src/r1cs.rs:154-158: the digest absorbed by NIZK::prove is the zlib stream itself).
    libtorch_cpu.so bundles the real C miniz (3.0.2) and exports mz_compress2; flate2's rust_backend is miniz_oxide, the Rust port of
    the same tdefl. spartan_amd/host/deflate.cc must produce the SAME BYTES — header, every block, Adler-32 — at level 6 (what
    Compression::default() selects: 128 probes) and, to show the restatement is tdefl itself and not a level-6 coincidence, at every
    other lazy-parsing level (4, 5, 7, 8, 9, 10 = 16 / 32 / 256 / 512 / 768 / 1500 probes). Corpus: the bincode of synthetic R1CS shapes
    2^4 .. 2^16 as the oracle serialises them (the real input; up to 9.4 MB, hundreds of dynamic-Huffman blocks), incompressible data
    (stored blocks), runs, text, tiny inputs (static blocks), inputs around the 31 KiB / 64 K-symbol block rules."""
    assert real_miniz_zlib(b"")[:2] == b"\x78\x9c"
    rng = random.Random(11)
    cases = {"empty": b"", "a": b"a", "hello": b"hello hello hello hello", "zeros": bytes(5000), "rand": bytes(rng.randrange(256) for _ in range(40000)),
             "runs": b"abcdefghij" * 20000, "rand4": bytes(rng.randrange(4) for _ in range(150000)),
             "text": open(os.path.join(ROOT, "SURVEY.md"), "rb").read(), "47": bytes(range(47)), "48": bytes(range(48)),
             "31k": bytes(rng.randrange(16) for _ in range(31 * 1024 + 1)), "64ksym": bytes(rng.randrange(2) for _ in range(70000))}
    for s in (4, 8, 9, 12, 14, 16):
        N = 1 << s
        oi = vp(orc.orc_instance_synthetic(sz(N), sz(N), sz(10), ctypes.c_uint64(3)))
        cases["shape2^%d" % s] = oracle_shape_bincode(orc, oi)
        orc.orc_instance_free(oi)
    assert len(cases["shape2^16"]) == 24 + 3 * (24 + 48 * 65536)
    for name, c in cases.items():
        want = real_miniz_zlib(c, 6)
        assert _zlib6(H, c) == want, name                      # the product's digest function, whole stream
        assert _zlib_probes(H, c, 128) == want, name
    probes = {4: 16, 5: 32, 7: 256, 8: 512, 9: 768, 10: 1500}
    for name in ("hello", "rand", "runs", "rand4", "text", "31k", "shape2^4", "shape2^9", "shape2^12", "shape2^14"):
        for level, pr in probes.items():
            assert _zlib_probes(H, cases[name], pr) == real_miniz_zlib(cases[name], level), (name, level)
    # the old-header variant (miniz < 2.2, miniz_oxide 0.3) changes the second byte and nothing else
    z0, z1 = _zlib6(H, cases["shape2^9"], 0), _zlib6(H, cases["shape2^9"], 1)
    assert z1[:2] == b"\x78\x01" and z0[:2] == b"\x78\x9c" and z1[2:] == z0[2:]
