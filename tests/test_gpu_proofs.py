"""Whole-proof byte parity: NIZK::prove / SNARK::prove through the HIP path (host driver over the C ABI) against
the oracle on the same seeded instance and RandomTape seed; the oracle's restated verifier accepts the bytes' source."""
import ctypes
import pytest
from tests.helpers import *

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def P():
    from spartan_amd import prover
    return prover


@pytest.fixture(scope="module")
def ctx(P):
    c = P.Ctx(0)
    yield c
    c.close()


def oracle_bytes(orc, p):
    n = orc.orc_proof_bytes(p, None, sz(0))
    b = (ctypes.c_uint8 * n)()
    orc.orc_proof_bytes(p, b, sz(n))
    return bytes(b)


def _oracle_threads(orc, s):
    """the oracle's row commitments run on all cores for the BASELINE-sized live comparisons (same bytes, less waiting)"""
    import os
    orc.orc_set_threads(ctypes.c_int(min(32, os.cpu_count() or 1) if s >= 14 else 1))


@pytest.mark.parametrize("s,seed", [(1, 0), (2, 1), (4, 2), (7, 3), (10, 4), (13, 5), (16, 6)])
def test_nizk_prove_bytes_match_oracle(P, ctx, orc, s, seed):
    _oracle_threads(orc, s)
    N = 1 << s
    ni = 10 if N > 16 else 1
    inst = P.Instance.produce_synthetic_r1cs(ctx, N, N, ni, seed=seed)
    digest = b"digest-%d" % s
    inst.set_digest(digest)
    gens = P.NIZKGens(ctx, N, N, ni)
    tape = P.seed_scalar(b"tape", seed)
    got = P.NIZK.prove(ctx, inst, inst.vars, inst.inputs, gens, b"nizk_example", tape)
    oi = vp(orc.orc_instance_synthetic(sz(N), sz(N), sz(ni), ctypes.c_uint64(seed)))
    og = vp(orc.orc_nizk_gens_new(sz(N), sz(N), sz(ni)))
    op = vp(orc.orc_nizk_prove(oi, og, digest, sz(len(digest)), b"nizk_example", tape, None))
    assert orc.orc_nizk_verify(op, oi, og, digest, sz(len(digest)), b"nizk_example") == 1
    want = oracle_bytes(orc, op)
    assert len(got) == len(want)
    assert got == want
    orc.orc_proof_free(op); orc.orc_nizk_gens_free(og); orc.orc_instance_free(oi)
    gens.free(); inst.free()


@pytest.mark.parametrize("s,seed", [(1, 0), (3, 1), (5, 2), (8, 3), (11, 4), (14, 5), (16, 6)])
def test_snark_encode_and_prove_bytes_match_oracle(P, ctx, orc, s, seed):
    """(16, 6) is BASELINE config 2's size, compared live; 2^20 and 2^22 are compared against the committed oracle digests
    in tests/test_golden.py::test_hip_path_reproduces_baseline_sized_fixtures."""
    _oracle_threads(orc, s)
    N = 1 << s
    ni = 10 if N > 16 else 1
    inst = P.Instance.produce_synthetic_r1cs(ctx, N, N, ni, seed=seed)
    gens = P.SNARKGens(ctx, N, N, ni, N)
    enc = P.SNARK.encode(ctx, inst, gens)
    oi = vp(orc.orc_instance_synthetic(sz(N), sz(N), sz(ni), ctypes.c_uint64(seed)))
    og = vp(orc.orc_snark_gens_new(sz(N), sz(N), sz(ni), sz(N)))
    oe = vp(orc.orc_snark_encode(oi, og))
    for which in (0, 1):  # ComputationCommitment: comm_comb_ops, comm_comb_mem
        n = orc.orc_encode_comm(oe, ctypes.c_int(which), None, sz(0))
        b = (ctypes.c_uint8 * (32 * n))()
        orc.orc_encode_comm(oe, ctypes.c_int(which), b, sz(32 * n))
        assert enc.comm(which) == bytes(b)
    tape = P.seed_scalar(b"tape", 100 + seed)
    times = {}
    got = P.SNARK.prove(ctx, inst, enc, inst.vars, inst.inputs, gens, b"snark_example", tape, times)
    op = vp(orc.orc_snark_prove(oi, og, oe, b"snark_example", tape, None))
    assert orc.orc_snark_verify(op, oi, og, oe, b"snark_example") == 1
    want = oracle_bytes(orc, op)
    assert len(got) == len(want)
    if got != want:
        first = next(i for i in range(len(got)) if got[i] != want[i])
        pytest.fail(f"first differing byte at offset {first} of {len(got)}")
    assert times["total"] > 0
    orc.orc_proof_free(op); orc.orc_encode_free(oe); orc.orc_snark_gens_free(og); orc.orc_instance_free(oi)
    enc.free(); gens.free(); inst.free()


def _caller_transcript_state(lib, fn):
    """a transcript that is NOT fresh: Transcript::new(b"caller protocol") that already absorbed two messages and drew a challenge"""
    from tests.test_host_transcript import _state
    return _state(lib, fn, b"caller protocol", [(0, b"session", b"\x01\x02\x03 some earlier statement"), (2, b"epoch", (77).to_bytes(8, "little")),
                                               (1, b"earlier-challenge", 32)])


@pytest.mark.parametrize("s,seed", [(6, 3), (12, 4)])
def test_prove_continues_a_caller_owned_transcript(P, ctx, orc, s, seed):
    """SNARK::prove / NIZK::prove take `transcript: &mut Transcript` (src/lib.rs:339-347, 501-509): the caller may have used it
    before. spz_snark_prove_t / spz_nizk_prove_t continue the 203-byte merlin state they are handed and give it back; the
    oracle proves on the same pre-used transcript. Same proof bytes, same transcript state afterwards (so whatever the caller
    draws next agrees too), and the oracle's verifier accepts on its own copy of the pre-used transcript."""
    N = 1 << s
    orc.orc_snark_prove_t.restype = vp; orc.orc_nizk_prove_t.restype = vp
    inst = P.Instance.produce_synthetic_r1cs(ctx, N, N, 10, seed=seed)
    oi = vp(orc.orc_instance_synthetic(sz(N), sz(N), sz(10), ctypes.c_uint64(seed)))
    tape = P.seed_scalar(b"tape", 40 + seed)
    # SNARK
    gens = P.SNARKGens(ctx, N, N, 10, N)
    enc = P.SNARK.encode(ctx, inst, gens)
    og = vp(orc.orc_snark_gens_new(sz(N), sz(N), sz(10), sz(N)))
    oe = vp(orc.orc_snark_encode(oi, og))
    st_h, st_o, st_v = _caller_transcript_state(P.H, "spz_merlin_state"), _caller_transcript_state(orc, "orc_merlin_state"), _caller_transcript_state(orc, "orc_merlin_state")
    assert bytes(st_h) == bytes(st_o)
    got = P.SNARK.prove_t(ctx, inst, enc, inst.vars, inst.inputs, gens, st_h, tape)
    op = vp(orc.orc_snark_prove_t(oi, og, oe, st_o, tape))
    assert got == oracle_bytes(orc, op)
    assert bytes(st_h) == bytes(st_o)                      # the caller's transcript after the proof
    assert orc.orc_snark_verify_t(op, oi, og, oe, st_v) == 1 and bytes(st_v) == bytes(st_o)   # prover and verifier transcripts stay in step
    fresh = P.SNARK.prove(ctx, inst, enc, inst.vars, inst.inputs, gens, b"caller protocol", tape)
    assert fresh != got                                     # the earlier messages are bound into the proof
    st2 = _caller_transcript_state(P.H, "spz_merlin_state")
    va = P.VarsAssignment(ctx, inst.vars)
    assert P.SNARK.prove_t(ctx, inst, enc, va, inst.inputs, gens, st2, tape) == got and bytes(st2) == bytes(st_h)   # resident assignment: same
    va.free()
    orc.orc_proof_free(op); orc.orc_encode_free(oe); orc.orc_snark_gens_free(og)
    enc.free(); gens.free()
    # NIZK
    inst.set_digest(b"shape-digest")
    ng = P.NIZKGens(ctx, N, N, 10)
    ong = vp(orc.orc_nizk_gens_new(sz(N), sz(N), sz(10)))
    st_h, st_o = _caller_transcript_state(P.H, "spz_merlin_state"), _caller_transcript_state(orc, "orc_merlin_state")
    got = P.NIZK.prove_t(ctx, inst, inst.vars, inst.inputs, ng, st_h, tape)
    op = vp(orc.orc_nizk_prove_t(oi, ong, b"shape-digest", sz(12), st_o, tape))
    assert got == oracle_bytes(orc, op) and bytes(st_h) == bytes(st_o)
    orc.orc_proof_free(op); orc.orc_nizk_gens_free(ong); orc.orc_instance_free(oi)
    ng.free(); inst.free()


def test_nizk_binds_to_the_shape_digest_it_computes(P, ctx, orc):
    """NIZK::prove absorbs R1CSShape::get_digest (lib.rs:514, r1cs.rs:154-158). Without a caller-supplied digest the driver
    computes it (deflate.cc): it must inflate (Python zlib) to the bincode of the shape as the oracle serialises it, and the
    proof must equal the oracle's proof over the same digest bytes."""
    import zlib
    s, seed = 8, 5
    N = 1 << s
    P.H.spz_instance_digest.restype = sz
    inst = P.Instance.produce_synthetic_r1cs(ctx, N, N, 10, seed=seed)
    n = P.H.spz_instance_digest(inst.h, None, sz(0)); buf = (ctypes.c_uint8 * n)(); P.H.spz_instance_digest(inst.h, buf, sz(n))
    digest = bytes(buf)
    oi = vp(orc.orc_instance_synthetic(sz(N), sz(N), sz(10), ctypes.c_uint64(seed)))
    m = orc.orc_instance_shape_bincode(oi, None, sz(0)); sb = (ctypes.c_uint8 * m)(); orc.orc_instance_shape_bincode(oi, sb, sz(m))
    assert zlib.decompress(digest) == bytes(sb)
    gens = P.NIZKGens(ctx, N, N, 10)
    tape = P.seed_scalar(b"tape", seed)
    got = P.NIZK.prove(ctx, inst, inst.vars, inst.inputs, gens, b"nizk_example", tape)     # no set_digest: computed
    og = vp(orc.orc_nizk_gens_new(sz(N), sz(N), sz(10)))
    op = vp(orc.orc_nizk_prove(oi, og, digest, sz(len(digest)), b"nizk_example", tape, None))
    assert got == oracle_bytes(orc, op)
    orc.orc_proof_free(op); orc.orc_nizk_gens_free(og); orc.orc_instance_free(oi)
    gens.free(); inst.free()


def test_generator_streams_match_oracle(P, ctx, orc):
    gens = P.SNARKGens(ctx, 64, 64, 10, 64)
    sat = gens.stream(0); ev = gens.stream(1)
    assert sat == gens_bytes(orc, len(sat) // 32 - 1, b"gens_r1cs_sat")
    assert ev == gens_bytes(orc, len(ev) // 32 - 1, b"gens_r1cs_eval")
    gens.free()


def test_instance_new_matches_synthetic_and_rejects_bad_input(P, ctx, orc):
    """Instance::new (lib.rs:121-228) fed with the oracle's exported entries gives the same proof as produce_synthetic_r1cs."""
    s, seed = 5, 9
    N = 1 << s
    oi = vp(orc.orc_instance_synthetic(sz(N), sz(N), sz(10), ctypes.c_uint64(seed)))
    nnz = [orc.orc_instance_nnz(oi, ctypes.c_int(k)) for k in range(3)]
    tot = sum(nnz)
    rows = (ctypes.c_uint64 * tot)(); cols = (ctypes.c_uint64 * tot)(); vals = (ctypes.c_uint64 * (4 * tot))()
    vars_ = (ctypes.c_uint64 * (4 * N))(); inputs = (ctypes.c_uint64 * 40)()
    orc.orc_instance_export(oi, rows, cols, vals, vars_, inputs)
    vb = b"".join(int(v).to_bytes(32, "little") for v in from_mont_array(vals, tot))
    inst = P.Instance.new(ctx, N, N, 10, nnz, rows, cols, vb)
    inst.set_digest(b"d")
    ref = P.Instance.produce_synthetic_r1cs(ctx, N, N, 10, seed=seed)
    ref.set_digest(b"d")
    assert list(ref.vars) == list(vars_)
    gens = P.NIZKGens(ctx, N, N, 10)
    tape = P.seed_scalar(b"tape", 1)
    a = P.NIZK.prove(ctx, inst, vars_, inputs, gens, b"nizk_example", tape)
    b = P.NIZK.prove(ctx, ref, ref.vars, ref.inputs, gens, b"nizk_example", tape)
    assert a == b
    # error paths of Instance::new (lib.rs tests :627-690): InvalidIndex, InvalidScalar
    bad_rows = (ctypes.c_uint64 * tot)(*rows); bad_rows[0] = N
    with pytest.raises(P.SpartanHipError, match="InvalidIndex"):
        P.Instance.new(ctx, N, N, 10, nnz, bad_rows, cols, vb)
    larger_than_mod = bytes([3, 0, 0, 0, 255, 255, 255, 255, 254, 91, 254, 255, 2, 164, 189, 83, 5, 216, 161, 9, 8, 216, 57, 51, 72, 125, 157, 41, 83, 167, 237, 115])
    with pytest.raises(P.SpartanHipError, match="InvalidScalar"):
        P.Instance.new(ctx, N, N, 10, nnz, rows, cols, larger_than_mod + vb[32:])
    gens.free(); inst.free(); ref.free(); orc.orc_instance_free(oi)


@pytest.mark.parametrize("s", [16, 20, 22, 24])
def test_snark_full_size_properties(P, ctx, orc, s):
    """BASELINE sizes (configs[1] 2^16, configs[2] 2^20, configs[4] 2^22) and 2^24 — the largest instance whose generator
    tables fit one GPU's HBM (about 160 GB of tables: 20-22 windows per stream) — where the oracle prover is too slow to run in a test: size-independent properties — README proof
    lengths, determinism, and the oracle's restated VERIFIER accepts the GPU proof bytes against the GPU computation commitment
    (and rejects a corrupted proof)."""
    N = 1 << s
    inst = P.Instance.produce_synthetic_r1cs(ctx, N, N, 10, seed=s)
    gens = P.SNARKGens(ctx, N, N, 10, N)
    if s == 24:
        # planned together (sp_gens_plan_pair) against the memory that is free in THIS process: the pair with the fewest additions per proof (the
        # witness's 2^24 scalars x windows(0) + the six derefs vectors x windows(1)) that fits: 22 / 20 (164 GB) in a process of its own, a
        # neighbouring pair when earlier tests hold memory; rounds 2-5: uniform 14 / 12 bits = 19 / 22 windows (177 GB), the bound asserted here
        assert gens.windows(0) + 6 * gens.windows(1) <= 19 + 6 * 22
    enc = P.SNARK.encode(ctx, inst, gens)
    tape = P.seed_scalar(b"tape", s)
    proof = P.SNARK.prove(ctx, inst, enc, inst.vars, inst.inputs, gens, b"snark_example", tape)
    va = P.VarsAssignment(ctx, inst.vars)   # the same proof from an assignment already in HBM
    assert P.SNARK.prove(ctx, inst, enc, va, inst.inputs, gens, b"snark_example", tape) == proof
    va.free()
    from tests.test_oracle_pins import sat_proof_len
    if s == 20:
        assert sat_proof_len(20) == 47024 and len(proof) == 47024 + 96 + 133720  # README.md:362,374 + 3 inst_evals
    og = vp(orc.orc_snark_gens_new(sz(N), sz(N), sz(10), sz(N)))
    ops, mem = enc.comm(0), enc.comm(1)
    def verify(b):
        return orc.orc_snark_verify_bytes(b, sz(len(b)), og, sz(N), sz(N), sz(10), sz(N), sz(2 * N), ops, sz(len(ops) // 32), mem,
                                          sz(len(mem) // 32), inst.inputs, b"snark_example")
    assert verify(proof) == 1
    bad = bytearray(proof); bad[8 + 32 * 3 + 5] ^= 0x10   # corrupt one witness-commitment share
    assert verify(bytes(bad)) in (0, -1)
    orc.orc_snark_gens_free(og); enc.free(); gens.free(); inst.free()


def test_padded_constraints_like_reference(P, ctx, orc):
    """lib.rs:672-753 test_padded_constraints: num_cons = 1, num_vars = 0, num_inputs = 3 (a^2 + b + 13 = z). Exercises the
    padding rules of Instance::new (zero-variable / one-constraint), SNARK and NIZK; bytes equal the oracle's and verify."""
    num_cons, num_vars, num_inputs, nnz_param = 1, 0, 3, 3
    le = lambda x: (x % Q).to_bytes(32, "little")
    A = [(0, num_vars + 2, le(1))]
    B = [(0, num_vars + 2, le(1))]
    C = [(0, num_vars + 1, le(1)), (0, num_vars, le(-13)), (0, num_vars + 3, le(-1))]
    nnz = [len(A), len(B), len(C)]
    ent = A + B + C
    rows = (ctypes.c_uint64 * len(ent))(*[e[0] for e in ent]); cols = (ctypes.c_uint64 * len(ent))(*[e[1] for e in ent])
    vals = b"".join(e[2] for e in ent)
    vars_ = (ctypes.c_uint64 * 4)()           # no variables assigned (the prover pads to num_vars_padded = 4)
    inputs = mont_array([16, 1, 2])
    inst = P.Instance.new(ctx, num_cons, num_vars, num_inputs, nnz, rows, cols, vals)
    inst.num_inputs = num_inputs
    err = ctypes.c_int(0)
    oi = vp(orc.orc_instance_new_padded(sz(num_cons), sz(num_vars), sz(num_inputs), (sz * 3)(*nnz), rows, cols, vals, vars_, sz(0), inputs,
                                        ctypes.byref(err)))
    assert err.value == 0 and oi
    tape = P.seed_scalar(b"tape", 77)
    empty = (ctypes.c_uint64 * 0)()
    # SNARK
    gens = P.SNARKGens(ctx, num_cons, num_vars, num_inputs, nnz_param)
    enc = P.SNARK.encode(ctx, inst, gens)
    got = P.SNARK.prove(ctx, inst, enc, empty, inputs, gens, b"snark_example", tape)
    og = vp(orc.orc_snark_gens_new(sz(num_cons), sz(num_vars), sz(num_inputs), sz(nnz_param)))
    oe = vp(orc.orc_snark_encode(oi, og))
    op = vp(orc.orc_snark_prove(oi, og, oe, b"snark_example", tape, None))
    assert orc.orc_snark_verify(op, oi, og, oe, b"snark_example") == 1
    assert got == oracle_bytes(orc, op)
    # NIZK
    inst.set_digest(b"padded")
    ngens = P.NIZKGens(ctx, num_cons, num_vars, num_inputs)
    got = P.NIZK.prove(ctx, inst, empty, inputs, ngens, b"nizk_example", tape)
    ong = vp(orc.orc_nizk_gens_new(sz(num_cons), sz(num_vars), sz(num_inputs)))
    onp = vp(orc.orc_nizk_prove(oi, ong, b"padded", sz(6), b"nizk_example", tape, None))
    assert orc.orc_nizk_verify(onp, oi, ong, b"padded", sz(6), b"nizk_example") == 1
    assert got == oracle_bytes(orc, onp)
    # error paths of the oracle restatement agree with the product's (InvalidIndex / InvalidScalar)
    bad_rows = (ctypes.c_uint64 * len(ent))(*[1] + [e[0] for e in ent[1:]])
    assert not orc.orc_instance_new_padded(sz(num_cons), sz(num_vars), sz(num_inputs), (sz * 3)(*nnz), bad_rows, cols, vals, vars_, sz(0), inputs, ctypes.byref(err)) and err.value == 1
    with pytest.raises(P.SpartanHipError, match="InvalidIndex"):
        P.Instance.new(ctx, num_cons, num_vars, num_inputs, nnz, bad_rows, cols, vals)
    ngens.free(); enc.free(); gens.free(); inst.free()


def test_concurrent_contexts_produce_identical_proofs(P, orc):
    """Three contexts on the same GPU proving from three host threads at once (bench.py's throughput mode): every proof
    equals its single-stream bytes, which equal the oracle's."""
    import threading
    s_ = 9; N = 1 << s_
    work = []
    for k in range(3):
        c = P.Ctx(0)
        inst = P.Instance.produce_synthetic_r1cs(c, N, N, 10, seed=50 + k)
        gens = P.SNARKGens(c, N, N, 10, N)
        enc = P.SNARK.encode(c, inst, gens)
        tape = P.seed_scalar(b"tape", 50 + k)
        single = P.SNARK.prove(c, inst, enc, inst.vars, inst.inputs, gens, b"snark_example", tape)
        work.append([c, inst, gens, enc, tape, single, []])

    def run(w):
        for _ in range(4):
            w[6].append(P.SNARK.prove(w[0], w[1], w[3], w[1].vars, w[1].inputs, w[2], b"snark_example", w[4]))
    ths = [threading.Thread(target=run, args=(w,)) for w in work]
    for t in ths: t.start()
    for t in ths: t.join()
    for k, w in enumerate(work):
        assert len(w[6]) == 4 and all(p == w[5] for p in w[6])
    # one of them against the oracle
    k = 1
    oi = vp(orc.orc_instance_synthetic(sz(N), sz(N), sz(10), ctypes.c_uint64(50 + k)))
    og = vp(orc.orc_snark_gens_new(sz(N), sz(N), sz(10), sz(N)))
    oe = vp(orc.orc_snark_encode(oi, og))
    op = vp(orc.orc_snark_prove(oi, og, oe, b"snark_example", work[k][4], None))
    assert oracle_bytes(orc, op) == work[k][5]
    for w in work:
        w[3].free(); w[2].free(); w[1].free(); w[0].close()


def test_wire_formats_of_gens_and_commitment_match_oracle(P, ctx, orc):
    """bincode of SNARKGens (lib.rs:278-282), ComputationCommitment (lib.rs:44-48) and ComputationDecommitment (lib.rs:50-54):
    same bytes as the oracle's writer; the lengths follow from the serde struct definitions."""
    s_ = 6; N = 1 << s_
    inst = P.Instance.produce_synthetic_r1cs(ctx, N, N, 10, seed=4)
    gens = P.SNARKGens(ctx, N, N, 10, N)
    enc = P.SNARK.encode(ctx, inst, gens)
    oi = vp(orc.orc_instance_synthetic(sz(N), sz(N), sz(10), ctypes.c_uint64(4)))
    og = vp(orc.orc_snark_gens_new(sz(N), sz(N), sz(10), sz(N)))
    oe = vp(orc.orc_snark_encode(oi, og))
    def ob(fn, h):
        n = fn(h, None, sz(0)); b = (ctypes.c_uint8 * n)(); fn(h, b, sz(n)); return bytes(b)
    gb = gens.serialize()
    assert gb == ob(orc.orc_snark_gens_bincode, og)
    cb = enc.serialize_commitment()
    assert cb == ob(orc.orc_commitment_bincode, oe)
    # sizes from the struct layout: MultiCommitGens{n} = 16 + 32*(n+1); PolyCommitmentGens(n) = 8 + mcg(n) + mcg(1)
    mcg = lambda n: 16 + 32 * (n + 1)
    pcg = lambda n: 8 + mcg(n) + mcg(1)
    r_sat = 1 << (s_ - s_ // 2)
    v_ops, v_mem, v_der = s_ + 4, s_ + 2, s_ + 3   # log2(nnz)+log2(16), max(nvx, nvy)+1, log2(nnz)+log2(8)
    R = lambda v: 1 << (v - v // 2)
    assert len(gb) == mcg(1) + mcg(3) + mcg(4) + pcg(r_sat) + pcg(R(v_ops)) + pcg(R(v_mem)) + pcg(R(v_der))
    assert len(cb) == 6 * 8 + (8 + 32 * (1 << (v_ops // 2))) + (8 + 32 * (1 << (v_mem // 2)))
    # ComputationDecommitment (lib.rs:50-54): the dense representation downloaded from the device, serde field order
    db = enc.serialize_decommitment()
    assert db == ob(orc.orc_decommitment_bincode, oe)
    dp = lambda n: 8 + 8 + 8 + 32 * n                      # DensePolynomial {num_vars, len, Vec<Scalar>}
    at = lambda n, cells: (8 + 3 * (8 + 8 * n)) + 2 * (8 + 3 * dp(n)) + dp(cells)   # AddrTimestamps for 3 matrices
    cells = 2 * N                                           # max(2^nvx, 2^nvy) with nvy = log2(2 * num_vars)
    assert len(db) == 8 + (8 + 3 * dp(N)) + 2 * at(N, cells) + dp(16 * N) + dp(2 * cells)
    enc.free(); gens.free(); inst.free()


@pytest.mark.parametrize("lc,lv,ni", [(6, 4, 3), (4, 7, 10), (9, 6, 20), (5, 5, 31), (7, 8, 0)])
def test_rectangular_instances_match_oracle(P, ctx, orc, lc, lv, ni):
    """num_cons != num_vars (so rx and ry differ in length and SparseMatPolyEvalProof's `equalize` pads either side,
    sparse_mlpoly.rs:1429-1445), and input counts from 0 to num_vars - 1."""
    nc, nv = 1 << lc, 1 << lv
    seed = lc * 100 + lv
    inst = P.Instance.produce_synthetic_r1cs(ctx, nc, nv, ni, seed=seed)
    gens = P.SNARKGens(ctx, nc, nv, ni, nc)
    enc = P.SNARK.encode(ctx, inst, gens)
    tape = P.seed_scalar(b"tape", seed)
    got = P.SNARK.prove(ctx, inst, enc, inst.vars, inst.inputs, gens, b"snark_example", tape)
    oi = vp(orc.orc_instance_synthetic(sz(nc), sz(nv), sz(ni), ctypes.c_uint64(seed)))
    assert orc.orc_instance_is_sat(oi) == 1
    og = vp(orc.orc_snark_gens_new(sz(nc), sz(nv), sz(ni), sz(nc)))
    oe = vp(orc.orc_snark_encode(oi, og))
    op = vp(orc.orc_snark_prove(oi, og, oe, b"snark_example", tape, None))
    assert orc.orc_snark_verify(op, oi, og, oe, b"snark_example") == 1
    assert got == oracle_bytes(orc, op)
    inst.set_digest(b"rect")
    ngens = P.NIZKGens(ctx, nc, nv, ni)
    gotn = P.NIZK.prove(ctx, inst, inst.vars, inst.inputs, ngens, b"nizk_example", tape)
    ong = vp(orc.orc_nizk_gens_new(sz(nc), sz(nv), sz(ni)))
    onp = vp(orc.orc_nizk_prove(oi, ong, b"rect", sz(4), b"nizk_example", tape, None))
    assert gotn == oracle_bytes(orc, onp)
    ngens.free(); enc.free(); gens.free(); inst.free()


def test_tiny_r1cs_of_the_reference(P, ctx, orc):
    """r1csproof.rs:474-560 produce_tiny_r1cs: three hand-written constraints in a 128 x 256 instance with 2 inputs — almost
    every row and column of A, B, C is empty, and B, C address the constant and the inputs. is_sat, NIZK and SNARK bytes
    equal the oracle's and verify."""
    num_cons, num_vars, num_inputs = 128, 256, 2
    le = lambda x: (x % Q).to_bytes(32, "little")
    A = [(0, 0, le(1)), (0, 1, le(1)), (1, 0, le(1)), (1, num_vars + 2, le(1)), (2, 4, le(1))]
    B = [(0, num_vars + 1, le(1)), (1, 2, le(1)), (2, num_vars, le(1))]
    C = [(0, 2, le(1)), (1, 3, le(1))]
    rng = random.Random(11)
    i0, i1, z1, z2 = (rng.randrange(Q) for _ in range(4))
    z3 = (z1 + z2) * i0 % Q
    z4 = (z1 + i1) * z3 % Q
    vars_py = [z1, z2, z3, z4, 0] + [0] * (num_vars - 5)
    nnz = [len(A), len(B), len(C)]
    ent = A + B + C
    rows = (ctypes.c_uint64 * len(ent))(*[e[0] for e in ent]); cols = (ctypes.c_uint64 * len(ent))(*[e[1] for e in ent])
    vals = b"".join(e[2] for e in ent)
    vars_ = mont_array(vars_py); inputs = mont_array([i0, i1])
    inst = P.Instance.new(ctx, num_cons, num_vars, num_inputs, nnz, rows, cols, vals)
    err = ctypes.c_int(0)
    oi = vp(orc.orc_instance_new_padded(sz(num_cons), sz(num_vars), sz(num_inputs), (sz * 3)(*nnz), rows, cols, vals, vars_, sz(num_vars), inputs,
                                        ctypes.byref(err)))
    assert err.value == 0 and oi and orc.orc_instance_is_sat(oi) == 1
    tape = P.seed_scalar(b"tape", 5)
    gens = P.SNARKGens(ctx, num_cons, num_vars, num_inputs, 5)
    enc = P.SNARK.encode(ctx, inst, gens)
    got = P.SNARK.prove(ctx, inst, enc, vars_, inputs, gens, b"snark_example", tape)
    og = vp(orc.orc_snark_gens_new(sz(num_cons), sz(num_vars), sz(num_inputs), sz(5)))
    oe = vp(orc.orc_snark_encode(oi, og))
    op = vp(orc.orc_snark_prove(oi, og, oe, b"snark_example", tape, None))
    assert orc.orc_snark_verify(op, oi, og, oe, b"snark_example") == 1
    assert got == oracle_bytes(orc, op)
    inst.set_digest(b"tiny")
    ngens = P.NIZKGens(ctx, num_cons, num_vars, num_inputs)
    got = P.NIZK.prove(ctx, inst, vars_, inputs, ngens, b"nizk_example", tape)
    ong = vp(orc.orc_nizk_gens_new(sz(num_cons), sz(num_vars), sz(num_inputs)))
    onp = vp(orc.orc_nizk_prove(oi, ong, b"tiny", sz(4), b"nizk_example", tape, None))
    assert orc.orc_nizk_verify(onp, oi, ong, b"tiny", sz(4), b"nizk_example") == 1
    assert got == oracle_bytes(orc, onp)
    ngens.free(); enc.free(); gens.free(); inst.free()


def test_host_and_device_point_encoding_give_the_same_proof(P, orc, monkeypatch):
    """Commitments of up to 8 rows are summed on the GPU and encoded (RFC 9496 §4.3.2) by the calling host core, because one
    serial inverse-square-root chain takes ~3 us there and ~100 us on a lone wavefront. Option encode.device = 1
    keeps that chain on the GPU: both contexts must produce the oracle's bytes."""
    s, seed = 9, 21
    N = 1 << s
    proofs = []
    for device_encode in (False, True):
        c = P.Ctx(0)
        if device_encode:
            c.set_option("testing.unlock", 1); c.set_option("encode.device", 1)
        inst = P.Instance.produce_synthetic_r1cs(c, N, N, 10, seed=seed)
        gens = P.SNARKGens(c, N, N, 10, N)
        enc = P.SNARK.encode(c, inst, gens)
        proofs.append(P.SNARK.prove(c, inst, enc, inst.vars, inst.inputs, gens, b"snark_example", P.seed_scalar(b"tape", seed)))
        enc.free(); gens.free(); inst.free(); c.close()
    oi = vp(orc.orc_instance_synthetic(sz(N), sz(N), sz(10), ctypes.c_uint64(seed)))
    og = vp(orc.orc_snark_gens_new(sz(N), sz(N), sz(10), sz(N)))
    oe = vp(orc.orc_snark_encode(oi, og))
    op = vp(orc.orc_snark_prove(oi, og, oe, b"snark_example", P.seed_scalar(b"tape", seed), None))
    assert proofs[0] == proofs[1] == oracle_bytes(orc, op)


def test_small_commitments_on_host_core_or_device_and_resident_assignment_give_the_same_proof(P, ctx, orc):
    """The 2..5-term commitments of the Sigma protocols run on the proving thread's core by default (small_msm.cc) and on
    the GPU with option commit.small_device = 1; the assignment may be a host buffer or a VarsAssignment already in HBM. Four
    combinations, one proof — the oracle's — for SNARK and NIZK."""
    s, seed = 10, 33
    N = 1 << s
    inst = P.Instance.produce_synthetic_r1cs(ctx, N, N, 10, seed=seed)
    inst.set_digest(b"digest-10")
    gens = P.SNARKGens(ctx, N, N, 10, N)
    ngens = P.NIZKGens(ctx, N, N, 10)
    enc = P.SNARK.encode(ctx, inst, gens)
    va = P.VarsAssignment(ctx, inst.vars)
    tape = P.seed_scalar(b"tape", seed)
    snark, nizk = [], []
    try:
        for mode in (1, 0):
            ctx.set_option("testing.unlock", 1); ctx.set_option("commit.small_device", 1 - mode)
            for v in (inst.vars, va):
                snark.append(P.SNARK.prove(ctx, inst, enc, v, inst.inputs, gens, b"snark_example", tape))
                nizk.append(P.NIZK.prove(ctx, inst, v, inst.inputs, ngens, b"nizk_example", tape))
    finally:
        ctx.set_option("commit.small_device", 0); ctx.set_option("testing.unlock", 0)
    oi = vp(orc.orc_instance_synthetic(sz(N), sz(N), sz(10), ctypes.c_uint64(seed)))
    og = vp(orc.orc_snark_gens_new(sz(N), sz(N), sz(10), sz(N)))
    oe = vp(orc.orc_snark_encode(oi, og))
    op = vp(orc.orc_snark_prove(oi, og, oe, b"snark_example", tape, None))
    assert all(p == oracle_bytes(orc, op) for p in snark)
    ong = vp(orc.orc_nizk_gens_new(sz(N), sz(N), sz(10)))
    onp = vp(orc.orc_nizk_prove(oi, ong, b"digest-10", sz(9), b"nizk_example", tape, None))
    assert all(p == oracle_bytes(orc, onp) for p in nizk)
    va.free(); ngens.free(); enc.free(); gens.free(); inst.free()


def test_every_ab_switch_gives_the_same_proof(orc):
    """Every tier-1 option of the library (spartan_amd/csrc/options.hpp: the A/B switches that restore the form an optimisation replaced, or
    move a piece of work between the proving core and the device) must give the oracle's bytes, and so must the tier-0 ones that change
    launch plans: the eq table as a factor vs bound like any table, hash layers fused with the first multiplication layer vs separate,
    dedicated vs unified addition in the inner-product trees, the end of the inner-product arguments on the proving core vs on the device,
    each Keccak-f form, the proof gate, one- vs two-round trips, the host tail, inline kernel arguments, the upload thread and chunks,
    the overlap placements, every row-MSM form: the queue form (the default) in three shapes, the strip / balanced forms it replaced for
    large commits, the LDS-staged one. One process per setting (tests/switch_worker.py, SPARTAN_OPTIONS) at 2^17 — the smallest size with
    throughput-sized batched rounds — against the oracle's proof of the same instance and tape. The test also checks that no tier-1 option
    of the table is left out of the list."""
    import hashlib, os, subprocess, sys
    s_, seed = 17, 5
    N = 1 << s_
    from spartan_amd import prover as P
    tape = P.seed_scalar(b"tape", seed)
    oi = vp(orc.orc_instance_synthetic(sz(N), sz(N), sz(10), ctypes.c_uint64(seed)))
    og = vp(orc.orc_snark_gens_new(sz(N), sz(N), sz(10), sz(N)))
    oe = vp(orc.orc_snark_encode(oi, og))
    op = vp(orc.orc_snark_prove(oi, og, oe, b"snark_example", tape, None))
    want = hashlib.sha256(oracle_bytes(orc, op)).hexdigest()
    from tests.helpers import options_env
    from spartan_amd import capi
    settings = [{}, {"spark.eq_factor": 0}, {"spark.hash_fuse": 0}, {"ipa.unified_tree": 1}, {"ipa.finish_device": 1},
                {"host.keccak": 1}, {"host.keccak": 2}, {"host.keccak": 3}, {"host.proof_gate": 1},
                {"spark.eq_factor": 0, "spark.hash_fuse": 0, "ipa.unified_tree": 1, "ipa.finish_device": 1},
                {"sumcheck.inline_args": 0}, {"ipa.fused": 0}, {"encode.device": 1}, {"commit.small_device": 1},
                {"sumcheck.double_round_max_len": 0, "sumcheck.host_tail": 0}, {"sumcheck.double_round_max_len": 512},
                {"sumcheck.launch_ahead": 1}, {"sumcheck.launch_ahead": 2}, {"sumcheck.launch_ahead": 2, "sumcheck.host_tail": 0}, {"sumcheck.launch_ahead": 1, "sumcheck.host_tail": 0},
                {"host.pin_thread": 0},
                {"spark.prod_layer2": 0}, {"spark.prod_layer2_max_log2": 14}, {"upload.overlap": 0, "upload.thread": 0}, {"upload.chunks": 2},
                {"overlap.derefs": 0}, {"overlap.eval_ahead": 0}, {"polyeval.eval_from_opening": 0}, {"bg.eighths": 4}, {"bg.eighths": 0},
                {"msm.form": 3}, {"msm.form": 3, "bg.eighths": 3, "upload.chunks": 1},
                {"msm.q_waves": 8, "msm.q_bg_waves": 12, "msm.q_units": 16}, {"msm.q_bg_waves": 4, "msm.q_units": 128, "bg.eighths": 0},
                {"msm.lds_bits": 10, "msm.form": 1}, {"msm.lds_bits": 9, "msm.form": 1, "overlap.derefs": 0}, {"msm.wbits": 11}]
    covered = {k for st in settings for k in st}
    not_proof_shaping = {"ipa.rerun_exceptional", "ipa.dedicated_uploaded", "shard.residue_transport", "shard.cubic_min_len", "host.callstats", "debug.ktime"}  # their own tests (test_gpu_large, test_gpu_shard) / diagnostics
    tier1 = {k for k, _d, _lo, _hi, tier, _doc in capi.options_table() if tier == 1}
    assert tier1 - covered - not_proof_shaping == set(), "tier-1 options without an A/B run: %s" % sorted(tier1 - covered - not_proof_shaping)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for st in settings:
        e = dict(os.environ, SPARTAN_OPTIONS=options_env(**{k.replace(".", "__"): v for k, v in st.items()}))
        r = subprocess.run([sys.executable, os.path.join(root, "tests", "switch_worker.py"), str(s_), str(seed)], env=e, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (st, r.stdout[-2000:], r.stderr[-2000:])
        line = [l for l in r.stdout.splitlines() if l.startswith("PROOF_SHA256")]
        assert line and line[0].split()[1] == want, (st, line)
