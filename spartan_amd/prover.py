"""ctypes binding of the host-side prover driver (spartan_amd/host, libspartan_host.so), which mirrors libspartan's
SNARK/NIZK API on top of the HIP C ABI. No fallback: raises if either library is missing or no GPU is present."""
import ctypes, os
from .capi import SpartanHipError, LIB_PATH, sz, vp

_HERE = os.path.dirname(os.path.abspath(__file__))
HOST_PATH = os.environ.get("SPARTAN_HOST_LIB") or os.path.join(_HERE, "lib", "libspartan_host.so")  # override: diagnostic builds (e.g. -DSPZ_HOSTPROF)
if not os.path.exists(HOST_PATH):
    raise SpartanHipError(f"{HOST_PATH} not built: run __graft_entry__.build()")
ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
H = ctypes.CDLL(HOST_PATH)
for f in ("spz_ctx_new", "spz_instance_new", "spz_instance_synthetic", "spz_snark_gens_new", "spz_nizk_gens_new", "spz_snark_encode",
          "spz_snark_prove", "spz_nizk_prove", "spz_ctx_raw", "spz_vars_assignment_new", "spz_snark_prove_resident", "spz_nizk_prove_resident",
          "spz_snark_prove_t", "spz_nizk_prove_t"):
    getattr(H, f).restype = vp
for f in ("spz_proof_bytes", "spz_encode_comm", "spz_snark_gens_stream", "spz_merlin_script", "spz_snark_gens_bincode", "spz_commitment_bincode", "spz_decommitment_bincode"):
    getattr(H, f).restype = sz
H.spz_last_error.restype = ctypes.c_char_p
for f in ("spz_ctx_free", "spz_instance_free", "spz_snark_gens_free", "spz_nizk_gens_free", "spz_encode_free", "spz_proof_free", "spz_vars_assignment_free"):
    getattr(H, f).argtypes = [vp]
u64p = ctypes.POINTER(ctypes.c_uint64)
TIME_NAMES = ["polycommit", "prove_sc_phase_one", "prove_sc_phase_two", "polyeval", "R1CSProof::prove", "eval_sparse_polys",
              "commit_nondet_witness", "build_layered_network", "evalproof_layered_network", "total"]


def _chk(h, what):
    if not h:
        raise SpartanHipError(f"{what} failed: {H.spz_last_error().decode()}")
    return vp(h)


class Ctx:
    def __init__(self, device=0):
        self.h = _chk(H.spz_ctx_new(ctypes.c_int(device)), "spz_ctx_new")

    def raw(self):
        """the underlying sp_ctx* (for profiling calls through spartan_amd.capi.lib)"""
        return vp(H.spz_ctx_raw(self.h))

    def set_option(self, key, value):
        """a library option of this context (sp_ctx_set_option; table: spartan_amd/csrc/options.hpp). Tier-1 (A/B / test) options need
        set_option("testing.unlock", 1) first."""
        if H.spz_ctx_set_option(self.h, key.encode(), str(int(value)).encode()) != 0:
            raise SpartanHipError(f"set_option({key}, {value}) refused")

    def set_commit_shard(self, dist, device="cpu"):
        """row-shard every DensePolynomial::commit over the ranks of `dist`, the bytes moved by torch.distributed
        (spartan_amd/shard.py; gloo in the CPU tests); None clears it. All ranks must then run identical prove() calls in lock-step."""
        from . import shard
        if dist is None or dist.get_world_size() == 1:
            H.spz_ctx_set_commit_shard(self.h, ctypes.c_int(0), ctypes.c_int(1), None, None); self._gather = None
            return
        self._gather = shard.make_gather_callback(dist, device)
        if H.spz_ctx_set_commit_shard(self.h, ctypes.c_int(dist.get_rank()), ctypes.c_int(dist.get_world_size()), self._gather, None) != 0:
            raise SpartanHipError(f"set_commit_shard: {H.spz_last_error().decode()}")

    def set_commit_shard_rccl(self, rank, world, unique_id):
        """the same with RCCL inside the library: `unique_id` = the 128 bytes rank 0 got from rccl_unique_id(), handed to
        every rank by the caller (bench.py: one torch.distributed broadcast); ncclAllGather on device buffers per commit"""
        if H.spz_ctx_set_commit_shard_rccl(self.h, ctypes.c_int(rank), ctypes.c_int(world), unique_id) != 0:
            raise SpartanHipError(f"set_commit_shard_rccl: {H.spz_last_error().decode()}")

    def set_commit_shard_virtual(self, nshards):
        """nshards row shards on this one GPU (sub-contexts with their own streams, in-process gather); <= 1 clears it"""
        if H.spz_ctx_set_commit_shard_virtual(self.h, ctypes.c_int(nshards)) != 0:
            raise SpartanHipError(f"set_commit_shard_virtual: {H.spz_last_error().decode()}")

    def shard_stats(self, reset=False):
        out = (ctypes.c_uint64 * 2)()
        H.spz_ctx_shard_stats(self.h, ctypes.c_int(1 if reset else 0), out)
        return {"gathers": int(out[0]), "bytes": int(out[1])}

    def close(self):
        if self.h:
            H.spz_ctx_free(self.h); self.h = None


class Instance:
    def __init__(self, h, num_cons, num_vars, num_inputs, vars_=None, inputs=None):
        self.h, self.num_cons, self.num_vars, self.num_inputs, self.vars, self.inputs = h, num_cons, num_vars, num_inputs, vars_, inputs

    @staticmethod
    def new(ctx, num_cons, num_vars, num_inputs, nnz, rows, cols, vals):
        """Instance::new (lib.rs:121): entries of A, B, C back to back; vals = 32 canonical little-endian bytes each."""
        h = _chk(H.spz_instance_new(ctx.h, sz(num_cons), sz(num_vars), sz(num_inputs), (sz * 3)(*nnz), rows, cols, vals), "Instance::new")
        return Instance(h, num_cons, num_vars, num_inputs)

    @staticmethod
    def produce_synthetic_r1cs(ctx, num_cons, num_vars, num_inputs, seed=0):
        v = (ctypes.c_uint64 * (4 * num_vars))(); i = (ctypes.c_uint64 * (4 * max(num_inputs, 1)))()
        h = _chk(H.spz_instance_synthetic(ctx.h, sz(num_cons), sz(num_vars), sz(num_inputs), ctypes.c_uint64(seed), v, i), "produce_synthetic_r1cs")
        return Instance(h, num_cons, num_vars, num_inputs, v, i)

    def set_digest(self, d):
        H.spz_instance_set_digest(self.h, d, sz(len(d)))

    def digest(self):
        """R1CSShapeDigest (src/r1cs.rs:154-158): the bytes set with set_digest, else computed by the host library (zlib level 6 of bincode(shape))"""
        H.spz_instance_digest.restype = sz
        n = H.spz_instance_digest(self.h, None, sz(0))
        buf = (ctypes.c_uint8 * n)()
        H.spz_instance_digest(self.h, buf, sz(n))
        return bytes(buf)

    def set_digest_header(self, old_header):
        """zlib header variant of the computed digest: False = 0x78 0x9C (miniz >= 2.2, miniz_oxide >= 0.4), True = 0x78 0x01"""
        if H.spz_instance_set_digest_header(self.h, ctypes.c_int(1 if old_header else 0)) != 0:
            raise SpartanHipError("set_digest_header: " + H.spz_last_error().decode())

    def free(self):
        if self.h:
            H.spz_instance_free(self.h); self.h = None


class SNARKGens:
    def __init__(self, ctx, num_cons, num_vars, num_inputs, num_nz_entries):
        self.h = _chk(H.spz_snark_gens_new(ctx.h, sz(num_cons), sz(num_vars), sz(num_inputs), sz(num_nz_entries)), "SNARKGens::new")

    def stream(self, which):
        n = H.spz_snark_gens_stream(self.h, ctypes.c_int(which), None, sz(0))
        b = (ctypes.c_uint8 * n)()
        H.spz_snark_gens_stream(self.h, ctypes.c_int(which), b, sz(n))
        return bytes(b)

    def window_bits(self, which):
        """signed window width of the fixed-base tables of generator stream `which` (0: gens_r1cs_sat, 1: gens_r1cs_eval)"""
        return int(H.spz_snark_gens_window_bits(self.h, ctypes.c_int(which)))

    def windows(self, which):
        """windows per scalar (= mixed additions per committed scalar) of the tables of generator stream `which`"""
        return int(H.spz_snark_gens_windows(self.h, ctypes.c_int(which)))

    def table_bytes(self, which):
        """HBM held by the window tables of generator stream `which`"""
        H.spz_snark_gens_table_bytes.restype = sz
        return int(H.spz_snark_gens_table_bytes(self.h, ctypes.c_int(which)))

    def serialize(self):
        n = H.spz_snark_gens_bincode(self.h, None, sz(0)); b = (ctypes.c_uint8 * n)()
        H.spz_snark_gens_bincode(self.h, b, sz(n))
        return bytes(b)

    def free(self):
        if self.h:
            H.spz_snark_gens_free(self.h); self.h = None


class NIZKGens:
    def __init__(self, ctx, num_cons, num_vars, num_inputs):
        self.h = _chk(H.spz_nizk_gens_new(ctx.h, sz(num_cons), sz(num_vars), sz(num_inputs)), "NIZKGens::new")

    def free(self):
        if self.h:
            H.spz_nizk_gens_free(self.h); self.h = None


def rccl_unique_id():
    """ncclGetUniqueId through the library (rank 0 calls it; the 128 bytes go to every rank)"""
    out = (ctypes.c_uint8 * 128)()
    if H.spz_rccl_unique_id(out) != 0:
        raise SpartanHipError(f"rccl_unique_id: {H.spz_last_error().decode()}")
    return bytes(out)


def keccak_variant():
    """which form of Keccak-f[1600] the host driver picked for this CPU (keccak.cc): plain | bmi2 | avx512"""
    H.spz_keccak_variant.restype = ctypes.c_char_p
    return H.spz_keccak_variant().decode()


def seed_scalar(domain, seed):
    """TEST/BENCH ONLY: the reproducible 64-bit-seed -> scalar map behind the RandomTape seeds of tests/ and bench.py. A real
    proof passes tape_seed=None (the tape is then seeded from OS entropy, like RandomTape::new, random.rs:13-15): a known or
    reused seed fixes every blind of the proof and gives up zero-knowledge."""
    out = (ctypes.c_uint64 * 4)()
    H.spz_seed_scalar(domain, ctypes.c_uint64(seed), out)
    return out


def _proof_bytes(p):
    n = H.spz_proof_bytes(p, None, sz(0))
    b = (ctypes.c_uint8 * n)()
    H.spz_proof_bytes(p, b, sz(n))
    H.spz_proof_free(p)
    return bytes(b)


class SNARK:
    @staticmethod
    def encode(ctx, inst, gens):
        return Encoded(_chk(H.spz_snark_encode(ctx.h, inst.h, gens.h), "SNARK::encode"))

    @staticmethod
    def prove(ctx, inst, enc, vars_, inputs, gens, transcript_label, tape_seed, times=None):
        """vars_: the assignment as Montgomery limbs (a ctypes uint64 array, read in place) or a VarsAssignment (already in HBM)"""
        tm = (ctypes.c_double * 10)()
        if isinstance(vars_, VarsAssignment):
            p = _chk(H.spz_snark_prove_resident(ctx.h, inst.h, gens.h, enc.h, vars_.h, inputs, sz(inst.num_inputs), transcript_label, tape_seed, tm),
                     "SNARK::prove")
        else:
            p = _chk(H.spz_snark_prove(ctx.h, inst.h, gens.h, enc.h, vars_, sz(len(vars_) // 4), inputs, sz(inst.num_inputs), transcript_label,
                                       tape_seed, tm), "SNARK::prove")
        if times is not None:
            times.update(dict(zip(TIME_NAMES, list(tm))))
        return _proof_bytes(p)


    @staticmethod
    def prove_t(ctx, inst, enc, vars_, inputs, gens, transcript_state, tape_seed):
        """SNARK::prove on a caller-owned transcript (lib.rs:339-347): `transcript_state` is a 203-byte ctypes array holding the
        merlin transcript (Strobe128 state, pos, pos_begin, cur_flags); it is continued by the proof and left in the state the
        proof ends in. Returns the proof bytes."""
        tm = (ctypes.c_double * 10)()
        res = isinstance(vars_, VarsAssignment)
        p = _chk(H.spz_snark_prove_t(ctx.h, inst.h, gens.h, enc.h, vars_.h if res else None, None if res else vars_, sz(0 if res else len(vars_) // 4),
                                     inputs, sz(inst.num_inputs), transcript_state, tape_seed, tm), "SNARK::prove")
        return _proof_bytes(p)


class NIZK:
    @staticmethod
    def prove_t(ctx, inst, vars_, inputs, gens, transcript_state, tape_seed):
        """NIZK::prove on a caller-owned transcript (lib.rs:501-509); see SNARK.prove_t"""
        tm = (ctypes.c_double * 10)()
        res = isinstance(vars_, VarsAssignment)
        p = _chk(H.spz_nizk_prove_t(ctx.h, inst.h, gens.h, vars_.h if res else None, None if res else vars_, sz(0 if res else len(vars_) // 4),
                                    inputs, sz(inst.num_inputs), transcript_state, tape_seed, tm), "NIZK::prove")
        return _proof_bytes(p)

    @staticmethod
    def prove(ctx, inst, vars_, inputs, gens, transcript_label, tape_seed, times=None):
        tm = (ctypes.c_double * 10)()
        if isinstance(vars_, VarsAssignment):
            p = _chk(H.spz_nizk_prove_resident(ctx.h, inst.h, gens.h, vars_.h, inputs, sz(inst.num_inputs), transcript_label, tape_seed, tm), "NIZK::prove")
        else:
            p = _chk(H.spz_nizk_prove(ctx.h, inst.h, gens.h, vars_, sz(len(vars_) // 4), inputs, sz(inst.num_inputs), transcript_label, tape_seed, tm),
                     "NIZK::prove")
        if times is not None:
            times.update(dict(zip(TIME_NAMES, list(tm))))
        return _proof_bytes(p)


class VarsAssignment:
    """VarsAssignment::new (lib.rs:56-105) with the scalars uploaded once: proofs over it start from the copy in HBM"""
    def __init__(self, ctx, vars_):
        self.n = len(vars_) // 4
        self.h = _chk(H.spz_vars_assignment_new(ctx.h, vars_, sz(self.n)), "VarsAssignment::new")

    def free(self):
        if self.h:
            H.spz_vars_assignment_free(self.h); self.h = None


class Encoded:
    def __init__(self, h):
        self.h = h

    def comm(self, which):
        n = H.spz_encode_comm(self.h, ctypes.c_int(which), None, sz(0))
        b = (ctypes.c_uint8 * (32 * n))()
        H.spz_encode_comm(self.h, ctypes.c_int(which), b, sz(32 * n))
        return bytes(b)

    def serialize_commitment(self):
        n = H.spz_commitment_bincode(self.h, None, sz(0)); b = (ctypes.c_uint8 * n)()
        H.spz_commitment_bincode(self.h, b, sz(n))
        return bytes(b)

    def serialize_decommitment(self):
        n = H.spz_decommitment_bincode(self.h, None, sz(0)); b = (ctypes.c_uint8 * n)()
        H.spz_decommitment_bincode(self.h, b, sz(n))
        return bytes(b)

    def free(self):
        if self.h:
            H.spz_encode_free(self.h); self.h = None
