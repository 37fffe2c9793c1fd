"""Test-side helpers: ctypes loaders for the ORACLE (oracle/liboracle.so), the host-compiled arithmetic
check shim (tests/csrc/libhostcheck.so) and Python big-int ground truth. Test infrastructure only."""
import ctypes, os, subprocess, random

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
Q = 2**252 + 27742317777372353535851937790883648493
P = 2**255 - 19
R = 2**256 % Q
RINV = pow(R, Q - 2, Q)
u64x4 = ctypes.c_uint64 * 4
vp = ctypes.c_void_p
sz = ctypes.c_size_t



def options_env(**kv):
    """SPARTAN_OPTIONS value (the library's one environment hook, spartan_amd/csrc/options.hpp) for a worker process: keys are written
    with `__` for the dot (msm__flat=0 -> msm.flat=0); A/B / test options are unlocked first"""
    return ",".join(["testing.unlock=1"] + ["%s=%d" % (k.replace("__", "."), int(v)) for k, v in kv.items()])

def _build_if_missing(path, cmd, cwd=ROOT):
    if not os.path.exists(path):
        subprocess.check_call(cmd, cwd=cwd, shell=True)


def load_oracle():
    so = os.environ.get("ORACLE_LIB") or os.path.join(ROOT, "oracle", "liboracle.so")  # the override: the sanitizer build (tests/test_sanitizers.py)
    _build_if_missing(so, "make -C oracle")
    L = ctypes.CDLL(so)
    for f in ("orc_instance_synthetic", "orc_instance_new", "orc_instance_new_padded", "orc_snark_gens_new", "orc_nizk_gens_new", "orc_snark_encode",
              "orc_snark_prove", "orc_nizk_prove"):
        getattr(L, f).restype = vp
    for f in ("orc_proof_bytes", "orc_instance_nnz", "orc_instance_shape_bincode", "orc_encode_comm", "orc_merlin_script", "orc_snark_gens_bincode", "orc_commitment_bincode", "orc_decommitment_bincode"):
        getattr(L, f).restype = sz
    return L


def load_hostcheck():
    so = os.path.join(ROOT, "tests", "csrc", "libhostcheck.so")
    src = os.path.join(ROOT, "tests", "csrc", "hostcheck.cc")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call("g++ -O2 -std=c++17 -fPIC -shared -Wno-unknown-pragmas tests/csrc/hostcheck.cc -o tests/csrc/libhostcheck.so",
                              cwd=ROOT, shell=True)
    return ctypes.CDLL(so)


# ---- scalar helpers: python int (canonical value) <-> Montgomery limbs
def to_mont_limbs(x):
    m = (x % Q) * R % Q
    return u64x4(*[(m >> (64 * i)) & (2**64 - 1) for i in range(4)])


def from_mont_limbs(l):
    m = sum(int(l[i]) << (64 * i) for i in range(4))
    assert m < Q, "limbs not fully reduced"
    return m * RINV % Q


def mont_array(vals):
    arr = (ctypes.c_uint64 * (4 * len(vals)))()
    for k, x in enumerate(vals):
        m = (x % Q) * R % Q
        for i in range(4):
            arr[4 * k + i] = (m >> (64 * i)) & (2**64 - 1)
    return arr


def from_mont_array(arr, n):
    out = []
    for k in range(n):
        m = sum(int(arr[4 * k + i]) << (64 * i) for i in range(4))
        assert m < Q
        out.append(m * RINV % Q)
    return out


def rand_scalars(rng, n, kind="uniform"):
    if kind == "uniform":
        return [rng.randrange(Q) for _ in range(n)]
    if kind == "small":
        return [rng.randrange(1 << 20) for _ in range(n)]
    if kind == "sparse":
        return [rng.randrange(Q) if rng.random() < 0.5 else 0 for _ in range(n)]
    if kind == "edge":
        pool = [0, 1, Q - 1, 2**252, 2**252 - 1, 127, 128, 129, 255, 256, (Q - 1) // 2, 0x80 * sum(256**i for i in range(31))]
        return [pool[rng.randrange(len(pool))] for _ in range(n)]
    raise ValueError(kind)


def gens_bytes(orc, n, label=b"gens_r1cs_sat"):
    """compressed generators of MultiCommitGens::new(n, label): n points G then h (oracle)."""
    buf = (ctypes.c_uint8 * (32 * (n + 1)))()
    orc.orc_multi_commit_gens(sz(n), label, buf)
    return bytes(buf)


# ---- bulk conversions for the large-shape tests (2^17..2^20 scalars): one mul-mod + to_bytes per element
def mont_bulk(vals):
    """python ints -> ctypes uint64 array of Montgomery limbs (same layout as mont_array, ~10x faster)"""
    raw = b"".join(((x % Q) * R % Q).to_bytes(32, "little") for x in vals)
    return (ctypes.c_uint64 * (4 * len(vals))).from_buffer_copy(raw)


def from_mont_bulk(arr, n):
    raw = bytes(arr)[:32 * n]
    out = []
    for k in range(n):
        m = int.from_bytes(raw[32 * k:32 * k + 32], "little")
        assert m < Q
        out.append(m * RINV % Q)
    return out


def fast_scalars(rng, n):
    """n uniform scalars from one getrandbits call per element"""
    return [rng.getrandbits(300) % Q for _ in range(n)]


_MINIZ = None


def real_miniz_zlib(data, level=6):
    """zlib stream of `data` from the REAL C miniz (3.0.2, mz_version "11.0.2") that libtorch_cpu.so bundles and exports (mz_compress2):
    the independent implementation the in-tree deflater (spartan_amd/host/deflate.cc) and the NIZK golden digests are pinned against.
    flate2's rust_backend runs miniz_oxide, the Rust port of this code; level 6 is Compression::default() (src/r1cs.rs:154-158)."""
    global _MINIZ
    if _MINIZ is None:
        import torch
        mz = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libtorch_cpu.so"))
        mz.mz_compress2.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_ulong), ctypes.c_char_p, ctypes.c_ulong, ctypes.c_int]
        mz.mz_compress2.restype = ctypes.c_int
        mz.mz_compressBound.restype = ctypes.c_ulong
        mz.mz_compressBound.argtypes = [ctypes.c_ulong]
        mz.mz_version.restype = ctypes.c_char_p
        _MINIZ = mz
    mz = _MINIZ
    cap = mz.mz_compressBound(len(data)) + 64
    out = ctypes.create_string_buffer(cap)
    n = ctypes.c_ulong(cap)
    rc = mz.mz_compress2(out, ctypes.byref(n), bytes(data), len(data), level)
    assert rc == 0, "mz_compress2 failed: %d" % rc
    return out.raw[:n.value]


def oracle_shape_bincode(orc, inst):
    n = orc.orc_instance_shape_bincode(inst, None, sz(0))
    b = (ctypes.c_uint8 * n)()
    orc.orc_instance_shape_bincode(inst, b, sz(n))
    return bytes(b)
