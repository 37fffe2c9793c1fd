"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/spartan_hip.h declares; without a GPU it refuses to create a context (no CPU fallback)."""
import ctypes, os, re
import pytest
from tests.helpers import ROOT


def header_symbols():
    src = open(os.path.join(ROOT, "include", "spartan_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sp_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from spartan_amd import capi
    syms = header_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(capi.lib, s), f"{s} declared in spartan_hip.h but not exported"
    assert set(capi.SYMBOLS) == set(syms)


def test_no_cpu_fallback_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from spartan_amd import capi
    h = ctypes.c_void_p()
    rc = capi.lib.sp_ctx_create(ctypes.c_int(0), ctypes.byref(h))
    assert rc == -3 and not h  # SP_EHIP: fails loudly, never computes on the CPU
    assert capi.lib.sp_strerror(rc).decode().startswith("HIP runtime error")


def _header_protos():
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_gpu_rs", os.path.join(ROOT, "rust_shim", "gen_gpu_rs.py"))
    gen = importlib.util.module_from_spec(spec); spec.loader.exec_module(gen)
    return gen, gen.parse_header()


def test_rust_binding_matches_header():
    """rust_shim/src/gpu.rs (the `extern "C"` block libspartan's `gpu` feature links against) declares every entry point of
    include/spartan_hip.h with the same name, arity, pointer depth and constness — and nothing else. No Rust toolchain
    exists here, so this is the mechanical check the binding gets."""
    import re
    gen, protos = _header_protos()
    src = open(os.path.join(ROOT, "rust_shim", "src", "gpu.rs")).read()
    block = src[src.index('extern "C" {'):]
    block = block[:block.index("\n}")]
    rust = {}
    for m in re.finditer(r"pub fn (sp_\w+)\((.*?)\)(?: -> ([^;]+))?;", block):
        args = [a.strip() for a in m.group(2).split(",") if a.strip()]
        rust[m.group(1)] = ([a.split(":", 1)[1].strip() for a in args], (m.group(3) or "").strip())
    assert sorted(rust) == sorted(p[0] for p in protos)
    from spartan_amd import capi
    assert sorted(rust) == sorted(capi.SYMBOLS)          # ... which are the symbols the shared library exports (test above)
    for name, ret, params in protos:
        rtypes, rret = rust[name]
        assert len(rtypes) == len(params), name
        for (ctype, _), rt in zip(params, rtypes):
            assert rt == gen.rust_type(ctype), (name, ctype, rt)
            assert rt.count("*") == ctype.count("*"), (name, ctype, rt)             # pointer depth
            if ctype.startswith("const ") and "*" in ctype:
                assert "*const" in rt, (name, ctype, rt)                              # constness of the pointee
        assert rret == {"int32_t": "i32", "uint64_t": "u64", "size_t": "usize", "int": "c_int", "const char*": "*const c_char", "void": ""}[ret], name
    # the generator is idempotent: the committed file is what it emits today
    committed = src
    gen.emit()
    assert open(os.path.join(ROOT, "rust_shim", "src", "gpu.rs")).read() == committed


def test_rust_seams_call_only_declared_symbols():
    """every `gpu::sp_*(...)` call in rust_shim/seams/*.rs and in the hand-written tail of gpu.rs names an entry point of the
    header and passes as many arguments as the C prototype takes"""
    import re, glob
    _, protos = _header_protos()
    arity = {p[0]: len(p[2]) for p in protos}
    files = glob.glob(os.path.join(ROOT, "rust_shim", "seams", "*.rs")) + [os.path.join(ROOT, "rust_shim", "src", "gpu_tail.rs.in")]
    ncalls = 0
    for f in files:
        src = re.sub(r"//[^\n]*", "", open(f).read())
        for m in re.finditer(r"\b(sp_\w+)\s*\(", src):
            name = m.group(1)
            if name not in arity:
                continue   # a Rust helper, not an FFI call
            depth, i, nargs, cur = 1, m.end(), 0, ""
            while depth:
                ch = src[i]
                if ch in "([{":
                    depth += 1
                elif ch in ")]}":
                    depth -= 1
                if depth == 1 and ch == ",":
                    nargs += 1; cur = ""
                elif depth >= 1:
                    cur += ch
                i += 1
            nargs += 1 if cur.strip() else 0
            assert nargs == arity[name], (os.path.basename(f), name, nargs, arity[name])
            ncalls += 1
    assert ncalls >= 20


def test_rust_seams_call_what_the_driver_calls():
    """The measured path must be the drop-in's path: the set of C-ABI entry points the C++ host driver calls (prover.cc + spark.inc + shard.cc,
    every switch included) equals the set the Rust seam bodies call (rust_shim/seams/*.rs + the hand-written tail of gpu.rs). An entry
    point the driver needs and no seam calls would mean the Rust crate cannot run the path that bench.py times."""
    import re, glob
    _, protos = _header_protos()
    declared = {p[0] for p in protos}

    def calls(files):
        found = set()
        for f in files:
            src = open(f).read()
            src = re.sub(r"//[^\n]*", "", src)
            found |= {m for m in re.findall(r"\b(sp_\w+)\s*\(", src) if m in declared}
        return found
    host = os.path.join(ROOT, "spartan_amd", "host")
    driver = calls([os.path.join(host, "prover.cc"), os.path.join(host, "spark.inc"), os.path.join(host, "shard.cc"), os.path.join(host, "libspartan.hpp")])
    seams = calls(glob.glob(os.path.join(ROOT, "rust_shim", "seams", "*.rs")) + [os.path.join(ROOT, "rust_shim", "src", "gpu_tail.rs.in")])
    assert len(driver) >= 55
    assert driver - seams == set(), f"called by the C++ driver, by no Rust seam: {sorted(driver - seams)}"
    # what only the Rust side touches: helpers of its own handle types (the C++ side keeps its shard contexts in shard.cc)
    assert seams - driver <= {"sp_gens_upload", "sp_table_download", "sp_ctx_device"}, f"called by a seam, never by the driver: {sorted(seams - driver)}"


def test_option_table_and_tiers():
    """The library's tunables are named options (spartan_amd/csrc/options.hpp, sp_ctx_set_option), not environment variables: unknown keys and
    out-of-range values are refused, A/B / test options (tier 1) are refused until testing.unlock is set, every option has a documented
    default inside its range. (Process-wide defaults: no GPU needed.)"""
    import subprocess, sys
    from spartan_amd import capi
    table = capi.options_table()
    assert 30 <= len(table) <= 44 and len({k for k, *_ in table}) == len(table)   # (51 at the start of round 6: the A/B switches of paths that lost every comparison since round 3 went with their kernels)
    for key, default, lo, hi, tier, doc in table:
        assert lo <= default <= hi and tier in (0, 1) and len(doc) > 10, key
    code = r"""
import ctypes, sys
sys.path.insert(0, %r)
from spartan_amd import capi
L = capi.lib
v = ctypes.c_int64()
assert L.sp_ctx_set_option(None, b"no.such.option", b"1") == -1
assert L.sp_ctx_set_option(None, b"bg.eighths", b"9") == -1 and L.sp_ctx_set_option(None, b"bg.eighths", b"x") == -1
assert L.sp_ctx_set_option(None, b"spark.eq_factor", b"0") == -1          # tier 1: locked
assert L.sp_ctx_get_option(None, b"spark.eq_factor", ctypes.byref(v)) == 0 and v.value == 1
assert L.sp_ctx_set_option(None, b"testing.unlock", b"1") == 0 and L.sp_ctx_set_option(None, b"spark.eq_factor", b"0") == 0
assert L.sp_ctx_get_option(None, b"spark.eq_factor", ctypes.byref(v)) == 0 and v.value == 0
assert L.sp_ctx_get_option(None, b"bg.eighths", ctypes.byref(v)) == 0 and v.value == 6      # from SPARTAN_OPTIONS
# values inside the range that no kernel supports are refused, not silently accepted (ADVICE r5: msm.lds_bits = 5 gave wrong commitments)
for bad in (b"1", b"4", b"5"):
    assert L.sp_ctx_set_option(None, b"msm.lds_bits", bad) == -1
assert L.sp_ctx_set_option(None, b"msm.lds_bits", b"6") == 0 and L.sp_ctx_set_option(None, b"msm.lds_bits", b"0") == 0
assert L.sp_ctx_set_option(None, b"msm.q_waves", b"5") == -1 and L.sp_ctx_set_option(None, b"msm.q_waves", b"8") == 0
print("OPTIONS_OK")
""" % ROOT
    env = dict(os.environ, SPARTAN_OPTIONS="bg.eighths=6")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "OPTIONS_OK" in r.stdout, r.stderr[-2000:]
    # a misspelt or locked entry in SPARTAN_OPTIONS must not silently measure the default: the process stops
    for bad in ("bg.eigths=6", "spark.eq_factor=0", "bg.eighths=99"):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, SPARTAN_OPTIONS=bad), capture_output=True, text=True, timeout=120)
        assert r.returncode != 0 and "refused" in r.stderr, (bad, r.stderr[-500:])


def test_no_environment_switches_in_the_product_sources():
    """one getenv in the library (SPARTAN_OPTIONS, options.hip) and none in the host driver"""
    import glob, re
    hits = []
    for f in glob.glob(os.path.join(ROOT, "spartan_amd", "csrc", "*")) + glob.glob(os.path.join(ROOT, "spartan_amd", "host", "*")):
        if os.path.isfile(f):
            for m in re.finditer(r'getenv\("(\w+)"\)', open(f, errors="ignore").read()):
                hits.append((os.path.basename(f), m.group(1)))
    assert hits == [("options.hip", "SPARTAN_OPTIONS")], hits
