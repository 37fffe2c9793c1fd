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
