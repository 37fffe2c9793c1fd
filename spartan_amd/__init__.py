"""spartan_amd — MI355X-native prover hot path for libspartan (microsoft/Spartan).

The product is the C-ABI shared library `spartan_amd/lib/libspartan_hip.so` (include/spartan_hip.h) plus the
host-side prover driver `libspartan_host.so` that mirrors libspartan's SNARK/NIZK API on top of it.
This package is only the ctypes binding used by tests and bench.py; there is no Python or CPU fallback.
"""
from .capi import lib, Ctx, Gens, Table, SpartanHipError  # noqa: F401
