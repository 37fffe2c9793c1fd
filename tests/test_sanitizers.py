"""Sanitizer pass on the host side (SURVEY.md section 5 "race detection / sanitizers"; the reference's CI has clippy + tests and no threads to
race, .github/workflows/rust.yml:66-90). CPU part, run by the default suite:
  * the ORACLE rebuilt with AddressSanitizer + UndefinedBehaviorSanitizer (make -C oracle SAN=1) re-runs the oracle's pin and golden tests;
  * the HOST DRIVER (spartan_amd/host -> libspartan_host_asan.so, same flags) re-runs its CPU tests: Merlin / STROBE / Keccak in every form,
    the tdefl restatement on real shape bincodes, bincode writers, host-side field and curve arithmetic, the few-term commitment engine.
Any report (ASan error, UBSan "runtime error") fails the test. The GPU part (ThreadSanitizer on the threaded driver: upload thread, ZK
look-ahead thread, three contexts in flight) is tests/test_gpu_sanitizers.py."""
import os, subprocess, sys
import pytest
from tests.helpers import ROOT

ASAN = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
UBSAN = subprocess.run(["gcc", "-print-file-name=libubsan.so"], capture_output=True, text=True).stdout.strip()
pytestmark = pytest.mark.skipif(not os.path.isabs(ASAN) or not os.path.exists(ASAN), reason="no libasan in this toolchain")


def _run_under_asan(test_files, extra_env, timeout=1500):
    env = dict(os.environ, LD_PRELOAD=ASAN + ":" + UBSAN, ASAN_OPTIONS="detect_leaks=0:verify_asan_link_order=0:abort_on_error=1",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1", OMP_NUM_THREADS="2", **extra_env)
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider"] + [os.path.join(ROOT, "tests", f) for f in test_files],
                       env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    report = r.stdout[-3000:] + r.stderr[-6000:]
    assert "AddressSanitizer" not in report and "runtime error:" not in report, report
    assert r.returncode == 0, report
    return r.stdout


def test_oracle_under_address_and_undefined_behaviour_sanitizers():
    so = os.path.join(ROOT, "oracle", "_san", "liboracle_san.so")
    srcs = [os.path.join(ROOT, "oracle", f) for f in os.listdir(os.path.join(ROOT, "oracle")) if f.endswith((".cc", ".h", ".inc"))]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "SAN=1"], stdout=subprocess.DEVNULL)
    out = _run_under_asan(["test_oracle_pins.py", "test_golden.py"], {"ORACLE_LIB": so})
    assert " passed" in out


def test_host_driver_under_address_and_undefined_behaviour_sanitizers():
    host = os.path.join(ROOT, "spartan_amd", "host")
    outdir = os.path.join(ROOT, "spartan_amd", "lib_san")
    so = os.path.join(outdir, "libspartan_host_asan.so")
    srcs = [os.path.join(host, f) for f in os.listdir(host)]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        os.makedirs(outdir, exist_ok=True)
        ccs = " ".join(sorted(f for f in srcs if f.endswith(".cc")))
        subprocess.check_call(f"g++ -O1 -g -std=c++17 -fPIC -shared -fsanitize=address,undefined -fno-omit-frame-pointer -fno-sanitize-recover=undefined -Wno-unknown-pragmas "
                              f"-D__HIP_PLATFORM_AMD__ -I/opt/rocm/include {ccs} -o {so} -L{ROOT}/spartan_amd/lib -lspartan_hip -L/opt/rocm/lib -lamdhip64 -ldl "
                              f"-Wl,-rpath,{ROOT}/spartan_amd/lib -Wl,-rpath,/opt/rocm/lib", shell=True, cwd=ROOT)
    out = _run_under_asan(["test_host_transcript.py", "test_host_arith.py"], {"SPARTAN_HOST_LIB": so})
    assert " passed" in out
