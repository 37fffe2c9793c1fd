"""ThreadSanitizer on the threaded host driver (VERDICT r4 missing #4): the proving thread spins on a host-mapped flag
while a ZK look-ahead thread and an upload thread run, and several contexts share the process-wide generator-table cache. The host code of
BOTH libraries is rebuilt with -fsanitize=thread (scripts/build_tsan.sh; device code untouched) and tests/tsan_worker.py proves SNARK and
NIZK at 2^12 from one thread and then from three threads at once. A data race reported in this repo's code fails the test; the report
is kept as gpurun_out/tsan_report.txt (copied to profiles/ by hand)."""
import glob, os, re, subprocess, sys
import pytest
from tests.helpers import ROOT

pytestmark = pytest.mark.gpu


def test_threaded_host_driver_under_thread_sanitizer():
    rt = glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.tsan-x86_64.so")
    if not rt:
        pytest.skip("no ThreadSanitizer runtime in this toolchain")
    lib = os.path.join(ROOT, "spartan_amd", "lib")
    built = [os.path.join(lib, "libspartan_hip_tsan.so"), os.path.join(lib, "libspartan_host_tsan.so")]
    srcs = glob.glob(os.path.join(ROOT, "spartan_amd", "csrc", "*")) + glob.glob(os.path.join(ROOT, "spartan_amd", "host", "*")) + [os.path.join(ROOT, "include", "spartan_hip.h")]
    if not all(os.path.exists(b) for b in built) or max(os.path.getmtime(x) for x in srcs) > min(os.path.getmtime(b) for b in built):
        subprocess.check_call(["bash", os.path.join(ROOT, "scripts", "build_tsan.sh")], stdout=subprocess.DEVNULL)   # the sanitizer build follows the sources
    env = dict(os.environ, LD_PRELOAD=rt[0], SPARTAN_HIP_LIB=os.path.join(lib, "libspartan_hip_tsan.so"), SPARTAN_HOST_LIB=os.path.join(lib, "libspartan_host_tsan.so"),
               TSAN_OPTIONS="halt_on_error=0:report_signal_unsafe=0:history_size=4:second_deadlock_stack=1:exitcode=0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tsan_worker.py"), "12"], env=env, capture_output=True, text=True, timeout=900)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    open(os.path.join(ROOT, "gpurun_out", "tsan_report.txt"), "w").write(r.stdout[-4000:] + "\n---- stderr ----\n" + r.stderr[-60000:])
    if "unexpected memory mapping" in r.stderr or "failed to intercept" in r.stderr:
        pytest.skip("ThreadSanitizer cannot share a process with the HIP runtime on this box: " + r.stderr[-300:])
    assert "TSAN_WORKER_OK" in r.stdout, r.stderr[-3000:]
    # A report is OURS when the innermost frame of one of its two accesses — below the sanitizer's own interceptor (malloc, free, memcpy,
    # pthread_*) — lies in this repo's libraries. What remains are accesses whose both ends are inside libamdhip64 / libhsa-runtime64 (the
    # HIP runtime's worker thread against the calling thread: allocations and locks of its own, synchronised by means ThreadSanitizer cannot
    # see in an uninstrumented library): profiles/r5_tsan_report.txt lists them.
    def innermost(block):
        for line in block:
            m = re.match(r"\s+#\d+ .*\((\S+?)\+0x[0-9a-f]+\)", line)
            if m and "libclang_rt.tsan" not in m.group(1):
                return m.group(1)
        return ""
    ours = []
    for rep in re.split(r"={18}\n", r.stderr):
        if "WARNING: ThreadSanitizer: data race" not in rep:
            continue
        lines = rep.split("\n")
        heads = [i for i, l in enumerate(lines) if re.match(r"\s+(Write|Read|Atomic|Previous)", l)]
        if any("libspartan_" in innermost(lines[i + 1:i + 8]) for i in heads):
            ours.append(rep)
    assert not ours, "data races in the host driver:\n" + "\n".join(ours[:3])[:6000]
