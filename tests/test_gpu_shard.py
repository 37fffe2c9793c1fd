"""Row-sharded commitments end to end: two lock-step ranks share the single GPU of the test box and
exchange commitment bytes over gloo (RCCL refuses two ranks on one device; on an 8-GPU node the same code runs with the
`nccl` backend). bench.py --shard-commits itself compares every sharded proof with the unsharded bytes."""
import json, os, socket, subprocess, sys

import pytest

from tests.helpers import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("s", [10, 14])
def test_sharded_commit_proof_is_byte_identical(s):
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   BENCH_DIST_BACKEND="gloo", BENCH_FORCE_DEVICE="0", BENCH_NO_PROF="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--shard-commits", "--log2-cons", str(s),
                                       "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--concurrent", "0"],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=600)
        assert p.returncode == 0, e[-3000:]
        outs.append(o)
    line = json.loads([l for l in outs[0].splitlines() if l.startswith("{")][-1])
    assert line["scaling"] == "strong" and line["n_gpus"] == 2
    assert "sharded over 2 GPUs" in line["config"]["parallelism"]
    # witness commit (r1csproof.rs:159) + derefs commit (sparse_mlpoly.rs:64-67): 2^(s/2) and 2^((s+3)/2) rows of 32 bytes
    assert line["config"]["all_gathers_per_proof"] == 2
    assert line["config"]["all_gather_bytes_per_proof"] == 32 * ((1 << (s // 2)) + (1 << ((s + 3) // 2)))
    assert line["value"] == pytest.approx((1 << s) / (line["ms_per_step"] * 1e-3), rel=1e-6)  # one proof, not two
