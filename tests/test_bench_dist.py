"""The N>1 path of bench.py on CPU: two ranks over gloo (world_size 2). Each rank proves its own independent
instance (no data-path collective), so what must be right is rank discovery from the env, the barrier, the MAX
all-reduce of elapsed time and the whole-job aggregation."""
import os, subprocess, sys, socket
from tests.helpers import ROOT

SNIPPET = r"""
import os, sys, json
sys.path.insert(0, %r)
import bench
rank, world, dist = bench.dist_setup(world_n := int(os.environ["WORLD_SIZE"]))
assert world == 2 and dist is not None
bench.dist_barrier(dist)
elapsed = 1.0 + rank            # rank 1 is the slow one
mx = bench.dist_max(dist, elapsed)
total = bench.dist_sum(dist, 1 << 20)
bench.dist_barrier(dist)
print(json.dumps({"rank": rank, "max": mx, "sum": total, "value": world * (1 << 20) / mx}))
dist.destroy_process_group()
"""


def test_two_rank_gloo_aggregation():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   BENCH_DIST_BACKEND="gloo")
        procs.append(subprocess.Popen([sys.executable, "-c", SNIPPET % ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=300)
        assert p.returncode == 0, e[-2000:]
        outs.append(o.strip().splitlines()[-1])
    import json
    res = sorted((json.loads(o) for o in outs), key=lambda d: d["rank"])
    assert [d["rank"] for d in res] == [0, 1]
    for d in res:
        assert d["max"] == 2.0            # MAX over ranks
        assert d["sum"] == 2 * (1 << 20)  # all ranks' units
        assert d["value"] == 2 * (1 << 20) / 2.0


SHARD_SNIPPET = r"""
import os, sys, json
sys.path.insert(0, %r)
import numpy as np
import bench
from spartan_amd import shard
rank, world, dist = bench.dist_setup(int(os.environ["WORLD_SIZE"]))
rows = 64                                   # a 64-row commitment: rank r owns rows [32r, 32r+32)
full = (np.arange(32 * rows, dtype=np.uint32) * 2654435761 >> 7).astype(np.uint8)
buf = np.zeros(32 * rows, dtype=np.uint8)
per = 32 * rows // world
buf[rank * per:(rank + 1) * per] = full[rank * per:(rank + 1) * per]
shard.all_gather_bytes(dist, buf, rank * per, per)
assert (buf == full).all()
# the ctypes callback the C++ driver calls (spz_ctx_set_commit_shard) does the same on a raw pointer
import ctypes
cb = shard.make_gather_callback(dist)
raw = (ctypes.c_uint8 * (32 * rows))()
view = np.ctypeslib.as_array(raw)
view[rank * per:(rank + 1) * per] = full[rank * per:(rank + 1) * per]
assert cb(None, raw, 32 * rows, rank * per, per) == 0 and (view == full).all()
# slices that do not tile the buffer in rank order are refused (callback reports, never raises through C++)
assert cb(None, raw, 32 * rows, 0, per + 1) == -1
bench.dist_barrier(dist)
print(json.dumps({"rank": rank, "ok": True}))
dist.destroy_process_group()
"""


def test_two_rank_gloo_commit_shard_gather():
    """host side of the row-sharded commit: the all-gather of 32-byte commitments over two gloo ranks"""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   BENCH_DIST_BACKEND="gloo")
        procs.append(subprocess.Popen([sys.executable, "-c", SHARD_SNIPPET % ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    for p in procs:
        o, e = p.communicate(timeout=300)
        assert p.returncode == 0, e[-2000:]
        assert '"ok": true' in o
