"""Row-sharded commitments across the GPUs of one node.

Every rank runs the same proof in lock-step (same instance, same RandomTape seed, same transcript); for each
DensePolynomial::commit (dense_mlpoly.rs:179-204) of L rows, rank r computes rows [r L/W, (r+1) L/W) on its GPU and the
ranks exchange the 32-byte compressed commitments with ONE all-gather of bytes (RCCL over xGMI with the `nccl` backend;
`gloo` in the CPU tests). There is no elliptic-curve reduction to do — rows are independent MSMs over shared generators.
Everything else (sum-checks, IPA, SPARK) is replicated: those steps are latency-bound chains.
"""
import ctypes

import numpy as np

STATS = {"gathers": 0, "bytes": 0}  # exchanges done by this process (bench.py reports them per proof)
GATHER_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint8), ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t)


def all_gather_bytes(dist, buf, off, ln, device="cpu"):
    """buf: writable numpy uint8 array of the full result; this rank's slice [off, off+ln) is filled in. Equal slices
    on every rank (slice r sits at r*ln). On return the whole of buf is filled in."""
    import torch
    world = dist.get_world_size()
    if ln * world != buf.size or off != dist.get_rank() * ln:
        raise ValueError("all_gather_bytes: slices must tile the buffer in rank order")
    mine = torch.from_numpy(buf[off:off + ln].copy()).to(device)
    parts = [torch.empty(ln, dtype=torch.uint8, device=device) for _ in range(world)]
    dist.all_gather(parts, mine)
    buf[:] = torch.cat(parts).cpu().numpy()
    STATS["gathers"] += 1
    STATS["bytes"] += int(buf.size)


def make_gather_callback(dist, device="cpu"):
    """ctypes callback for spz_ctx_set_commit_shard. Keep the returned object alive while the context uses it."""
    def cb(_user, ptr, total, off, ln):
        try:
            buf = np.ctypeslib.as_array(ptr, shape=(total,))
            all_gather_bytes(dist, buf, off, ln, device)
            return 0
        except Exception as e:  # an exception must not unwind through the C++ caller
            import sys
            print(f"commit-shard gather failed: {e!r}", file=sys.stderr)
            return -1
    return GATHER_FN(cb)
