"""The N>1 host layer on CPU: world_size-2 gloo processes exercise the weight-blob broadcast
and the frame sharding (the data path itself has no collective)."""
import hashlib
import os
import socket

import numpy as np
import pytest

from infur_amd import dist as idist


def test_shard_range_partitions_everything():
    for n in (0, 1, 7, 8, 64, 65, 300):
        for world in (1, 2, 3, 4, 8):
            seen = []
            for r in range(world):
                lo, hi = idist.shard_range(n, r, world)
                assert 0 <= lo <= hi <= n and hi - lo in (n // world, n // world + 1)
                seen.extend(range(lo, hi))
            assert seen == list(range(n))
            for f in range(n):
                lo, hi = idist.shard_range(n, idist.owner_of(f, n, world), world)
                assert lo <= f < hi
    assert idist.shard_range(64, 3, 8) == (24, 32)  # BASELINE configs[3]: 64 frames / 8 GPUs = 8 each
    assert [idist.round_robin_owner(i, 4) for i in range(6)] == [0, 1, 2, 3, 0, 1]
    with pytest.raises(ValueError):
        idist.shard_range(4, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_frames, q):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # rank 0 owns the blob; everyone must end up with identical bytes
        blob = bytes(np.random.default_rng(1).integers(0, 256, 100_003, dtype=np.uint8)) if rank == 0 else None
        t = idist.broadcast_blob(blob, device="cpu")
        sha = hashlib.sha256(t.numpy().tobytes()).hexdigest()
        # frame sharding: each rank "processes" its contiguous range (stand-in result = frame id plane)
        lo, hi = idist.shard_range(n_frames, rank, world)
        local = [np.full((2, 2), i, np.int32) for i in range(lo, hi)]
        merged = idist.gather_masks(local, n_frames)
        ok = merged is None or [int(m[0, 0]) for m in merged] == list(range(n_frames))
        q.put((rank, sha, t.numel(), lo, hi, ok))
    finally:
        dist.destroy_process_group()


def test_broadcast_and_sharding_world2():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 7, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, sha0, n0, lo0, hi0, ok0), (r1, sha1, n1, lo1, hi1, ok1) = res
    assert sha0 == sha1 and n0 == n1 == 100_003
    assert (lo0, hi0, lo1, hi1) == (0, 4, 4, 7) and ok0 and ok1


def test_single_process_broadcast_is_identity():
    t = idist.broadcast_blob(b"abc123", device="cpu")
    assert bytes(t.numpy().tobytes()) == b"abc123"
