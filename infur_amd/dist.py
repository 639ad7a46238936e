"""Frame-sharded multi-GPU host layer: one process per GPU over ``torch.distributed``.

The path has no cross-frame dependence (infur/src/app.rs:107-153 keeps only buffer-reuse
state; ``Model::is_dirty`` is constant false, infur/src/predict_onnx.rs:336-338), so frames
shard across ranks with NO steady-state collective.  The only exchange step is the one-off
replication of the weight blob at model load: rank 0 holds (or reads) the blob and
broadcasts it -- RCCL over xGMI on GPUs (backend "nccl"), gloo in the CPU tests -- then
every rank repacks it locally (``infur_model_load_blob_dev``).
"""
from __future__ import annotations

import os
from typing import List, Optional, Tuple

import numpy as np


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) slice of ``n_items`` frames owned by ``rank``; sizes differ by at most 1."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    q, r = divmod(n_items, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def owner_of(frame_id: int, n_items: int, world: int) -> int:
    """Rank that owns ``frame_id`` under ``shard_range`` (results are written in frame-id order)."""
    for r in range(world):
        lo, hi = shard_range(n_items, r, world)
        if lo <= frame_id < hi:
            return r
    raise ValueError(frame_id)


def round_robin_owner(frame_id: int, world: int) -> int:
    """Streaming mode: frame i goes to rank i % world (bounded reorder window on the consumer)."""
    return frame_id % world


def broadcast_blob(blob: Optional[bytes], device=None, src: int = 0):
    """Replicate the weight blob from ``src`` to every rank.  Returns a uint8 tensor on
    ``device`` (cuda for RCCL, cpu for gloo).  Without an initialised process group the
    blob is simply moved to ``device``."""
    import torch
    import torch.distributed as dist

    dev = torch.device(device) if device is not None else torch.device("cpu")
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        assert blob is not None
        return torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(dev)
    rank = dist.get_rank()
    n = torch.tensor([len(blob) if rank == src else 0], dtype=torch.int64, device=dev)
    dist.broadcast(n, src=src)
    if rank == src:
        t = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(dev)
    else:
        t = torch.empty(int(n.item()), dtype=torch.uint8, device=dev)
    dist.broadcast(t, src=src)
    return t


def load_model_everywhere(ctx, blob: Optional[bytes], src: int = 0, coll_device: Optional[str] = None):
    """Broadcast + per-rank load.  ``ctx``: infur_amd.processors.Context on this rank's GPU.
    ``coll_device``: where the collective runs (default: this rank's GPU, i.e. RCCL).
    -> (blob bytes, milliseconds spent in the collective alone; 0.0 without a process group)."""
    import time

    import torch
    import torch.distributed as dist

    gpu = f"cuda:{ctx.device}"
    # a one-rank group counts when INFUR_BENCH_FORCE_COLLECTIVES=1: the collective calls then execute (on RCCL for
    # backend nccl) although there is nobody to talk to -- how the nccl branch is exercised on a 1-GPU box
    forced = os.environ.get("INFUR_BENCH_FORCE_COLLECTIVES") == "1"
    multi = dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or forced)
    bcast_ms = 0.0
    if not multi:
        t = broadcast_blob(blob, device=gpu, src=src)
    else:
        dev = torch.device(coll_device or gpu)
        rank = dist.get_rank()
        n = torch.tensor([len(blob) if rank == src else 0], dtype=torch.int64, device=dev)
        dist.broadcast(n, src=src)
        # the payload is staged where the collective runs BEFORE the clock starts: bcast_ms is the broadcast alone
        if rank == src:
            t = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(dev)
        else:
            t = torch.empty(int(n.item()), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        dist.broadcast(t, src=src)
        torch.cuda.synchronize()
        bcast_ms = (time.perf_counter() - t0) * 1e3
    t = t.to(gpu)
    torch.cuda.synchronize()
    ctx.check(ctx.L.infur_model_load_blob_dev(ctx.h, t.data_ptr(), t.numel()))
    return t.numel(), bcast_ms


def gather_masks(local: List[np.ndarray], n_items: int):
    """Collect per-rank result lists on rank 0 in frame-id order (test/utility path, not timed)."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return list(local)
    world, rank = dist.get_world_size(), dist.get_rank()
    out = [None] * world
    dist.all_gather_object(out, local)
    if rank != 0:
        return None
    merged = []
    for r in range(world):
        lo, hi = shard_range(n_items, r, world)
        assert len(out[r]) == hi - lo
        merged.extend(out[r])
    return merged
