"""Sharding independent environments across ranks (one process per GPU).

Environments never interact and the model is read-only, so the batch dimension is cut into contiguous
slices, one per rank, and each rank steps its slice with no collective on the data path (SURVEY section 8e;
the reference has no multi-GPU path at all, only per-GPU benchmark processes, benchmarks/conftest.py:28-52).
The only communication offered is an optional gather of the state, off the hot loop.
"""

from __future__ import annotations

import torch
import torch.distributed as dist


def shard_range(total: int, world_size: int, rank: int) -> tuple[int, int]:
    """Contiguous [begin, end) slice of `total` environments owned by `rank` (sizes differ by at most one)."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    base, rem = divmod(total, world_size)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def shard_data(d, world_size: int, rank: int):
    """The rank's slice of a batched Data (leading batch dim)."""
    b, e = shard_range(d.qpos.shape[0], world_size, rank)
    return d[b:e]


def gather_state(d, names=("qpos", "qvel")) -> dict:
    """All-gather selected state leaves from every rank (RCCL over xGMI on GPUs, gloo on CPU).

    Shards may differ in length by one environment, so each rank pads to the longest shard."""
    world = dist.get_world_size()
    out = {}
    n_local = torch.tensor([d.qpos.shape[0]], dtype=torch.int64, device=d.qpos.device)
    counts = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(counts, n_local)
    counts = [int(c.item()) for c in counts]
    nmax = max(counts)
    for n in names:
        t = getattr(d, n)
        pad = torch.zeros((nmax,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        pad[: t.shape[0]] = t
        parts = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(parts, pad)
        out[n] = torch.cat([p[:c] for p, c in zip(parts, counts)], dim=0)
    return out
