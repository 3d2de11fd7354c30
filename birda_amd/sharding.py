"""Multi-GPU sharding of a segment list: one process per GPU, no data-path collective.

Every segment is independent through every stage of the path (reference
`src/pipeline/processor.rs:363-367` treats batch rows independently), so rank g of G owns
the contiguous block [g*N/G, (g+1)*N/G) of the global segment list and the only exchange is
the gather of per-segment results to rank 0.  Over RCCL when the tensors are
on GPUs (backend "nccl"), over gloo in the CPU tests.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block partition; concatenating the shards in rank order restores the list."""
    lo = (n_total * rank) // world
    hi = (n_total * (rank + 1)) // world
    return lo, hi


def assign_by_duration(durations, world: int):
    """Directory mode: contiguous runs of the file list per rank, cut where the cumulative audio duration crosses
    g/world of the total, so output order = file order and every GPU gets about the same audio.
    Returns world lists of file indices (some may be empty)."""
    total = float(sum(durations))
    out = [[] for _ in range(world)]
    if total <= 0.0:
        for i in range(len(durations)):               # no duration hints: equal file counts
            out[(i * world) // len(durations)].append(i)
        return out
    acc = 0.0
    for i, d in enumerate(durations):
        mid = acc + 0.5 * float(d)                    # a file goes where its midpoint falls
        out[min(world - 1, int(mid * world / total))].append(i)
        acc += float(d)
    return out


def gather_results(local: torch.Tensor, n_total: int, rank: int, world: int) -> Optional[torch.Tensor]:
    """Gather per-segment rows ([n_local, ...]) to rank 0 in segment order.  Shards may differ
    by one row, so every rank pads to the largest shard for the collective."""
    if world == 1:
        return local
    sizes = [shard_range(n_total, r, world) for r in range(world)]
    max_n = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((max_n,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    if dist.get_backend() == "nccl":
        out = torch.empty((world * max_n,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, pad)
        if rank != 0:
            return None
        parts = [out[r * max_n: r * max_n + (hi - lo)] for r, (lo, hi) in enumerate(sizes)]
        return torch.cat(parts, 0)
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == 0 else None
    dist.gather(pad, bufs, dst=0)
    if rank != 0:
        return None
    return torch.cat([bufs[r][: hi - lo] for r, (lo, hi) in enumerate(sizes)], 0)
