"""Sharding of a segment list over the GPUs of a node: the partition rules and the result rows, no framework.

Every segment is independent through every stage of the path (reference `src/pipeline/processor.rs:363-367` treats
batch rows independently), so shard g of G owns a contiguous block of the global list and the only exchange is the
gather of per-segment results to rank 0.  The same rules are exported by the library for a
single-process host (`bh_shard_range`, `bh_shard_ranges_weighted`, `bh_multi_*`, include/birda_hip.h); this module is
their Python mirror for the one-process-per-GPU launch (`bench.py` under torch.distributed.run), which brings its own
collective: `gather_rows` takes the exchange step as a callable, so nothing here imports a tensor library.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np


def shard_range(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block partition; concatenating the shards in rank order restores the list (bh_shard_range)."""
    lo = (n_total * rank) // world
    hi = (n_total * (rank + 1)) // world
    return lo, hi


def shard_ranges_weighted(weights: Sequence[int], world: int) -> List[int]:
    """world + 1 cut points of a contiguous partition balanced by weight (bh_shard_ranges_weighted): item i goes to the
    shard its midpoint falls in on the cumulative-weight axis.  For mixed-rate lists (BASELINE config 5) the weight is
    the segment's SOURCE sample count."""
    from . import _lib
    w = np.ascontiguousarray(weights, np.uint64)
    bounds = np.zeros(world + 1, np.uint64)
    rc = _lib.load().bh_shard_ranges_weighted(w.ctypes.data, len(w), world, bounds.ctypes.data)
    if rc != 0:
        raise ValueError("shard_ranges_weighted: bad arguments")
    return [int(b) for b in bounds]


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


def pack_topk(index: np.ndarray, confidence: np.ndarray) -> np.ndarray:
    """[n, k] int32 indices + [n, k] f32 confidences -> [n, 2k] int32 rows (the confidences' bit patterns): what travels
    through the gather.  Indices stay integers at any class count."""
    index = np.ascontiguousarray(index, np.int32)
    confidence = np.ascontiguousarray(confidence, np.float32)
    return np.concatenate([index, confidence.view(np.int32)], axis=1)


def unpack_topk(rows: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    rows = np.ascontiguousarray(rows, np.int32)
    k = rows.shape[1] // 2
    return rows[:, :k].copy(), rows[:, k:].copy().view(np.float32)


def gather_rows(local: np.ndarray, n_total: int, rank: int, world: int,
                exchange: Callable[[np.ndarray], Optional[List[np.ndarray]]]) -> Optional[np.ndarray]:
    """Gather per-segment rows ([n_local, ...]) to rank 0 in segment order.  Shards may differ by one row, so every
    rank pads to the largest shard; `exchange(padded)` is the collective (all ranks call it; it returns the world
    padded blocks in rank order on rank 0, anything elsewhere)."""
    if world == 1:
        return local
    sizes = [shard_range(n_total, r, world) for r in range(world)]
    max_n = max(hi - lo for lo, hi in sizes)
    pad = np.zeros((max_n,) + tuple(local.shape[1:]), local.dtype)
    pad[: local.shape[0]] = local
    blocks = exchange(pad)
    if rank != 0:
        return None
    return np.concatenate([np.asarray(blocks[r])[: hi - lo] for r, (lo, hi) in enumerate(sizes)], 0)
