"""The N > 1 path on CPU: world_size-2 gloo processes shard a segment list, 'classify' their
shard and gather the per-segment results to rank 0 (birda_amd/sharding.py)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from birda_amd.sharding import gather_results, shard_range


def test_shard_ranges_partition_the_list():
    for n in (0, 1, 7, 1000, 10000, 10001):
        for w in (1, 2, 3, 8):
            r = [shard_range(n, g, w) for g in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
            assert max(h - l for l, h in r) - min(h - l for l, h in r) <= 1
    assert shard_range(10000, 3, 8) == (3750, 5000)   # C3: 1 250 segments per GPU


def _fake_topk(i):
    rng = np.random.default_rng(1000 + i)
    idx = rng.integers(0, 6522, 5).astype(np.float32)
    conf = np.sort(rng.random(5).astype(np.float32))[::-1]
    return np.concatenate([idx, conf])


def _worker(rank, world, port, n_total, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(n_total, rank, world)
    local = torch.from_numpy(np.stack([_fake_topk(i) for i in range(lo, hi)]) if hi > lo else np.zeros((0, 10), np.float32))
    full = gather_results(local, n_total, rank, world)
    if rank == 0:
        np.save(out_path, full.numpy())
    else:
        assert full is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [1001, 64])
def test_two_rank_gather_restores_segment_order(tmp_path, n_total):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "full.npy")
    mp.spawn(_worker, args=(2, port, n_total, out), nprocs=2, join=True)
    full = np.load(out)
    want = np.stack([_fake_topk(i) for i in range(n_total)])
    assert full.shape == want.shape and np.array_equal(full, want)
