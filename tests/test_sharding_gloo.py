"""The N > 1 path on CPU: world_size-2 gloo processes shard a segment list, 'classify' their
shard and gather the per-segment results to rank 0 (birda_amd/sharding.py)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from birda_amd.sharding import gather_rows, pack_topk, shard_range, shard_ranges_weighted, unpack_topk


def test_shard_ranges_partition_the_list():
    for n in (0, 1, 7, 1000, 10000, 10001):
        for w in (1, 2, 3, 8):
            r = [shard_range(n, g, w) for g in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
            assert max(h - l for l, h in r) - min(h - l for l, h in r) <= 1
    assert shard_range(10000, 3, 8) == (3750, 5000)   # C3: 1 250 segments per GPU


def test_weighted_ranges_balance_mixed_rate_lists():
    """BASELINE config 5: 22.05 / 44.1 / 48 kHz segments round-robin, balanced by SOURCE samples."""
    src = {22050: 66150, 44100: 132300, 48000: 144000}
    w = [src[(22050, 44100, 48000)[i % 3]] for i in range(10000)]
    for world in (1, 2, 3, 8):
        b = shard_ranges_weighted(w, world)
        assert b[0] == 0 and b[-1] == len(w) and all(x <= y for x, y in zip(b, b[1:]))
        loads = [sum(w[b[g]:b[g + 1]]) for g in range(world)]
        assert max(loads) - min(loads) <= 2 * max(w), (world, loads)
    # a skewed list: the first quarter holds half of the samples -> with two shards the cut sits near that quarter, not at n / 2
    w = [4] * 100 + [1] * 400
    b = shard_ranges_weighted(w, 2)
    assert 95 <= b[1] <= 105
    # no weights to go by: equal counts; more shards than items: empty shards at the end, still a partition
    assert shard_ranges_weighted([0] * 10, 4) == [shard_range(10, g, 4)[0] for g in range(4)] + [10]
    b = shard_ranges_weighted([5, 5], 8)
    assert b[0] == 0 and b[-1] == 2 and sorted(b) == b


def test_packed_rows_keep_indices_integral():
    """The gather carries int32 rows: class indices beyond 2**24 (not exact in f32) survive, confidences are bit-exact."""
    idx = np.array([[0, 16777217, 2147483647, -1, 5]], np.int32)
    conf = np.array([[0.9, 0.5, 1e-30, 0.0, np.float32(0.1)]], np.float32)
    i2, c2 = unpack_topk(pack_topk(idx, conf))
    assert np.array_equal(i2, idx) and np.array_equal(c2.view(np.int32), conf.view(np.int32))


def _fake_topk(i):
    rng = np.random.default_rng(1000 + i)
    idx = rng.integers(0, 2 ** 31 - 1, 5).astype(np.int32)
    conf = np.sort(rng.random(5).astype(np.float32))[::-1]
    return pack_topk(idx[None], conf[None])[0]


def _worker(rank, world, port, n_total, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(n_total, rank, world)
    local = np.stack([_fake_topk(i) for i in range(lo, hi)]) if hi > lo else np.zeros((0, 10), np.int32)

    def exchange(pad):                      # the collective the launcher brings: gloo here, RCCL on the GPUs (bench.py)
        t = torch.from_numpy(pad)
        bufs = [torch.empty_like(t) for _ in range(world)] if rank == 0 else None
        dist.gather(t, bufs, dst=0)
        return [b.numpy() for b in bufs] if rank == 0 else None
    full = gather_rows(local, n_total, rank, world, exchange)
    # bench.py's own gather (torch tensors, the path the driver's multi-GPU run takes) must agree
    import bench
    full_t = bench.torch_gather_rows(torch.from_numpy(local), n_total, rank, world, True)
    if rank == 0:
        assert np.array_equal(full_t.numpy(), full)
        np.save(out_path, full)
    else:
        assert full is None and full_t is None
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
