"""World-size-2 gloo tests of the multi-GPU sharding's host logic (retake/sharded.py)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, fn_name, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ret[rank] = globals()[fn_name](rank, world)
    finally:
        dist.destroy_process_group()


def _spawn(fn_name, world=2):
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "video-retake_amd"))
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), fn_name, ret), nprocs=world, join=True)
    return [ret[r] for r in range(world)]


def _offsets(rank, world):
    from retake import sharded

    # per-layer last provisional temporal id of each rank's block (3 layers)
    last = torch.tensor([[7, 9, 5], [6, 8, 8]][rank], dtype=torch.int64)
    return sharded.exchange_temporal_offsets(last, first_start=11).tolist()


def test_temporal_offsets_exclusive_prefix():
    r0, r1 = _spawn("_offsets")
    assert r0 == [11, 11, 11]                 # rank 0 starts where the (text) prefix ended
    assert r1 == [11 + 8, 11 + 10, 11 + 6]    # previous block's last id + 1


def _gathers(rank, world):
    from retake import sharded

    rows = torch.arange(6, dtype=torch.float32).reshape(3, 2) + 100 * rank       # distance rows [T_loc, N]
    full = sharded.all_gather_rows(rows)
    k = torch.full((1, 2, 3, 4), float(rank))                                     # kept K [1, Hkv, n_loc, D]
    cat = sharded.all_gather_cat(k, 2)
    pos = torch.arange(3)[None, None].repeat(3, 1, 1) + 10 * rank                 # [3, 1, n_loc]
    pcat = sharded.all_gather_cat(pos, -1)
    return full.tolist(), cat.shape, cat[0, 0, :, 0].tolist(), pcat[0, 0].tolist()


def test_all_gathers_keep_rank_order():
    for full, shape, col, p in _spawn("_gathers"):
        assert full == [[0, 1], [2, 3], [4, 5], [100, 101], [102, 103], [104, 105]]
        assert tuple(shape) == (1, 2, 6, 4) and col == [0, 0, 0, 1, 1, 1]
        assert p == [0, 1, 2, 10, 11, 12]


def test_shard_chunks_balanced_contiguous():
    from retake import sharded

    assert sharded.shard_chunks(64, 8) == [(8 * r, 8 * r + 8) for r in range(8)]
    assert sharded.shard_chunks(10, 4) == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert sharded.shard_chunks(2, 4) == [(0, 1), (1, 2), (2, 2), (2, 2)]


def test_block_offsets_reproduce_sequential_chain():
    """Sequential chain (single GPU): start of chunk c+1 = last compressed id of chunk c + 1.  With blocks
    processed independently from id 0, adding the exclusive prefix of (last+1) must give the same ids."""
    import numpy as np

    rng = np.random.default_rng(0)
    spans = rng.integers(6, 9, size=8)  # compressed temporal span of each chunk
    seq_start = np.concatenate([[0], np.cumsum(spans)[:-1]])
    # two blocks of four chunks, each starting at provisional id 0
    for blocks in ([(0, 4), (4, 8)], [(0, 2), (2, 5), (5, 8)]):
        last = [int(np.sum(spans[a:b]) - 1) for a, b in blocks]
        delta = np.concatenate([[0], np.cumsum(np.array(last) + 1)[:-1]])
        for (a, b), d in zip(blocks, delta):
            prov = np.concatenate([[0], np.cumsum(spans[a:b])[:-1]])
            np.testing.assert_array_equal(prov + d, seq_start[a:b])
