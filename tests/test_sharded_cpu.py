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


# ---------------------------------------------------------------------------------------------------
# DPSelect at ratio < 1 on sharded frames: the exchange plan (who sends which kept frame where)
# ---------------------------------------------------------------------------------------------------
def _random_selection(seed, T, N, t, sync):
    g = torch.Generator().manual_seed(seed)
    if sync:
        return torch.sort(torch.randperm(T, generator=g)[:t]).values
    return torch.stack([torch.sort(torch.randperm(T, generator=g)[:t]).values for _ in range(N)], 1)


@pytest.mark.parametrize("sync", [True, False])
@pytest.mark.parametrize("T,N,t,world", [(32, 5, 13, 4), (16, 3, 1, 2), (24, 4, 23, 8), (8, 2, 4, 1), (64, 7, 16, 8)])
def test_frame_exchange_plan_reassembles_global_gather(T, N, t, world, sync):
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "video-retake_amd"))
    from retake.sharded import plan_frame_exchange

    C = 3
    x = torch.arange(T * N * C, dtype=torch.float32).reshape(T, N, C)
    idx = _random_selection(7 * T + t, T, N, t, sync)
    T_own = T // world
    start, cmax, local, place = plan_frame_exchange(idx, T_own, world)
    assert start.shape == (world + 1, 1 if sync else N) and int(start[-1].min()) == t
    assert local.shape == (world, cmax, 1 if sync else N) and int(local.min()) >= 0 and int(local.max()) < T_own
    # what each rank would send (local gather of its own frames), in rank order, then the placement gather
    blocks = []
    for r in range(world):
        own = x[r * T_own:(r + 1) * T_own]
        if sync:
            blocks.append(own[local[r, :, 0]])
        else:
            blocks.append(torch.gather(own, 0, local[r][:, :, None].expand(-1, -1, C)))
    cat = torch.cat(blocks, 0)
    if sync:
        out, ref = cat[place[:, 0]], x[idx]
    else:
        out = torch.gather(cat, 0, place[:, :, None].expand(-1, -1, C))
        ref = torch.gather(x, 0, idx[:, :, None].expand(-1, -1, C))
    assert torch.equal(out, ref)


def test_frame_exchange_plan_handles_a_rank_with_nothing_kept():
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "video-retake_amd"))
    from retake.sharded import plan_frame_exchange

    idx = torch.tensor([0, 1, 2, 12, 13])            # ranks 1 and 2 of 4 (T_own 4) keep nothing
    start, cmax, local, place = plan_frame_exchange(idx, 4, 4)
    assert start[:, 0].tolist() == [0, 3, 3, 3, 5] and cmax == 3
    assert place[:, 0].tolist() == [0, 1, 2, 9, 10]


def _exchange(rank, world):
    """Both ranks hold the same selection; each gathers its own kept frames, all-gathers the padded blocks and
    places them - the torch indexing stands in for rtk_gather_frames, the collective is the real one."""
    from retake import sharded

    T, N, C, t = 12, 3, 2, 7
    x = torch.arange(T * N * C, dtype=torch.float32).reshape(T, N, C)
    idx = _random_selection(5, T, N, t, False)
    T_own = T // world
    _, cmax, local, place = sharded.plan_frame_exchange(idx, T_own, world)
    own = x[rank * T_own:(rank + 1) * T_own]
    block = torch.gather(own, 0, local[rank][:, :, None].expand(-1, -1, C)).contiguous()
    blocks = sharded.all_gather_rows(block)
    out = torch.gather(blocks, 0, place[:, :, None].expand(-1, -1, C))
    ref = torch.gather(x, 0, idx[:, :, None].expand(-1, -1, C))
    return bool(torch.equal(out, ref)), tuple(blocks.shape)


def test_frame_exchange_over_gloo():
    for ok, shape in _spawn("_exchange"):
        assert ok and shape[1:] == (3, 2)


# ---------------------------------------------------------------------------------------------------
# cache assembly: every layer in two collectives == three per-layer all-gathers + cat
# ---------------------------------------------------------------------------------------------------
def _assemble(rank, world):
    from retake import sharded

    layers, Hkv, n, D = 3, 2, 5, 4
    g = torch.Generator().manual_seed(100 + rank)
    keys = [torch.randn((1, Hkv, n, D), generator=g) for _ in range(layers)]
    vals = [torch.randn((1, Hkv, n, D), generator=g) for _ in range(layers)]
    ok = True
    for pshape in ((3, 1, n), (1, n)):
        pos = [torch.randint(0, 1000, pshape, generator=g) for _ in range(layers)]
        ka, va, pa = sharded.all_gather_caches(keys, vals, pos)
        for l in range(layers):
            ok &= bool(torch.equal(ka[l], sharded.all_gather_cat(keys[l], 2)))
            ok &= bool(torch.equal(va[l], sharded.all_gather_cat(vals[l], 2)))
            ok &= bool(torch.equal(pa[l], sharded.all_gather_cat(pos[l], -1)))
            ok &= tuple(ka[l].shape) == (1, Hkv, world * n, D) and tuple(pa[l].shape) == pshape[:-1] + (world * n,)
    return ok


def test_cache_assembly_two_collectives_equal_per_layer_gathers():
    assert all(_spawn("_assemble"))
    assert all(_spawn("_assemble", world=3))


def _chunk_gather(rank, world):
    """Per-chunk asynchronous gathers + finish == one gather of the concatenated blocks at the end."""
    from retake import sharded

    layers, Hkv, D, chunks = 2, 3, 4, 3
    sizes = [5, 5, 2]                                   # rows kept per chunk (same on every rank)
    g = torch.Generator().manual_seed(7 + rank)
    cg = sharded.ChunkGather()
    k_all = [[] for _ in range(layers)]
    v_all = [[] for _ in range(layers)]
    for c in range(chunks):
        ks = [torch.randn((Hkv, sizes[c] + 3, D), generator=g)[:, 3:] for _ in range(layers)]     # strided views
        vs = [torch.randn((Hkv, sizes[c], D), generator=g) for _ in range(layers)]
        cg.start(ks, vs)
        for l in range(layers):
            k_all[l].append(ks[l])
            v_all[l].append(vs[l])
    assert cg.rows() == sum(sizes)
    kv = cg.finish()
    keys = [torch.cat(k_all[l], 1)[None] for l in range(layers)]
    vals = [torch.cat(v_all[l], 1)[None] for l in range(layers)]
    pos = [torch.zeros((1, sum(sizes)), dtype=torch.int64) for _ in range(layers)]
    ka, va, _ = sharded.all_gather_caches(keys, vals, pos)
    ok = tuple(kv.shape) == (2, layers, Hkv, world * sum(sizes), D)
    for l in range(layers):
        ok &= bool(torch.equal(kv[0, l][None], ka[l])) and bool(torch.equal(kv[1, l][None], va[l]))
    # the offsets table every rank computes
    last = torch.tensor([4 + rank, 9 - rank], dtype=torch.int64)
    table = sharded.exchange_temporal_offsets(last, first_start=3, all_ranks=True)
    mine = sharded.exchange_temporal_offsets(last, first_start=3)
    ok &= bool(torch.equal(table[rank], mine)) and tuple(table.shape) == (world, 2) and table[0].tolist() == [3, 3]
    return ok


def test_chunk_gather_equals_gather_at_the_end():
    assert all(_spawn("_chunk_gather"))
    assert all(_spawn("_chunk_gather", world=3))


# ---------------------------------------------------------------------------------------------------
# ragged blocks: a chunk count that is not a multiple of the world size gives the last ranks fewer rows
# ---------------------------------------------------------------------------------------------------
def _ragged(rank, world):
    from retake import sharded

    n = [5, 3, 3][rank]                                                 # rows this rank holds
    counts = sharded.gather_counts(n, "cpu")
    rows = torch.arange(n * 2, dtype=torch.float32).reshape(n, 2) + 100 * rank
    full = sharded.all_gather_rows_ragged(rows)
    layers, Hkv, D = 2, 2, 4
    keys = [torch.full((1, Hkv, n, D), float(10 * rank + l)) + torch.arange(n)[None, None, :, None] for l in range(layers)]
    vals = [k + 0.5 for k in keys]
    pos = [torch.arange(n)[None, None].repeat(3, 1, 1) + 1000 * rank + l for l in range(layers)]
    K, V, P = sharded.all_gather_caches(keys, vals, pos)
    return counts, full.tolist(), [k.shape for k in K], K[1][0, 1, :, 2].tolist(), V[0][0, 0, :, 0].tolist(), P[1][2, 0].tolist()


def test_ragged_blocks_are_padded_and_trimmed():
    want_rows = [[0, 1], [2, 3], [4, 5], [6, 7], [8, 9], [100, 101], [102, 103], [104, 105], [200, 201], [202, 203], [204, 205]]
    for counts, full, shapes, kcol, vcol, pcol in _spawn("_ragged", world=3):
        assert counts == [5, 3, 3] and full == want_rows
        assert all(tuple(s) == (1, 2, 11, 4) for s in shapes)
        assert kcol == [1, 2, 3, 4, 5, 11, 12, 13, 21, 22, 23]                      # layer 1: 10*rank + 1 + row
        assert vcol == [0.5, 1.5, 2.5, 3.5, 4.5, 10.5, 11.5, 12.5, 20.5, 21.5, 22.5]
        assert pcol == [1, 2, 3, 4, 5, 1001, 1002, 1003, 2001, 2002, 2003]


def test_phase_timer_sums_spans_between_marks_per_step():
    """PhaseTimer on the host clock (its form under gloo): a phase is the span up to its mark, the two orders in which
    `finalize` marks rotate / assembly both add up, `step` is start -> last mark, means are per recorded step."""
    import time

    from retake.sharded import PhaseTimer

    pt = PhaseTimer(torch.device("cpu"))
    pt.mark("ignored")                       # nothing recorded before the first begin()
    for order in (("dpselect", "blocks", "offsets", "assembly", "rotate"), ("dpselect", "blocks", "offsets", "rotate", "assembly")):
        pt.begin()
        for name in order:
            time.sleep(0.01 if name != "blocks" else 0.03)
            pt.mark(name)
    pt.note_bytes("assembly_rows_received", 123)
    ms = pt.per_step_ms()
    assert set(ms) == {"dpselect", "blocks", "offsets", "rotate", "assembly", "step"}
    assert 25 <= ms["blocks"] <= 60 and all(8 <= ms[k] <= 30 for k in ("dpselect", "offsets", "rotate", "assembly"))
    assert abs(ms["step"] - sum(ms[k] for k in PhaseTimer.ORDER)) < 1e-6
    assert pt.bytes == {"assembly_rows_received": 123}
    assert PhaseTimer(torch.device("cpu")).per_step_ms()["step"] == 0.0
