"""rtk_pivotkv_compact_batched: the eviction scan of a chunk's units as ONE in-place launch (ticket-ordered workgroups,
rotary tables of the distinct new ids in LDS) - against torch.gather, against the two-launch path it replaces
(rtk_pivotkv_evict_batched[_rope] + rtk_pivotkv_place_batched), bit for bit, and through PivotKVCache.

Reference: longvideo_cache.py:278-288 (gathers), :297-306 (re-rotation at the new ids), :308-310 (position cache),
:313-318 (cache rebuild)."""
import ctypes as C
import os
import sys
import types

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "video-retake_amd"))

import synth  # noqa: E402

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def _keep_sets(L, keep, n, g):
    """kept-index patterns: random, all sources inside the destination range, all beyond it, a dense head + sparse
    tail, shifted by one (every row reads its right neighbour's destination)."""
    out = []
    for i in range(n):
        kind = i % 5
        if kind == 0:
            idx = torch.sort(torch.randperm(L, generator=g, device=dev())[:keep]).values
        elif kind == 1:
            idx = torch.arange(keep, device=dev())
        elif kind == 2:
            idx = torch.arange(L - keep, L, device=dev())
        elif kind == 3:
            h = keep // 2
            idx = torch.cat([torch.arange(h, device=dev()),
                             h + torch.sort(torch.randperm(L - h, generator=g, device=dev())[:keep - h]).values])
        else:
            idx = torch.arange(1, keep + 1, device=dev()) if keep < L else torch.arange(keep, device=dev())
        out.append(idx.contiguous())
    return out


def _mrope_ids(keep, g, big=0):
    """new ids of kept rows as the selection leaves them: slowly growing temporal ids, h / w inside a small grid"""
    t = torch.sort(torch.randint(0, 6, (keep,), generator=g, device=dev())).values + 40 + big
    h = torch.randint(0, 14, (keep,), generator=g, device=dev()) + 40 + big
    w = torch.randint(0, 14, (keep,), generator=g, device=dev()) + 40 + big
    return torch.stack([t, h, w]).contiguous()


@pytest.mark.parametrize("dtype,Hkv,D,L,keep", [(torch.bfloat16, 4, 128, 6272, 1568), (torch.bfloat16, 4, 128, 2304, 576),
                                                (torch.bfloat16, 6, 128, 300, 77), (torch.float32, 4, 128, 300, 77),
                                                (torch.float16, 8, 64, 500, 333), (torch.bfloat16, 2, 96, 260, 100),
                                                (torch.bfloat16, 4, 128, 64, 1), (torch.bfloat16, 4, 128, 640, 639)])
@pytest.mark.parametrize("k_mode", [0, 1, 2])
def test_compact_batched_equals_gather_and_two_launch_path(dtype, Hkv, D, L, keep, k_mode):
    """All three K modes, several dtypes / head counts / head dims, kept sets that exercise every ordering hazard; twice
    on the same sync workspace (every launch must leave it zeroed).
      V (and K, mode 2): byte-identical to torch.gather, the rest of the tail untouched;
      K, mode 1: the un-rotated rows verbatim;  K, mode 0: the bytes rtk_pivotkv_evict_batched_rope writes."""
    import retake._native as nv

    n, P, cap = 7, 3, L + 150
    dt = {torch.bfloat16: nv.RTK_BF16, torch.float16: nv.RTK_F16, torch.float32: nv.RTK_F32}[dtype]
    es = 4 if dtype == torch.float32 else 2
    S = synth.YARN_FACTOR4_ATTENTION_SCALING
    inv = torch.from_numpy(synth.inv_freq(D)).to(dev())
    h2 = D // 2
    secs = [h2 // 4, (h2 - h2 // 4) // 2, h2 - h2 // 4 - (h2 - h2 // 4) // 2]
    sec = (C.c_int * 3)(*secs)
    g = torch.Generator(device=dev()).manual_seed(1000 + L + keep + k_mode)
    n_ints = nv.lib.rtk_pivotkv_compact_sync_ints(n, Hkv, keep, D, dt)
    assert n_ints > 0
    sync = torch.zeros(n_ints, dtype=torch.int32, device=dev())
    for rep in range(2):
        units = (nv.CompactUnit * n)()
        eunits = (nv.EvictUnit * n)()
        hold = []
        for i, idx in enumerate(_keep_sets(L, keep, n, g)):
            k = torch.randn((Hkv, L, D), generator=g, device=dev()).to(dtype)       # the chunk's rows in the tail
            v = torch.randn((Hkv, L, D), generator=g, device=dev()).to(dtype)
            ku = torch.randn((Hkv, L, D), generator=g, device=dev()).to(dtype)      # the un-rotated rows (own buffer)
            P0 = 13 * i
            kc = torch.zeros((Hkv, cap, D), dtype=dtype, device=dev())
            vc = torch.zeros_like(kc)
            kc[:, P0:P0 + L] = k
            vc[:, P0:P0 + L] = v
            pos_new = _mrope_ids(keep, g, big=(100000 if i == 5 else 0))
            if i == 6:   # ids that spread too far for a block's table: the per-thread route
                pos_new = torch.randint(0, 120000, (3, keep), generator=g, device=dev())
            pos_dst = torch.zeros((3, keep + 20), dtype=torch.int64, device=dev())
            u = units[i]
            u.k_src, u.k_src_stride_h = (ku.data_ptr(), L * D) if k_mode != 2 else (None, 0)
            u.k_tail, u.k_tail_stride_h = kc.data_ptr() + P0 * D * es, cap * D
            u.v_tail, u.v_tail_stride_h = vc.data_ptr() + P0 * D * es, cap * D
            u.keep_idx = idx.data_ptr()
            u.pos_src, u.pos_src_stride = pos_new.data_ptr(), keep
            u.pos_dst, u.pos_dst_stride = pos_dst.data_ptr() + 8 * 9, keep + 20
            # the path it replaces, for the rotated rows
            kd = torch.zeros((Hkv, keep, D), dtype=dtype, device=dev())
            vd = torch.zeros_like(kd)
            e = eunits[i]
            e.k_src, e.k_src_stride_h, e.v_src, e.v_src_stride_h = ku.data_ptr(), L * D, v.data_ptr(), L * D
            e.keep_idx = idx.data_ptr()
            e.cos_new = e.sin_new = None
            e.k_dst, e.k_dst_stride_h, e.v_dst, e.v_dst_stride_h = kd.data_ptr(), keep * D, vd.data_ptr(), keep * D
            e.pos_src, e.pos_src_stride, e.pos_dst, e.pos_dst_stride = pos_new.data_ptr(), keep, None, 0
            hold.append((k, v, ku, kc, vc, idx, pos_new, pos_dst, P0, kd, vd))
        if k_mode == 0:
            nv.check(nv.lib.rtk_pivotkv_evict_batched_rope(eunits, n, Hkv, D, keep, P, dt, nv.ptr(inv), float(S), sec, 3,
                                                           nv.round_mode(dtype), 0, nv.stream()), "evict_rope")
        nv.check(nv.lib.rtk_pivotkv_compact_batched(units, n, Hkv, D, keep, P, dt, k_mode, nv.ptr(inv), float(S), sec, 3,
                                                    nv.round_mode(dtype), nv.ptr(sync), n_ints, nv.stream()),
                 "compact_batched")
        torch.cuda.synchronize()
        for (k, v, ku, kc, vc, idx, pos_new, pos_dst, P0, kd, vd) in hold:
            assert torch.equal(vc[:, P0:P0 + keep], v[:, idx]), "V rows"
            assert torch.equal(vc[:, P0 + keep:P0 + L], v[:, keep:]) and int(vc[:, :P0].abs().sum()) == 0
            assert int(vc[:, P0 + L:].abs().sum()) == 0
            if k_mode == 2:
                assert torch.equal(kc[:, P0:P0 + keep], k[:, idx])
            elif k_mode == 1:
                assert torch.equal(kc[:, P0:P0 + keep], ku[:, idx])
            else:
                assert torch.equal(kc[:, P0:P0 + keep].view(torch.int16 if es == 2 else torch.int32),
                                   kd.view(torch.int16 if es == 2 else torch.int32)), "rotated K rows"
                assert float(kd.float().abs().sum()) > 0
            assert torch.equal(kc[:, P0 + keep:P0 + L], k[:, keep:])
            assert torch.equal(pos_dst[:, 9:9 + keep], pos_new) and int(pos_dst[:, :9].abs().sum()) == 0
        # the workspace is left as it was found: tickets, counters and flags at zero
        assert int(sync.abs().sum()) == 0


def test_compact_plain_rope_ids():
    """P = 1 (LLaVA-Video: plain 1-D ids, one id per kept row): the table cannot pay, the per-thread route must still
    write rtk_pivotkv_evict_batched_rope's bytes."""
    import retake._native as nv

    Hkv, D, L, keep, n = 4, 128, 800, 200, 8
    g = torch.Generator(device=dev()).manual_seed(5)
    inv = torch.from_numpy(synth.inv_freq(D)).to(dev())
    n_ints = nv.lib.rtk_pivotkv_compact_sync_ints(n, Hkv, keep, D, nv.RTK_BF16)
    sync = torch.zeros(n_ints, dtype=torch.int32, device=dev())
    units = (nv.CompactUnit * n)()
    eunits = (nv.EvictUnit * n)()
    hold = []
    for i, idx in enumerate(_keep_sets(L, keep, n, g)):
        v = torch.randn((Hkv, L, D), generator=g, device=dev()).bfloat16()
        ku = torch.randn((Hkv, L, D), generator=g, device=dev()).bfloat16()
        kc = torch.zeros((Hkv, L, D), dtype=torch.bfloat16, device=dev())
        vc = v.clone()
        pos_new = (torch.arange(keep, device=dev()) + 3000 * i).view(1, keep).contiguous()
        kd = torch.zeros((Hkv, keep, D), dtype=torch.bfloat16, device=dev())
        vd = torch.zeros_like(kd)
        u, e = units[i], eunits[i]
        u.k_src, u.k_src_stride_h, u.k_tail, u.k_tail_stride_h = ku.data_ptr(), L * D, kc.data_ptr(), L * D
        u.v_tail, u.v_tail_stride_h, u.keep_idx = vc.data_ptr(), L * D, idx.data_ptr()
        u.pos_src, u.pos_src_stride, u.pos_dst, u.pos_dst_stride = pos_new.data_ptr(), keep, None, 0
        e.k_src, e.k_src_stride_h, e.v_src, e.v_src_stride_h = ku.data_ptr(), L * D, v.data_ptr(), L * D
        e.keep_idx, e.cos_new, e.sin_new = idx.data_ptr(), None, None
        e.k_dst, e.k_dst_stride_h, e.v_dst, e.v_dst_stride_h = kd.data_ptr(), keep * D, vd.data_ptr(), keep * D
        e.pos_src, e.pos_src_stride, e.pos_dst, e.pos_dst_stride = pos_new.data_ptr(), keep, None, 0
        hold.append((v, ku, kc, vc, idx, pos_new, kd, vd))
    nv.check(nv.lib.rtk_pivotkv_evict_batched_rope(eunits, n, Hkv, D, keep, 1, nv.RTK_BF16, nv.ptr(inv), 1.0, None, 0, 1, 0,
                                                   nv.stream()), "evict_rope")
    nv.check(nv.lib.rtk_pivotkv_compact_batched(units, n, Hkv, D, keep, 1, nv.RTK_BF16, 0, nv.ptr(inv), 1.0, None, 0, 1,
                                                nv.ptr(sync), n_ints, nv.stream()), "compact")
    torch.cuda.synchronize()
    for (v, ku, kc, vc, idx, pos_new, kd, vd) in hold:
        assert torch.equal(kc[:, :keep].view(torch.int16), kd.view(torch.int16)) and torch.equal(vc[:, :keep], v[:, idx])


def test_compact_argument_errors():
    import retake._native as nv

    units = (nv.CompactUnit * 1)()
    x = torch.zeros((4, 8, 128), dtype=torch.bfloat16, device=dev())
    idx = torch.arange(4, device=dev())
    sync = torch.zeros(256, dtype=torch.int32, device=dev())
    u = units[0]
    u.k_src, u.k_src_stride_h, u.k_tail, u.k_tail_stride_h = x.data_ptr(), 8 * 128, x.data_ptr(), 8 * 128
    u.v_tail, u.v_tail_stride_h, u.keep_idx = x.data_ptr(), 8 * 128, idx.data_ptr()
    u.pos_src = u.pos_dst = None
    inv = torch.ones(64, device=dev())
    call = lambda mode, sy, n: nv.lib.rtk_pivotkv_compact_batched(units, 1, 4, 128, 4, 1, nv.RTK_BF16, mode, nv.ptr(inv), 1.0,  # noqa: E731
                                                                  None, 0, 1, sy, n, nv.stream())
    assert call(1, nv.ptr(sync), 256) == nv.RTK_EINVAL and b"own" in nv.lib.rtk_last_error()      # k_src aliases the tail
    y = torch.zeros_like(x)
    u.k_src = y.data_ptr()
    assert call(0, nv.ptr(sync), 256) == nv.RTK_EINVAL and b"pos_src" in nv.lib.rtk_last_error()
    assert call(1, None, 256) == nv.RTK_EINVAL
    assert call(1, nv.ptr(sync), 8) == nv.RTK_EWORKSPACE
    assert call(7, nv.ptr(sync), 256) == nv.RTK_EINVAL
    assert call(1, nv.ptr(sync), 256) == 0
    torch.cuda.synchronize()


@pytest.mark.parametrize("reforge", [True, False])
@pytest.mark.parametrize("L", [640, 2304])
def test_cache_in_place_compaction_equals_staged(L, reforge):
    """PivotKVCache with the in-place compaction (default) leaves the cache the staged two-launch flush leaves, bit for bit:
    keys, values, ids, lengths - over three chunks and three layers, with a key-patch mask."""
    import retake.longvideo_cache as lc

    Hq, Hkv, D, layers, n_chunks = 28, 4, 128, 3, 3
    sec = [16, 24, 24]
    rot = synth.RotaryStub(synth.inv_freq(D), synth.YARN_FACTOR4_ATTENTION_SCALING, device=dev())

    def run(in_place):
        cfg = types.SimpleNamespace(hidden_size=Hq * D, num_hidden_layers=layers, num_attention_heads=Hq,
                                    num_key_value_heads=Hkv,
                                    longvideo_kwargs={"kvcache_compression": True, "kvcache_compression_kwargs": {
                                        "compression_ratio": 0.25, "compression_method": "pivotkv",
                                        "pos_embed_reforge": reforge, "in_place_compaction": in_place}})
        cache = lc.build_kvcache(cfg)
        for c in range(n_chunks):
            pos = torch.from_numpy(synth.mrope_position_ids(10 + 10 * c, L // 64, 8, 8, hw0=2)).to(dev())
            cache.keypatches_mask_chunk = torch.from_numpy(np.random.default_rng(c).uniform(size=L) < 0.3).to(dev())
            cache.kvcache_compression = True
            for l in range(layers):
                q0, k0, v = synth.qkv_chunk(900 + 10 * c + l, Hq, Hkv, L, D)
                q0, k0, v = (torch.from_numpy(x).to(dev()).bfloat16() for x in (q0, k0, v))
                cache.shift_temporal_ids_(pos, l)
                cos, sin = rot(v, pos)
                q, k = lc.apply_multimodal_rotary_pos_emb(q0, k0, cos, sin, sec)
                cache.update(k, v, l, {"sin": sin, "cos": cos, "query_states": q, "position_ids": pos,
                                       "rotary_emb": rot, "mrope_section": sec})
            cache.after_forward()
        assert (cache._batch.compact_sync is not None) == in_place
        return cache

    a, b = run(True), run(False)
    for l in range(layers):
        assert torch.equal(a.key_cache[l].view(torch.int16), b.key_cache[l].view(torch.int16))
        assert torch.equal(a.value_cache[l].view(torch.int16), b.value_cache[l].view(torch.int16))
        if reforge:
            assert torch.equal(a.position_cache[l], b.position_cache[l])
        assert a.get_seq_length(l) == b.get_seq_length(l) == n_chunks * int(0.25 * L)
