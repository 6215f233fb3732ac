"""A WHOLE 2048-frame video (BASELINE configs[2] size) against the CPU oracle, chunk by chunk and as a final cache.

64 chunks of L = 6272 tokens, fp32 (the parity dtype), M-RoPE, pos_embed_reforge, key-patch mask from DPSelect, 4x
compression: the chain  continuity shift (qwen2_vl.py:68-73) -> update (longvideo_cache.py:217-310) -> id reforge +
compaction (:283-318)  runs 64 times on ONE layer's cache (100,352 kept rows at the end) next to the oracle on the host.
Bar: kept indices of every chunk bit-exact wherever the oracle's own k-th boundary is not at fp32-noise level (margin
aware, like test_hip_parity.test_pivotkv_full_chunk_vs_oracle_margin_aware), and the FINAL cache - ids exact, V exact,
K within 1e-5 - plus num_evicted_tokens.  Two product routes: the attention prologue (update_pre_rope on the pre-RoPE
projections, what the build's attention patch calls for video chunks) for the 64-chunk video, and the reference's
cache_kwargs protocol (update on rotated tensors after shift_temporal_ids_) on layer 27 of a 28-layer cache for 4 chunks.
"""
import time
import types

import numpy as np
import pytest
import torch

import synth
from oracle import oracle as orc

pytestmark = pytest.mark.gpu

Hq, Hkv, D, L, RATIO = 28, 4, 128, 6272, 0.25
SEC = [16, 24, 24]
A = synth.YARN_FACTOR4_ATTENTION_SCALING
KEEP = int(RATIO * L)
NOISE = 2e-5      # a decision whose oracle margin is below this may come out either way in another fp32 summation order


def dev():
    return torch.device("cuda:0")


def _cfg(layers):
    return types.SimpleNamespace(hidden_size=Hq * D, num_hidden_layers=layers, num_attention_heads=Hq, num_key_value_heads=Hkv,
                                 longvideo_kwargs={"kvcache_compression": True, "kvcache_compression_kwargs": {
                                     "compression_ratio": RATIO, "compression_method": "pivotkv", "pos_embed_reforge": True}})


def _dpselect_mask(n_chunks):
    """Key-patch mask of the whole synthetic video from the product's DPSelect (shipped default: ratio 1.0, patch_sync
    False), and how many of its entries the oracle's DPSelect on the same frames disagrees with."""
    import bench as B
    import retake.visual_compression as vc

    T = n_chunks * 32
    frames = torch.cat([B.chunk_frames(c, dev(), torch.float32) for c in range(n_chunks)])[None]
    _, mask = vc.memory_bank_compress_keyframe(frames, T, 3, sync=False)
    mask = mask.cpu().numpy()
    o_mask = orc.dpselect(frames[0].cpu().numpy()[None], T, 3, False)[1]
    return mask, int((mask != o_mask).sum())


def _chunk_ids(c):
    return synth.mrope_position_ids(16 + 32 * c, 32, 14, 14, hw0=16)


def _compare_chunk(c, score, idx, oc, stats):
    so = oc.last["score"]
    err = np.abs(score - so).max()
    assert err < 5e-6, (c, err)
    srt = np.sort(so)[::-1]
    gap = srt[KEEP - 1] - srt[KEEP]
    diff = np.setxor1d(idx, oc.last["keep_idx"])
    stats["max_score_err"] = max(stats["max_score_err"], float(err))
    stats["min_gap"] = min(stats["min_gap"], float(gap))
    if gap > NOISE:
        assert diff.size == 0, f"chunk {c}: kept sets differ although the oracle's margin is {gap:.2e}"
    elif diff.size:
        assert np.abs(so[diff] - srt[KEEP - 1]).max() < NOISE, f"chunk {c}: a disagreement away from the threshold"
        stats["fragile_chunks_differing"].append(c)
    return diff.size == 0


def test_whole_video_64_chunks_through_the_prologue_against_the_oracle():
    import retake.longvideo_cache as lc

    n_chunks = 64
    t0 = time.time()
    mask, mask_mismatch = _dpselect_mask(n_chunks)
    assert mask_mismatch <= 8, mask_mismatch      # peak decisions at fp32-noise margins only (async rows: 401,408 entries)
    rot = synth.RotaryStub(synth.inv_freq(D), A, device=dev())
    rot_cpu = synth.RotaryStub(synth.inv_freq(D), A)
    cache = lc.build_kvcache(_cfg(1), reserve_tokens=n_chunks * KEEP + L)
    oc = orc.OraclePivotKV(Hq, Hkv, D, RATIO, True)
    stats = {"max_score_err": 0.0, "min_gap": np.inf, "fragile_chunks_differing": []}
    same = []
    for c in range(n_chunks):
        q0, k0, v = synth.qkv_chunk(1000 * 0 + c, Hq, Hkv, L, D)      # SURVEY 8(d) cfg 3: seed = 1000 * layer + chunk
        pos = _chunk_ids(c)
        m = mask[c * L:(c + 1) * L]
        # --- oracle: what the reference's attention patch + cache do (shift, rotate, update)
        prev = oc.get_prev_temporal_idx(0)
        pos_sh = pos.copy()
        pos_sh[0, 0] += prev + 1 - pos_sh[0, 0, 0]
        pt = torch.from_numpy(pos_sh)
        q = synth.rope_forward(torch.from_numpy(q0), pt, rot_cpu, SEC).numpy()
        k = synth.rope_forward(torch.from_numpy(k0), pt, rot_cpu, SEC).numpy()
        oc.keypatches_mask_chunk = m
        oc.update(k, v, 0, q=q, position_ids=pos_sh, rotary=rot_cpu, mrope_section=SEC)
        # --- product: the prologue gets the pre-RoPE projections and the UNSHIFTED ids
        cache.keypatches_mask_chunk = torch.from_numpy(m).to(dev())
        cache.kvcache_compression = True
        cache.before_forward()
        ids = torch.from_numpy(pos).to(dev())
        qd, kd, vd = (torch.from_numpy(a).to(dev()).transpose(1, 2).contiguous().transpose(1, 2) for a in (q0, k0, v))
        out = cache.update_pre_rope(qd, kd, vd, 0, ids, rot, SEC)
        assert out is not None and out[1].shape[2] == c * KEEP + L        # [compressed prefix | whole chunk] for the attention
        score = cache.last_scores.cpu().numpy()
        idx = cache.last_keep_indices.cpu().numpy()
        cache.after_forward()
        assert torch.equal(ids.cpu(), pt), f"chunk {c}: the caller's ids after the chunk are not the shifted ones"
        same.append(_compare_chunk(c, score, idx, oc, stats))
        if not same[-1]:   # ids of later chunks follow the last kept id: must still agree for the run to stay comparable
            assert int(cache.get_prev_temporal_idx(0)) == oc.get_prev_temporal_idx(0)
    # --- the final cache: 100,352 rows
    n = n_chunks * KEEP
    K, V, P = cache.key_cache[0].cpu().numpy(), cache.value_cache[0].cpu().numpy(), cache.position_cache[0].cpu().numpy()
    assert K.shape == (1, Hkv, n, D) and oc.key_cache[0].shape == K.shape
    rows = np.repeat(np.array(same), KEEP)
    assert rows.mean() > 0.9
    np.testing.assert_array_equal(P[..., rows], oc.position_cache[0][..., rows])
    np.testing.assert_array_equal(V[:, :, rows], oc.value_cache[0][:, :, rows])
    kerr = np.abs(K[:, :, rows] - oc.key_cache[0][:, :, rows]).max()
    assert kerr <= 1e-5, kerr
    assert cache.num_evicted_tokens == oc.num_evicted_tokens == [n_chunks * (L - KEEP)]
    assert cache.get_seq_length(0) == n
    print(f"\n[whole video, prologue route] {n_chunks} chunks x L {L}: {sum(same)} chunks with bit-exact kept indices, "
          f"{len(stats['fragile_chunks_differing'])} differing inside fp32 noise of a fragile boundary {stats['fragile_chunks_differing']}; "
          f"max |score - oracle| {stats['max_score_err']:.2e}, smallest oracle margin {stats['min_gap']:.2e}; final cache {n} rows: "
          f"ids exact, V exact, max |K - oracle| {kerr:.2e}; DPSelect mask entries differing from the oracle's {mask_mismatch}; "
          f"{time.time() - t0:.0f} s")


def test_layer_27_of_a_28_layer_cache_against_the_oracle():
    """4 chunks through all 28 layers on the reference-protocol route (update on rotated tensors after the on-device
    continuity shift; the ids tensor is shared by the layers of a chunk and shifted in place): the chunk-batched flush of 28
    units, compared on layers 0 and 27 (SURVEY 8(d) cfg 3 seeds 1000 * layer + chunk)."""
    import retake.longvideo_cache as lc

    n_chunks, layers, watch = 4, 28, (0, 27)
    mask, _ = _dpselect_mask(n_chunks)
    rot = synth.RotaryStub(synth.inv_freq(D), A, device=dev())
    rot_cpu = synth.RotaryStub(synth.inv_freq(D), A)
    cache = lc.build_kvcache(_cfg(layers), reserve_tokens=n_chunks * KEEP + L)
    ocs = {l: orc.OraclePivotKV(Hq, Hkv, D, RATIO, True) for l in watch}
    stats = {"max_score_err": 0.0, "min_gap": np.inf, "fragile_chunks_differing": []}
    same = {l: [] for l in watch}
    g = torch.Generator(device=dev()).manual_seed(77)
    for c in range(n_chunks):
        m = mask[c * L:(c + 1) * L]
        cache.keypatches_mask_chunk = torch.from_numpy(m).to(dev())
        cache.kvcache_compression = True
        ids = torch.from_numpy(_chunk_ids(c)).to(dev())
        for l in range(layers):
            cache.shift_temporal_ids_(ids, l)
            if l in watch:
                q0, k0, v = synth.qkv_chunk(1000 * l + c, Hq, Hkv, L, D)
                oc = ocs[l]
                prev = oc.get_prev_temporal_idx(0)
                pos_sh = _chunk_ids(c)
                pos_sh[0, 0] += prev + 1 - pos_sh[0, 0, 0]
                assert np.array_equal(ids.cpu().numpy(), pos_sh)
                pt = torch.from_numpy(pos_sh)
                q = synth.rope_forward(torch.from_numpy(q0), pt, rot_cpu, SEC)
                k = synth.rope_forward(torch.from_numpy(k0), pt, rot_cpu, SEC)
                oc.keypatches_mask_chunk = m
                oc.update(k.numpy(), v, 0, q=q.numpy(), position_ids=pos_sh, rotary=rot_cpu, mrope_section=SEC)
                qd, kd, vd = q.to(dev()), k.to(dev()), torch.from_numpy(v).to(dev())
            else:   # the other layers: device-made inputs, only there to fill the batch
                qd, kd, vd = (1.7 * torch.randn((1, h, L, D), generator=g, device=dev()) for h in (Hq, Hkv, Hkv))
            cache.update(kd, vd, l, {"query_states": qd, "position_ids": ids, "rotary_emb": rot, "mrope_section": SEC})
        assert len(cache._batch.pending) == layers
        cache.after_forward()
        b = cache._batch
        for l in watch:
            same[l].append(_compare_chunk(c, b.score[l].cpu().numpy(), b.keep_idx[l].cpu().numpy(), ocs[l], stats))
    n = n_chunks * KEEP
    for l in watch:
        rows = np.repeat(np.array(same[l]), KEEP)
        assert rows.mean() >= 0.5
        oc = ocs[l]
        np.testing.assert_array_equal(cache.position_cache[l].cpu().numpy()[..., rows], oc.position_cache[0][..., rows])
        np.testing.assert_array_equal(cache.value_cache[l].cpu().numpy()[:, :, rows], oc.value_cache[0][:, :, rows])
        kerr = np.abs(cache.key_cache[l].cpu().numpy()[:, :, rows] - oc.key_cache[0][:, :, rows]).max()
        assert kerr <= 1e-5, (l, kerr)
        assert cache.num_evicted_tokens[l] == oc.num_evicted_tokens[0] == n_chunks * (L - KEEP)
        print(f"\n[28-layer cache, update route] layer {l}: {sum(same[l])} of {n_chunks} chunks bit-exact, final cache {n} rows: ids exact, "
              f"V exact, max |K - oracle| {kerr:.2e}")
