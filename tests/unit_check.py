"""Single-unit launches of the PivotKV scoring / selection entry points, straight through the C ABI.

Shared by tests/test_hip_parity.py and bench.py's untimed self-check: the chunk-batched launches
(`rtk_pivotkv_score_passes_batched`, gridDim.y = layers of the chunk; `rtk_pivotkv_select_batched`) must give,
for every unit, exactly what one-unit launches of `rtk_pivotkv_score` + `rtk_pivotkv_select` give on the same inputs
(reference semantics: longvideo_cache.py:248-277).
"""
from __future__ import annotations

import ctypes as C

import torch


def per_unit_score_select(q, k, pos2d, mask, keep, reforge, inv_freq, attention_scaling, sections, stream_sync=True,
                          score_dt=None):
    """q [1,Hq,L,D], k [1,Hkv,L,D] (rotated, any head/token strides), pos2d [P,L] int64 contiguous, mask [L] bool or
    None.  Returns (score [L] fp32 with the mask override applied, keep_idx [keep] int64, pos_out [P,keep] int64)."""
    import retake._native as nv

    dev = q.device
    _, Hq, L, D = q.shape
    Hkv = k.shape[1]
    P = pos2d.shape[0]
    dt = nv.dtype_code(q) if score_dt is None else score_dt   # e.g. nv.RTK_BF16_FAST: same payload, other score arithmetic
    cos = sin = None
    if reforge:
        cos = torch.empty((L, D), dtype=torch.float32, device=dev)
        sin = torch.empty_like(cos)
        sec = (C.c_int * len(sections))(*sections) if sections else None
        nv.check(nv.lib.rtk_rope_table(nv.ptr(pos2d), L, P, L, nv.ptr(inv_freq), D, float(attention_scaling), sec,
                                       len(sections) if sections else 0, nv.round_mode(q.dtype), nv.ptr(cos),
                                       nv.ptr(sin), nv.stream()), "rtk_rope_table")
    wsb = nv.lib.rtk_pivotkv_score_workspace_bytes(Hq, Hkv, L, D, dt)
    ws = torch.empty(wsb + 256, dtype=torch.uint8, device=dev)
    ws_ptr = C.c_void_p((ws.data_ptr() + 255) & ~255)
    score = torch.empty(L, dtype=torch.float32, device=dev)
    nv.check(nv.lib.rtk_pivotkv_score(nv.ptr(q), q.stride(1), q.stride(2), nv.ptr(k), k.stride(1), k.stride(2), Hq, Hkv,
                                      L, D, dt, nv.ptr(cos), nv.ptr(sin), float(attention_scaling) if reforge else 1.0,
                                      nv.ptr(score), None, ws_ptr, wsb, nv.stream()), "rtk_pivotkv_score")
    keep_idx = torch.empty(keep, dtype=torch.int64, device=dev)
    pos_out = torch.empty((P, keep), dtype=torch.int64, device=dev)
    selb = nv.lib.rtk_pivotkv_select_workspace_bytes(L)
    sel_ws = torch.empty(selb + 256, dtype=torch.uint8, device=dev)
    sel_ptr = C.c_void_p((sel_ws.data_ptr() + 255) & ~255)
    m = mask.to(torch.bool).contiguous() if mask is not None else None
    nv.check(nv.lib.rtk_pivotkv_select(nv.ptr(score), nv.ptr(m), L, keep, nv.ptr(pos2d), P, int(reforge),
                                       nv.ptr(keep_idx), None, nv.ptr(pos_out), keep, sel_ptr, selb, nv.stream()),
             "rtk_pivotkv_select")
    if stream_sync:
        torch.cuda.synchronize()
    return score, keep_idx, pos_out


def check_batch_against_units(cache, layers, inputs, masks, keep, inv_freq, attention_scaling, sections):
    """`cache` has just flushed a chunk whose layer l was updated with inputs[l] = (q, k) and masks[l].  Asserts that the
    batched launches left, for every layer in `layers`, bitwise the score / kept indices / new ids of one-unit launches
    on the ids the layer actually used (the cache's private copy: the attention patch shifts the shared tensor per layer).
    Returns the list of (layer, score, keep_idx) for further checks."""
    b = cache._batch
    assert b is not None and not b.pending, "flush first (after_forward)"
    out = []
    for l in layers:
        q, k = inputs[l]
        pos2d = b.pos_old[l].contiguous()
        s1, i1, p1 = per_unit_score_select(q, k, pos2d, masks[l], keep, b.reforge, inv_freq, attention_scaling, sections,
                                           score_dt=b.score_dt)   # same score arithmetic AND split policy as the batch
        sb, ib = b.score[l], b.keep_idx[l]
        if not torch.equal(sb, s1):
            bad = (sb != s1).nonzero().flatten()
            raise AssertionError(f"layer {l}: batched score differs from the one-unit launch at {bad.numel()} of "
                                 f"{sb.numel()} tokens (first {bad[:5].tolist()}, max |d| {(sb - s1).abs().max().item():.3e})")
        if not torch.equal(ib, i1):
            raise AssertionError(f"layer {l}: batched keep_idx differs from the one-unit selection")
        if b.P and not torch.equal(b.pos_new[:, l], p1):
            raise AssertionError(f"layer {l}: batched new position ids differ from the one-unit selection")
        out.append((l, sb, ib))
    return out
