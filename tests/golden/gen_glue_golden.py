#!/usr/bin/env python3
"""Golden vectors for the glue around the hot path, produced by driving the imported
REFERENCE (/root/reference, read-only, build container only) with the stub modules of tests/glue_stubs.py:

  glue_qwen2vl_forward_*.npz   retake_Qwen2VLForConditionalGeneration_forward (qwen2_vl.py:522-764): what the language
                               model is handed per text segment / video chunk, cache flags and key-patch mask slices
  glue_llava_forward_*.npz     retake_LlavaOnevisionForConditionalGeneration_forward (llava_onevision.py:306-583) incl.
                               the t*side^2 -> t*pooled+1 key-patch mask truncation (:486) and the front trim (:261)
  glue_llava.npz               compress_video_tokens / forge_input_chunks / segment_input_ids / get_chunk_size of LLaVA
  glue_attention_qwen2vl.npz   retake_Qwen2VLAttention_forward (qwen2_vl.py:42-122) over text -> 2 video chunks -> text ->
  glue_attention_llava.npz     decode with the reference PivotKVCache: outputs, shifted ids, final cache
                               (retake_Qwen2Attention_forward, llava_onevision.py:59-141, likewise)

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_glue_golden.py
"""
from __future__ import annotations

import os
import sys
import types

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import gen_golden as G  # noqa: E402
import glue_stubs as gs  # noqa: E402

torch.set_num_threads(8)


def dpselect_margin(bank, tgt, sync):
    """Smallest decision margin (fp64) of DPSelect on bank [1,T,N,C]: the fixture must not sit on a fragile decision."""
    d32, d64 = G.dis_matrices(bank)
    rows = d64.mean(1, keepdims=True).T if sync else d64.T
    pm = G.peak_margins(rows)
    pk = (rows > np.concatenate([np.full((rows.shape[0], 1), -np.inf), rows[:, :-1]], 1)) & \
         (rows >= np.concatenate([rows[:, 1:], np.full((rows.shape[0], 1), -np.inf)], 1))
    return min(pm, G.topk_margin(rows + 2.0 * pk, tgt))


# --------------------------------------------------------------------------------------
# model forwards
# --------------------------------------------------------------------------------------
def gen_qwen_forward(qv, outdir):
    cases = {
        "base": dict(cfg=dict(ratio=0.5), inp=dict(grid_t=24)),
        "fcs_sync": dict(cfg=dict(ratio=1.0, sync=True, frame_chunk_size=5, chunk_frames=6), inp=dict(grid_t=16, seed=79)),
        "dynamic": dict(cfg=dict(ratio=0.5, dynamic=40), inp=dict(grid_t=24, seed=80)),
        # the prompt fits max_input_length: the dynamic ratio becomes the integer 1 (qwen2_vl.py:553-554)
        "dynamic_fits": dict(cfg=dict(ratio=0.5, dynamic=100000), inp=dict(grid_t=24, seed=83)),
    }
    for name, c in cases.items():
        cfg = gs.qwen_config(**c["cfg"])
        me = gs.make_qwen_model(qv, cfg)
        kw = gs.qwen_inputs(**c["inp"])
        T = int(kw["video_grid_thw"][0, 0])
        emb = gs.StubVisual(24, 32)(kw["pixel_values_videos"], grid_thw=kw["video_grid_thw"]).reshape(1, T, -1, 32)
        ratio = c["cfg"]["ratio"]
        m = dpselect_margin(emb, max(1, round(ratio * T)), c["cfg"].get("sync", False))
        assert m > G.FRAGILE, (name, m)
        out = qv.retake_Qwen2VLForConditionalGeneration_forward(me, return_dict=False, **{k: v.clone() for k, v in kw.items()})
        rec = gs.calls_to_record(me.model.calls, "call")
        rec["logits"] = out[0].numpy()
        rec["visual_calls"] = np.array(me.visual.calls, dtype=np.int64)
        rec["kv_ratio_after"] = float(cfg.longvideo_kwargs["kvcache_compression_kwargs"]["compression_ratio"])
        rec["cache_ratio"] = float(out[1].compression_ratio)
        rec["cache_class"] = type(out[1]).__name__
        rec["margin"] = m
        np.savez_compressed(os.path.join(outdir, f"glue_qwen2vl_forward_{name}.npz"), **rec)
        print(f"glue_qwen2vl_forward_{name}: {len(me.model.calls)} model calls, visual calls {me.visual.calls}, "
              f"kv ratio {rec['kv_ratio_after']:.4f}, margin {m:.2e}")


def gen_qwen_image(qv, outdir):
    """Image tokens beside the video (qwen2_vl.py:593-596, :631-645): features scattered into the text segment; the
    ValueError when features and tokens disagree."""
    cfg = gs.qwen_config(ratio=0.5)
    me = gs.make_qwen_model(qv, cfg)
    kw = gs.qwen_inputs_with_image(grid_t=16, seed=87)
    out = qv.retake_Qwen2VLForConditionalGeneration_forward(me, return_dict=False, **{k: v.clone() for k, v in kw.items()})
    rec = gs.calls_to_record(me.model.calls, "call")
    rec["logits"] = out[0].numpy()
    rec["visual_calls"] = np.array(me.visual.calls, dtype=np.int64)
    me = gs.make_qwen_model(qv, gs.qwen_config(ratio=0.5))
    try:
        qv.retake_Qwen2VLForConditionalGeneration_forward(me, return_dict=False, **gs.qwen_inputs_with_image(grid_t=16, seed=87, rows_per_token=3))
        rec["mismatch_exc"], rec["mismatch_msg"] = "none", ""
    except Exception as e:  # noqa: BLE001
        rec["mismatch_exc"], rec["mismatch_msg"] = type(e).__name__, str(e)
    np.savez_compressed(os.path.join(outdir, "glue_qwen2vl_forward_image.npz"), **rec)
    print(f"glue_qwen2vl_forward_image: {len(me.model.calls)} calls after the failed forward, visual calls "
          f"{rec['visual_calls'].tolist()}, mismatch -> {rec['mismatch_exc']}: {rec['mismatch_msg']}")


def gen_qwen_prompt_guided(qv, outdir):
    """Prompt-guided forge_input_chunks of Qwen2-VL (qwen2_vl.py:500-517): on an instance of the HF class (the branch
    checks isinstance) and the NotImplementedError for any other `self`."""
    cfg = gs.qwen_config(ratio=0.5, prompt_guided=True)
    seg, cp, pos, am, ie = gs.prompt_guided_case()
    out = qv.retake_Qwen2VLForConditionalGeneration_forge_input_chunks(gs.bare_hf_qwen2vl(cfg), 3, 7, seg, cp, pos, am, None, ie)
    rec = {"cp": out[0].numpy(), "pos": out[1].numpy(), "am": out[2].numpy(), "ie": out[3].numpy(),
           "prompt_length": int(out[4])}
    try:
        qv.retake_Qwen2VLForConditionalGeneration_forge_input_chunks(types.SimpleNamespace(config=cfg), 3, 7, seg, cp, pos, am, None, ie)
        rec["other_class_exc"] = "none"
    except Exception as e:  # noqa: BLE001
        rec["other_class_exc"] = type(e).__name__
    cfg1 = gs.qwen_config(ratio=0.5, kv_ratio=1, prompt_guided=True)   # ratio 1: the branch is off (:502)
    out = qv.retake_Qwen2VLForConditionalGeneration_forge_input_chunks(types.SimpleNamespace(config=cfg1), 3, 7, seg, cp, pos, am, None, ie)
    rec["ratio1_prompt_length_is_none"] = out[4] is None
    rec["ratio1_cp"] = out[0].numpy()
    np.savez_compressed(os.path.join(outdir, "glue_qwen2vl_prompt_guided.npz"), **rec)
    print(f"glue_qwen2vl_prompt_guided: prompt_length {rec['prompt_length']}, other class -> {rec['other_class_exc']}, "
          f"ratio 1 -> prompt_length None: {rec['ratio1_prompt_length_is_none']}")


def gen_qwen_generate(qv, outdir):
    """Prefill without position ids + two decode forwards (qwen2_vl.py:543-590, :721-733): glue_stubs.qwen_generate_steps."""
    cfg = gs.qwen_config(ratio=0.5)
    me, outs = gs.qwen_generate_steps(qv, cfg)
    rec = gs.calls_to_record(me.model.calls, "call")
    rec["prefill_logits"] = outs[0][0].numpy()
    for i, o in enumerate(outs[1:-1]):
        rec[f"decode{i}_logits"] = o[0].numpy()
    last = outs[-1]
    rec["last_logits"] = last.logits.numpy()
    rec["last_rope_deltas"] = last.rope_deltas.numpy()
    rec["last_class"] = type(last).__name__
    rec["last_cache_is_prefill_cache"] = last.past_key_values is outs[0][1]
    rec["rope_deltas_attr"] = me.rope_deltas.numpy()
    np.savez_compressed(os.path.join(outdir, "glue_qwen2vl_generate.npz"), **rec)
    print(f"glue_qwen2vl_generate: {len(me.model.calls)} model calls, decode ids "
          f"{[c['position_ids'].reshape(3, -1)[:, 0].tolist() for c in me.model.calls[-2:]]}, deltas {rec['rope_deltas_attr'].tolist()}")


def gen_llava(lo, outdir):
    cases = {
        "base": dict(cfg=dict(ratio=0.5), inp=dict(T=12)),
        "fcs_sync": dict(cfg=dict(ratio=1.0, sync=True, frame_chunk_size=5, chunk_frames=3), inp=dict(T=10, seed=81)),
        "dynamic_odd": dict(cfg=dict(ratio=0.5, dynamic=30, side=5), inp=dict(T=8, side=5, seed=82)),
        "dynamic_fits": dict(cfg=dict(ratio=0.5, dynamic=100000), inp=dict(T=12, seed=84)),
    }
    for name, c in cases.items():
        cfg = gs.llava_config(**c["cfg"])
        kw, bank = gs.llava_inputs(**c["inp"])
        me = gs.make_llava_model(lo, cfg, bank)
        T = bank.shape[0]
        m = dpselect_margin(bank[None], max(1, round(c["cfg"]["ratio"] * T)), c["cfg"].get("sync", False))
        assert m > G.FRAGILE, (name, m)
        out = lo.retake_LlavaOnevisionForConditionalGeneration_forward(me, return_dict=False,
                                                                      **{k: v.clone() for k, v in kw.items()})
        rec = gs.calls_to_record(me.language_model.calls, "call")
        rec["logits"] = out[0].numpy()
        rec["tower_calls"] = np.array(me.vision_tower.calls, dtype=np.int64)
        rec["kv_ratio_after"] = float(cfg.longvideo_kwargs["kvcache_compression_kwargs"]["compression_ratio"])
        rec["cache_ratio"] = float(out[1].compression_ratio)
        rec["margin"] = m
        np.savez_compressed(os.path.join(outdir, f"glue_llava_forward_{name}.npz"), **rec)
        n_mask = sum(int(cc["mask"].sum()) for cc in me.language_model.calls if cc["mask"] is not None)
        print(f"glue_llava_forward_{name}: {len(me.language_model.calls)} LM calls, tower calls {me.vision_tower.calls}, "
              f"mask bits seen {n_mask}, margin {m:.2e}")

    # function level (llava_onevision.py:144-303)
    cfg = gs.llava_config(ratio=0.5)
    kw, bank = gs.llava_inputs(T=12)
    me = gs.make_llava_model(lo, cfg, bank)
    rec = {}
    seg = me.segment_input_ids(kw["input_ids"])
    rec["seg_s"], rec["seg_e"] = np.array([int(s) for s, _, _ in seg]), np.array([int(e) for _, e, _ in seg])
    rec["seg_t"] = np.array([t for _, _, t in seg])
    rec["chunk_size"] = me.get_chunk_size(cfg, kw["pixel_values_videos"])
    out = me.compress_video_tokens(input_ids=kw["input_ids"].clone(), attention_mask=kw["attention_mask"].clone(),
                                   selected_video_feature=bank.clone(), position_ids=kw["position_ids"].clone(),
                                   cache_position=kw["cache_position"].clone(), labels=None)
    for n, o in zip(["ids", "am", "feat", "pos", "cp"], out[:5]):
        rec["cvt_" + n] = o.numpy()
    rec["cvt_tgt"] = int(out[5])
    rec["cvt_mask"] = out[6].numpy()
    S = kw["input_ids"].shape[1]
    ie = torch.arange(S * 2, dtype=torch.float32).reshape(1, S, 2)
    o = me.forge_input_chunks(5, 17, seg, kw["position_ids"], kw["cache_position"], kw["attention_mask"], None, ie)
    for n, t in zip(["pos", "cp", "am", "ie"], o[:4]):
        rec["fic_" + n] = t.numpy()
    rec["fic_prompt_length_is_none"] = o[4] is None
    cfg.longvideo_kwargs["kvcache_compression_kwargs"]["prompt_guided_compression"] = True
    o = me.forge_input_chunks(5, 17, seg, kw["position_ids"] + 100, kw["cache_position"], kw["attention_mask"], None, ie)
    for n, t in zip(["pos", "cp", "am", "ie"], o[:4]):
        rec["ficp_" + n] = t.numpy()
    rec["ficp_prompt_length"] = int(o[4])
    np.savez_compressed(os.path.join(outdir, "glue_llava.npz"), **rec)
    print("glue_llava: segments", seg, "chunk", rec["chunk_size"], "ids", S, "->", out[0].shape[1], "tgt", out[5])


# --------------------------------------------------------------------------------------
# attention prologue + cache over a realistic call sequence
# --------------------------------------------------------------------------------------
def attention_scenario(lc, forward, llava, seed, ratio=0.5, fa2=False, attn_impl=None):
    """text(5) -> video chunk (32) -> video chunk (32) -> text(3) -> decode(1), two layers sharing one reference
    PivotKVCache (reforge on, ratio 0.5; or 1 - what the dynamic ratio gives a prompt that fits, qwen2_vl.py:553-554).
    Returns the record or None when a top-k decision is fragile."""
    hidden, heads, kvh, D = 64, 4, 2, 16
    S = 1.1386
    layers = [gs.StubAttention(l, hidden, heads, kvh, None if llava else (2, 3, 3), S, seed=seed) for l in range(2)]
    if attn_impl:
        for a in layers:
            a.config._attn_implementation = attn_impl
    cfg = G.make_config(heads, kvh, D, 2, ratio, True, llava=llava)
    cache = lc.PivotKVCache(cfg)
    g = torch.Generator().manual_seed(seed)
    rec = {"llava": llava, "seed": seed, "attention_scaling": S, "ratio": float(ratio), "fa2": fa2}
    for l, a in enumerate(layers):
        for i, w in enumerate(a.weights()):
            rec[f"w{l}_{i}"] = w
    # uncompressed numbering of the prompt: 5 text, 2 x 32 video tokens (2 grids of 4x4 each), 3 text, 1 decode
    steps = [("text", 5), ("video", 32), ("video", 32), ("text", 3), ("decode", 1)]
    captured = []
    orig_topk = torch.Tensor.topk

    def spy(self, *a, **k):
        captured.append(self.detach().clone())
        return orig_topk(self, *a, **k)

    total = 0
    t_next = 0
    for si, (kind, n) in enumerate(steps):
        x = torch.randn(1, n, hidden, generator=g)
        if kind == "video":
            if llava:
                pos = (torch.arange(n) + total)[None]
            else:
                pos = torch.from_numpy(gs.synth.mrope_position_ids(t_next, 2, 4, 4, hw0=t_next))
            t_next += 2 if not llava else 0
            cache.kvcache_compression = True
            cache.keypatches_mask_chunk = torch.rand(n, generator=g) < 0.3
        else:
            base = total if llava else t_next
            pos = (torch.arange(n) + base)[None] if llava else (torch.arange(n) + base)[None, None].repeat(3, 1, 1)
            t_next += n if not llava else 0
            cache.kvcache_compression = False
            cache.keypatches_mask_chunk = None
        total += n
        mask4 = gs.causal_mask(n, total)
        cp = torch.arange(total - n, total)
        rec[f"s{si}_x"], rec[f"s{si}_pos_in"], rec[f"s{si}_mask4"] = x.numpy(), pos.numpy().copy(), mask4.numpy()
        rec[f"s{si}_kpmask"] = cache.keypatches_mask_chunk.numpy() if cache.keypatches_mask_chunk is not None else np.zeros(0, bool)
        rec[f"s{si}_kind"] = kind
        pos_shared = pos.clone()       # HF hands the same ids tensor to every layer of the forward
        for l, att in enumerate(layers):
            torch.Tensor.topk = spy
            try:
                with torch.no_grad():
                    if llava:
                        o = forward(att, x, None, mask4, cache, cp, position_ids=pos_shared)
                    else:   # the FA2 path gets no mask from HF when nothing is padded (bsz 1)
                        o = forward(att, x, None if fa2 else mask4, pos_shared, cache, False, True, cp)
            finally:
                torch.Tensor.topk = orig_topk
            rec[f"s{si}_l{l}_out"] = o[0].numpy()
            rec[f"s{si}_l{l}_pos_after"] = pos_shared.numpy().copy()
            if kind == "video":
                sc = captured.pop().numpy().astype(np.float64)
                srt = -np.sort(-sc)
                keep = max(1, int(ratio * n))
                if keep < n and srt[keep - 1] - srt[keep] < 1e-4:
                    return None
        if hasattr(cache, "after_forward"):
            cache.after_forward()
    for l in range(2):
        rec[f"cache_k{l}"] = cache.key_cache[l].numpy()
        rec[f"cache_v{l}"] = cache.value_cache[l].numpy()
        rec[f"cache_pos{l}"] = cache.position_cache[l].numpy()
    rec["num_evicted"] = np.array(cache.num_evicted_tokens, dtype=np.int64)
    rec["n_steps"] = len(steps)
    return rec


def gen_llava_interface(lc, lo, outdir):
    """The non-eager dispatch of the LLaVA attention patch (llava_onevision.py:118-139): an attention function registered
    in transformers' ALL_ATTENTION_FUNCTIONS under a test-only name, the same scenario as glue_attention_llava."""
    from transformers.modeling_utils import ALL_ATTENTION_FUNCTIONS

    ALL_ATTENTION_FUNCTIONS["retake_test_stub"] = gs.attention_interface_stub
    for attempt in range(3000):
        del gs.INTERFACE_CALLS[:]
        rec = attention_scenario(lc, lo.retake_Qwen2Attention_forward, True, 7000 + attempt, attn_impl="retake_test_stub")
        if rec is not None:
            break
    else:
        raise RuntimeError("llava interface")
    calls = list(gs.INTERFACE_CALLS)
    rec["interface_dropout"] = np.array([c[0] for c in calls])
    rec["interface_scaling"] = np.array([c[1] for c in calls])
    rec["interface_window"] = np.array([c[2] for c in calls])
    rec["interface_kwargs"] = np.array([",".join(c[3]) for c in calls])
    np.savez_compressed(os.path.join(outdir, "glue_attention_llava_interface.npz"), **rec)
    print(f"glue_attention_llava_interface: seed {rec['seed']}, {len(calls)} interface calls, kwargs {calls[0][3]}")


def gen_llava_generate(lo, outdir):
    """Chunked prefill + two decode forwards of the LLaVA glue (llava_onevision.py:330-353, :548-583):
    glue_stubs.llava_generate_steps."""
    cfg = gs.llava_config(ratio=0.5)
    me, outs = gs.llava_generate_steps(lo, cfg)
    rec = gs.calls_to_record(me.language_model.calls, "call")
    rec["prefill_logits"] = outs[0][0].numpy()
    for i, o in enumerate(outs[1:-1]):
        rec[f"decode{i}_logits"] = o[0].numpy()
    last = outs[-1]
    rec["last_logits"] = last.logits.numpy()
    rec["last_class"] = type(last).__name__
    rec["last_cache_is_prefill_cache"] = last.past_key_values is outs[0][1]
    rec["last_video_hidden_states_is_none"] = last.video_hidden_states is None
    rec["last_image_hidden_states_is_none"] = last.image_hidden_states is None
    np.savez_compressed(os.path.join(outdir, "glue_llava_generate.npz"), **rec)
    print(f"glue_llava_generate: {len(me.language_model.calls)} LM calls, class {rec['last_class']}")


def gen_fa2_sliding_window(lc, qv, outdir):
    """The sliding-window branch of the FA2 patch (qwen2_vl.py:268-294; dead in the shipped configs, Qwen2-VL has
    use_sliding_window = false): what reaches `_flash_attention_forward` - the trimmed padding mask, the window - over two
    text steps on a cache that already holds tokens, and the ValueError of a past shorter than the window."""
    calls = []

    def recording_stub(q, k, v, attention_mask, query_length, **kw):
        calls.append((None if attention_mask is None else attention_mask.clone(), kw.get("sliding_window")))
        kw.pop("sliding_window", None)
        return gs.flash_attention_forward_stub(q, k, v, None, query_length, **kw)

    qv._flash_attention_forward = recording_stub
    rec = {}
    for case, (n0, n1, window) in {"trim": (6, 5, 7), "short_past": (3, 6, 7)}.items():
        att = gs.StubAttention(0, 64, 4, 2, (2, 3, 3), 1.0, seed=77)
        att.config.use_sliding_window, att.config.sliding_window, att.config.max_window_layers = True, window, 0
        cache = lc.PivotKVCache(G.make_config(4, 2, 16, 1, 0.5, True))
        cache.kvcache_compression = False
        g = torch.Generator().manual_seed(91)
        rec[f"{case}_w"] = np.array([0])  # placeholder so that the key order is stable
        for i, w in enumerate(att.weights()):
            rec[f"{case}_w{i}"] = w
        total = 0
        del calls[:]
        for si, n in enumerate((n0, n1)):
            x = torch.randn(1, n, 64, generator=g)
            pos = (torch.arange(n) + total)[None, None].repeat(3, 1, 1)
            total += n
            am = torch.ones(1, total, dtype=torch.int64)
            am[0, 0] = 0
            rec[f"{case}_s{si}_x"] = x.numpy()
            try:
                with torch.no_grad():
                    o = qv.retake_Qwen2VLFlashAttention2_forward(att, x, am, pos, cache, False, True, torch.arange(total - n, total))
                rec[f"{case}_s{si}_out"] = o[0].numpy()
                rec[f"{case}_s{si}_exc"] = "none"
                m, sw = calls[-1]
                rec[f"{case}_s{si}_mask_to_fa"] = m.numpy()
                rec[f"{case}_s{si}_window_to_fa"] = -1 if sw is None else int(sw)
            except Exception as e:  # noqa: BLE001
                rec[f"{case}_s{si}_exc"] = type(e).__name__
        rec[f"{case}_shape"] = np.array([n0, n1, window])
    np.savez_compressed(os.path.join(outdir, "glue_attention_qwen2vl_fa2_sliding.npz"), **rec)
    print("glue_attention_qwen2vl_fa2_sliding:", {k: (str(v) if k.endswith("exc") else getattr(v, "shape", v))
                                                  for k, v in rec.items() if "mask_to_fa" in k or k.endswith("exc") or "window_to" in k})


def gen_attention(lc, qv, lo, outdir):
    for name, fwd, llava in (("qwen2vl", qv.retake_Qwen2VLAttention_forward, False),
                             ("llava", lo.retake_Qwen2Attention_forward, True)):
        for attempt in range(50):
            rec = attention_scenario(lc, fwd, llava, 300 + attempt)
            if rec is not None:
                break
            print(f"  [attention_{name}] seed {300 + attempt}: fragile -> reseed")
        else:
            raise RuntimeError(name)
        np.savez_compressed(os.path.join(outdir, f"glue_attention_{name}.npz"), **rec)
        print(f"glue_attention_{name}: seed {rec['seed']}, cache lengths {[rec[f'cache_k{l}'].shape[2] for l in range(2)]}, "
              f"evicted {rec['num_evicted'].tolist()}")
        rec = attention_scenario(lc, fwd, llava, 400, ratio=1)
        np.savez_compressed(os.path.join(outdir, f"glue_attention_{name}_ratio1.npz"), **rec)
        print(f"glue_attention_{name}_ratio1: cache lengths {[rec[f'cache_k{l}'].shape[2] for l in range(2)]}, "
              f"evicted {rec['num_evicted'].tolist()}")
    # the FlashAttention-2 patch (qwen2_vl.py:224-363) - the attention every shipped config selects - around a plain-torch
    # stand-in for transformers' `_flash_attention_forward` (the flash-attn package is not in this image)
    qv._flash_attention_forward = gs.flash_attention_forward_stub
    for attempt in range(3000):   # four tie-free top-k boundaries in a row (the 1.0 ties of the key patches) are rare
        rec = attention_scenario(lc, qv.retake_Qwen2VLFlashAttention2_forward, False, 500 + attempt, fa2=True)
        if rec is not None:
            break
    else:
        raise RuntimeError("fa2")
    np.savez_compressed(os.path.join(outdir, "glue_attention_qwen2vl_fa2.npz"), **rec)
    # the SDPA patch (qwen2_vl.py:125-221): torch's own kernel on both sides, the 4-D mask sliced to the key length
    for attempt in range(3000):
        rec2 = attention_scenario(lc, qv.retake_Qwen2VLSdpaAttention_forward, False, 4000 + attempt)
        if rec2 is not None:
            break
    else:
        raise RuntimeError("sdpa")
    rec2["sdpa"] = True
    np.savez_compressed(os.path.join(outdir, "glue_attention_qwen2vl_sdpa.npz"), **rec2)
    print(f"glue_attention_qwen2vl_sdpa: seed {rec2['seed']}, cache lengths {[rec2[f'cache_k{l}'].shape[2] for l in range(2)]}")
    print(f"glue_attention_qwen2vl_fa2: seed {rec['seed']}, cache lengths {[rec[f'cache_k{l}'].shape[2] for l in range(2)]}, "
          f"evicted {rec['num_evicted'].tolist()}")


def main():
    vc, lc = G.import_reference()
    qv = G.import_reference_glue()
    import importlib

    lo = importlib.import_module("retake.llava_onevision")
    assert lo.__file__.startswith(G.REF)
    lo.eager_attention_forward = gs.eager_attention_forward_448   # the 4.48 function the reference was written against
    gen_qwen_forward(qv, HERE)
    gen_qwen_generate(qv, HERE)
    gen_qwen_image(qv, HERE)
    gen_qwen_prompt_guided(qv, HERE)
    gen_llava(lo, HERE)
    gen_llava_generate(lo, HERE)
    gen_attention(lc, qv, lo, HERE)
    gen_fa2_sliding_window(lc, qv, HERE)
    gen_llava_interface(lc, lo, HERE)


if __name__ == "__main__":
    main()
