"""Host-side glue against the reference goldens and its documented behaviour."""
import types

import numpy as np
import pytest
import torch

import golden_util as gu


def _me(cfg):
    return types.SimpleNamespace(config=cfg)


def _glue_cfg():
    return types.SimpleNamespace(
        video_token_id=151656, vision_config=types.SimpleNamespace(spatial_merge_size=2, temporal_patch_size=1),
        longvideo_kwargs={"chunked_prefill_frames": 8, "visual_compression": True,
                          "visual_compression_kwargs": {"compression_ratio": 0.5, "compression_method": "Keyframe",
                                                        "patch_sync": False, "return_keyframe_mask": True},
                          "kvcache_compression": True,
                          "kvcache_compression_kwargs": {"compression_ratio": 0.5, "compression_method": "pivotkv"}})


def test_segment_input_ids_matches_reference():
    import retake.qwen2_vl as q

    g = gu.load("glue_qwen2vl")
    me = _me(_glue_cfg())
    for ids_key, pre in (("ids", "seg"), ("ids2", "seg2")):
        seg = q.retake_Qwen2VLForConditionalGeneration_segment_input_ids(me, torch.from_numpy(g[ids_key]))
        assert [s for s, _, _ in seg] == g[pre + "_s"].tolist()
        assert [e for _, e, _ in seg] == g[pre + "_e"].tolist()
        assert [t for _, _, t in seg] == g[pre + "_t"].tolist()


def test_get_chunk_size_matches_reference():
    import retake.qwen2_vl as q

    g = gu.load("glue_qwen2vl")
    cfg = _glue_cfg()
    assert q.retake_Qwen2VLForConditionalGeneration_get_chunk_size(_me(cfg), cfg, torch.from_numpy(g["thw"])) == int(g["chunk_size"])
    cfg.longvideo_kwargs = {}
    assert q.retake_Qwen2VLForConditionalGeneration_get_chunk_size(_me(cfg), cfg, torch.from_numpy(g["thw"])) is None


def test_forge_input_chunks_matches_reference():
    import retake.qwen2_vl as q

    g = gu.load("glue_qwen2vl")
    S = g["ids"].shape[1]
    pos = torch.arange(S)[None, None].repeat(3, 1, 1)
    am = torch.ones(1, S, dtype=torch.long)
    cp = torch.arange(S)
    ie = torch.arange(S * 2, dtype=torch.float32).reshape(1, S, 2)
    seg = list(zip(g["seg_s"].tolist(), g["seg_e"].tolist(), g["seg_t"].tolist()))
    out = q.retake_Qwen2VLForConditionalGeneration_forge_input_chunks(_me(_glue_cfg()), 9, 21, seg, cp, pos, am, None, ie)
    for name, t in zip(["cp", "pos", "am", "ie"], out[:4]):
        np.testing.assert_array_equal(t.numpy(), g["fic_" + name])
    assert out[4] is None


def test_forge_input_chunks_prompt_guided():
    """Prompt-guided mode (qwen2_vl.py:500-517) against the reference: the trailing text segment appended to the chunk
    on an instance of the HF model class, NotImplementedError for any other `self`, off at compression_ratio 1."""
    import glue_stubs as gs
    import retake.qwen2_vl as q

    g = gu.load("glue_qwen2vl_prompt_guided")
    cfg = gs.qwen_config(ratio=0.5, prompt_guided=True)
    seg, cp, pos, am, ie = gs.prompt_guided_case()
    out = q.retake_Qwen2VLForConditionalGeneration_forge_input_chunks(gs.bare_hf_qwen2vl(cfg), 3, 7, seg, cp, pos, am, None, ie)
    for name, t in zip(["cp", "pos", "am", "ie"], out[:4]):
        np.testing.assert_array_equal(t.numpy(), g[name], err_msg=name)
    assert int(out[4]) == int(g["prompt_length"]) == 5
    assert out[1][0, 0, 4].item() == out[1][0, 0, 3].item() + 1   # prompt ids continue after the chunk's last temporal id
    assert str(g["other_class_exc"]) == "NotImplementedError"
    with pytest.raises(NotImplementedError):
        q.retake_Qwen2VLForConditionalGeneration_forge_input_chunks(_me(cfg), 3, 7, seg, cp, pos, am, None, ie)
    cfg1 = gs.qwen_config(ratio=0.5, kv_ratio=1, prompt_guided=True)
    out = q.retake_Qwen2VLForConditionalGeneration_forge_input_chunks(_me(cfg1), 3, 7, seg, cp, pos, am, None, ie)
    assert (out[4] is None) == bool(g["ratio1_prompt_length_is_none"])
    np.testing.assert_array_equal(out[0].numpy(), g["ratio1_cp"])


def test_llava_helpers():
    import retake.llava_onevision as lo

    cfg = types.SimpleNamespace(video_token_index=9, vision_config=types.SimpleNamespace(patch_size=14),
                                longvideo_kwargs={"chunked_prefill_frames": 32})
    me = types.SimpleNamespace(config=cfg, pool_stride=2)
    ids = torch.tensor([[1, 9, 9, 9, 2, 2]])
    assert lo.retake_LlavaOnevisionForConditionalGeneration_segment_input_ids(me, ids) == [(0, 1, "text"), (1, 4, "video"),
                                                                                        (4, 6, "text")]
    pv = torch.zeros(1, 64, 3, 384, 384)
    # min(32, 64) * ceil(27/2)^2 = 32 * 196
    assert lo.retake_LlavaOnevisionForConditionalGeneration_get_chunk_size(me, cfg, pv) == 32 * 196
    S = 12
    pos = torch.arange(S)[None]
    out = lo.retake_LlavaOnevisionForConditionalGeneration_forge_input_chunks(
        me, 2, 6, [(0, 2, "text"), (2, 10, "video"), (10, 12, "text")], pos, torch.arange(S), torch.ones(1, S),
        None, torch.zeros(1, S, 4))
    assert out[0].tolist() == [[2, 3, 4, 5]] and out[1].shape[0] == 6 and out[2].shape[1] == 6 and out[4] is None


def test_dynamic_compression_ratio():
    from retake import _prefill

    cfg = types.SimpleNamespace(longvideo_kwargs={"kvcache_compression": True, "kvcache_compression_kwargs": {
        "dynamic_compression_ratio": True, "max_input_length": 40000, "compression_ratio": 0.5}})
    _prefill.apply_dynamic_compression_ratio(cfg, 30000)
    assert cfg.longvideo_kwargs["kvcache_compression_kwargs"]["compression_ratio"] == 1
    _prefill.apply_dynamic_compression_ratio(cfg, 401409)
    assert cfg.longvideo_kwargs["kvcache_compression_kwargs"]["compression_ratio"] == 40000 / 401409


class _FakeCache:
    def __init__(self):
        self.kvcache_compression = True
        self.keypatches_mask_chunk = None
        self.log = []


def test_chunked_prefill_driver_order_and_mask_slices():
    from retake import _prefill

    cache = _FakeCache()
    mask = torch.zeros(1, 30, dtype=torch.bool)
    mask[0, 5:25:3] = True
    segs = [(0, 5, "text"), (5, 25, "video"), (25, 30, "text")]

    def run_text(s, e):
        cache.log.append(("text", s, e, cache.kvcache_compression, cache.keypatches_mask_chunk))
        return {"past_key_values": cache}

    def run_chunk(ss, ee):
        cache.log.append(("video", ss, ee, cache.kvcache_compression, cache.keypatches_mask_chunk.clone()))
        return {"past_key_values": cache}

    _prefill.run_chunked_prefill(segs, 8, cache, mask, run_text, run_chunk)
    kinds = [(k, s, e, c) for k, s, e, c, _ in cache.log]
    assert kinds == [("text", 0, 5, False), ("video", 5, 13, True), ("video", 13, 21, True), ("video", 21, 25, True),
                     ("text", 25, 30, False)]
    assert torch.equal(cache.log[1][4], mask[0, 5:13]) and torch.equal(cache.log[3][4], mask[0, 21:25])
    assert cache.keypatches_mask_chunk is None and cache.kvcache_compression is False  # off for decoding


def test_monkeypatch_surface():
    import retake.monkeypatch as mp

    with pytest.raises(NotImplementedError):
        mp.patch_qwen2vl("snapkv")
    with pytest.raises(NotImplementedError):
        mp.patch_llava_onevision("other")
    cfg = types.SimpleNamespace(rope_scaling={"type": "mrope", "mrope_section": [16, 24, 24]})
    exp = {"scaling_factor": 4, "longvideo_kwargs": {"kvcache_compression": True}}
    mp.patch_qwen2vl_config(cfg, exp)
    assert cfg.rope_scaling == {"mrope_section": [16, 24, 24], "rope_type": "yarn", "factor": 4, "beta_fast": 32.0,
                                "beta_slow": 1.0}
    assert cfg.longvideo_kwargs == {"kvcache_compression": True}
    cfg2 = types.SimpleNamespace(text_config=types.SimpleNamespace())
    mp.patch_llava_onevision_config(cfg2, exp)
    assert cfg2.text_config.rope_scaling["rope_type"] == "yarn" and cfg2.longvideo_kwargs["kvcache_compression"]
    cfg3 = types.SimpleNamespace(rope_scaling={"type": "mrope"})
    mp.patch_qwen2vl_config(cfg3, {})
    assert cfg3.rope_scaling == {"type": "mrope"} and cfg3.longvideo_kwargs == {}


def test_patch_rebinds_hf_classes():
    import transformers.models.qwen2_vl.modeling_qwen2_vl as m

    import retake.monkeypatch as mp
    import retake.qwen2_vl as q

    saved = {k: m.Qwen2VLForConditionalGeneration.__dict__.get(k) for k in
             ("forward", "compress_video_tokens", "segment_input_ids", "get_chunk_size", "forge_input_chunks")}
    att_saved = m.Qwen2VLAttention.forward
    try:
        mp.patch_qwen2vl("retake")
        assert m.Qwen2VLForConditionalGeneration.forward is q.retake_Qwen2VLForConditionalGeneration_forward
        assert m.Qwen2VLForConditionalGeneration.segment_input_ids is q.retake_Qwen2VLForConditionalGeneration_segment_input_ids
        assert m.Qwen2VLAttention.forward is q.retake_Qwen2VLAttention_forward
    finally:
        m.Qwen2VLAttention.forward = att_saved
        for k, v in saved.items():
            if v is None:
                delattr(m.Qwen2VLForConditionalGeneration, k)
            else:
                setattr(m.Qwen2VLForConditionalGeneration, k, v)


def test_patch_llava_rebinds_hf_classes_and_adds_rotary_module():
    """patch_llava_onevision on the installed transformers: the model class gets the five glue methods, Qwen2Attention the
    patched forward and an __init__ that gives every layer its own rotary module (llava_onevision.py:48-56) - checked by
    constructing a real (tiny) HF Qwen2Attention."""
    import transformers.models.llava_onevision.modeling_llava_onevision as ml
    import transformers.models.qwen2.modeling_qwen2 as m2
    from transformers import Qwen2Config

    import retake.llava_onevision as lo
    import retake.monkeypatch as mp

    names = ("forward", "compress_video_tokens", "segment_input_ids", "get_chunk_size", "forge_input_chunks")
    saved = {k: ml.LlavaOnevisionForConditionalGeneration.__dict__.get(k) for k in names}
    att_saved = (m2.Qwen2Attention.__init__, m2.Qwen2Attention.forward)
    try:
        mp.patch_llava_onevision("retake")
        assert ml.LlavaOnevisionForConditionalGeneration.forward is lo.retake_LlavaOnevisionForConditionalGeneration_forward
        assert ml.LlavaOnevisionForConditionalGeneration.get_chunk_size is lo.retake_LlavaOnevisionForConditionalGeneration_get_chunk_size
        assert m2.Qwen2Attention.forward is lo.retake_Qwen2Attention_forward
        cfg = Qwen2Config(hidden_size=64, num_attention_heads=4, num_key_value_heads=2, num_hidden_layers=1,
                          intermediate_size=128, vocab_size=32, max_position_embeddings=256)
        att = m2.Qwen2Attention(cfg, layer_idx=0)
        assert isinstance(att.rotary_emb, m2.Qwen2RotaryEmbedding)
        cos, sin = att.rotary_emb(torch.zeros(1, 5, 64), torch.arange(5)[None])
        assert cos.shape == (1, 5, 16) and float(getattr(att.rotary_emb, "attention_scaling", 1.0)) == 1.0
    finally:
        m2.Qwen2Attention.__init__, m2.Qwen2Attention.forward = att_saved
        for k, v in saved.items():
            if v is None:
                delattr(ml.LlavaOnevisionForConditionalGeneration, k)
            else:
                setattr(ml.LlavaOnevisionForConditionalGeneration, k, v)


# ---------------------------------------------------------------------------------------------------
# model forwards driven with stub modules against goldens recorded from the reference (tests/golden/gen_glue_golden.py).
# On CPU the DPSelect call inside compress_video_tokens is served by the CPU oracle (the product has no CPU path);
# tests/test_hip_parity.py runs the same scenarios with the HIP kernels on the GPU.
# ---------------------------------------------------------------------------------------------------
def _oracle_keyframe(memory_bank, tgt_mem_len, window_size=3, sync=True):
    from oracle import oracle as orc

    out, mask, _, _ = orc.dpselect(memory_bank.numpy(), tgt_mem_len, window_size, sync)
    return torch.from_numpy(out), torch.from_numpy(mask)


QWEN_FWD = {"base": (dict(ratio=0.5), dict(grid_t=24)),
            "fcs_sync": (dict(ratio=1.0, sync=True, frame_chunk_size=5, chunk_frames=6), dict(grid_t=16, seed=79)),
            "dynamic": (dict(ratio=0.5, dynamic=40), dict(grid_t=24, seed=80)),
            "dynamic_fits": (dict(ratio=0.5, dynamic=100000), dict(grid_t=24, seed=83))}   # ratio becomes the integer 1
LLAVA_FWD = {"base": (dict(ratio=0.5), dict(T=12)),
             "fcs_sync": (dict(ratio=1.0, sync=True, frame_chunk_size=5, chunk_frames=3), dict(T=10, seed=81)),
             "dynamic_odd": (dict(ratio=0.5, dynamic=30, side=5), dict(T=8, side=5, seed=82)),
             "dynamic_fits": (dict(ratio=0.5, dynamic=100000), dict(T=12, seed=84))}


def run_qwen_forward(name, device="cpu"):
    import glue_stubs as gs
    import retake.qwen2_vl as q

    ck, ik = QWEN_FWD[name]
    cfg = gs.qwen_config(**ck)
    me = gs.make_qwen_model(q, cfg)
    kw = gs.qwen_inputs(device=device, **ik)
    out = q.retake_Qwen2VLForConditionalGeneration_forward(me, return_dict=False, **kw)
    g = gu.load("glue_qwen2vl_forward_" + name)
    gs.assert_calls_equal(me.model.calls, g, "call", emb_tol=0.0 if device == "cpu" else 1e-6)
    np.testing.assert_allclose(out[0].cpu().numpy(), g["logits"], rtol=0, atol=0.0 if device == "cpu" else 1e-6)
    np.testing.assert_array_equal(np.array(me.visual.calls, dtype=np.int64).reshape(g["visual_calls"].shape), g["visual_calls"])
    # dynamic ratio is written back into the shared config dict and captured by the cache at construction (P0b)
    assert cfg.longvideo_kwargs["kvcache_compression_kwargs"]["compression_ratio"] == float(g["kv_ratio_after"])
    assert out[1].compression_ratio == float(g["cache_ratio"]) and type(out[1]).__name__ == str(g["cache_class"])
    assert out[1].kvcache_compression is False and out[1].keypatches_mask_chunk is None   # off for decoding


def run_qwen_image(device="cpu"):
    """Image tokens beside the video (qwen2_vl.py:593-596, :631-645): features scattered into the text segment, and the
    reference's ValueError - class and message - when features and tokens disagree."""
    import glue_stubs as gs
    import retake.qwen2_vl as q

    g = gu.load("glue_qwen2vl_forward_image")
    me = gs.make_qwen_model(q, gs.qwen_config(ratio=0.5))
    out = q.retake_Qwen2VLForConditionalGeneration_forward(me, return_dict=False,
                                                           **gs.qwen_inputs_with_image(grid_t=16, seed=87, device=device))
    tol = 0.0 if device == "cpu" else 1e-6
    gs.assert_calls_equal(me.model.calls, g, "call", emb_tol=tol)
    np.testing.assert_allclose(out[0].cpu().numpy(), g["logits"], rtol=0, atol=tol)
    np.testing.assert_array_equal(np.array(me.visual.calls, dtype=np.int64), g["visual_calls"])
    me = gs.make_qwen_model(q, gs.qwen_config(ratio=0.5))
    with pytest.raises(ValueError) as ei:
        q.retake_Qwen2VLForConditionalGeneration_forward(
            me, return_dict=False, **gs.qwen_inputs_with_image(grid_t=16, seed=87, rows_per_token=3, device=device))
    assert str(g["mismatch_exc"]) == "ValueError" and str(ei.value) == str(g["mismatch_msg"])
    assert not me.model.calls


def test_qwen2vl_forward_with_image_tokens_matches_reference(monkeypatch):
    import retake.qwen2_vl as q

    monkeypatch.setattr(q, "memory_bank_compress_keyframe", _oracle_keyframe)
    run_qwen_image()


def run_qwen_generate(device="cpu"):
    """Prefill without position ids (get_rope_index) + two decode forwards on the returned cache (qwen2_vl.py:543-590,
    :721-733) against the reference's call record: ids from cache_position + rope_deltas, cache passed through, output
    class and fields of the return_dict form."""
    import glue_stubs as gs
    import retake.qwen2_vl as q

    g = gu.load("glue_qwen2vl_generate")
    me, outs = gs.qwen_generate_steps(q, gs.qwen_config(ratio=0.5), device=device)
    tol = 0.0 if device == "cpu" else 1e-6
    gs.assert_calls_equal(me.model.calls, g, "call", emb_tol=tol)
    np.testing.assert_allclose(outs[0][0].cpu().numpy(), g["prefill_logits"], rtol=0, atol=tol)
    for i, o in enumerate(outs[1:-1]):
        np.testing.assert_allclose(o[0].cpu().numpy(), g[f"decode{i}_logits"], rtol=0, atol=tol)
    last = outs[-1]
    np.testing.assert_allclose(last.logits.cpu().numpy(), g["last_logits"], rtol=0, atol=tol)
    np.testing.assert_array_equal(last.rope_deltas.cpu().numpy(), g["last_rope_deltas"])
    assert type(last).__name__ == str(g["last_class"])
    assert (last.past_key_values is outs[0][1]) == bool(g["last_cache_is_prefill_cache"])
    np.testing.assert_array_equal(me.rope_deltas.cpu().numpy(), g["rope_deltas_attr"])


def test_qwen2vl_generate_sequence_matches_reference(monkeypatch):
    import retake.qwen2_vl as q

    monkeypatch.setattr(q, "memory_bank_compress_keyframe", _oracle_keyframe)
    run_qwen_generate()


def run_llava_generate(device="cpu"):
    """Chunked prefill + two decode forwards of the LLaVA glue (llava_onevision.py:330-353, :548-583) against the
    reference's call record; the return_dict form carries the reference's class and None-ness of its fields."""
    import glue_stubs as gs
    import retake.llava_onevision as lo

    g = gu.load("glue_llava_generate")
    me, outs = gs.llava_generate_steps(lo, gs.llava_config(ratio=0.5), device=device)
    tol = 0.0 if device == "cpu" else 1e-6
    gs.assert_calls_equal(me.language_model.calls, g, "call", emb_tol=tol)
    np.testing.assert_allclose(outs[0][0].cpu().numpy(), g["prefill_logits"], rtol=0, atol=tol)
    for i, o in enumerate(outs[1:-1]):
        np.testing.assert_allclose(o[0].cpu().numpy(), g[f"decode{i}_logits"], rtol=0, atol=tol)
    last = outs[-1]
    np.testing.assert_allclose(last.logits.cpu().numpy(), g["last_logits"], rtol=0, atol=tol)
    assert type(last).__name__ == str(g["last_class"])
    assert (last.past_key_values is outs[0][1]) == bool(g["last_cache_is_prefill_cache"])
    assert (last.video_hidden_states is None) == bool(g["last_video_hidden_states_is_none"])
    assert (last.image_hidden_states is None) == bool(g["last_image_hidden_states_is_none"])


def test_llava_generate_sequence_matches_reference(monkeypatch):
    import retake.qwen2_vl as q

    monkeypatch.setattr(q, "memory_bank_compress_keyframe", _oracle_keyframe)
    run_llava_generate()


def run_llava_forward(name, device="cpu"):
    import glue_stubs as gs
    import retake.llava_onevision as lo

    ck, ik = LLAVA_FWD[name]
    cfg = gs.llava_config(**ck)
    kw, bank = gs.llava_inputs(device=device, **ik)
    me = gs.make_llava_model(lo, cfg, bank.to(device))
    me.image_newline = me.image_newline.to(device)
    out = lo.retake_LlavaOnevisionForConditionalGeneration_forward(me, return_dict=False, **kw)
    g = gu.load("glue_llava_forward_" + name)
    gs.assert_calls_equal(me.language_model.calls, g, "call", emb_tol=0.0 if device == "cpu" else 1e-6)
    np.testing.assert_allclose(out[0].cpu().numpy(), g["logits"], rtol=0, atol=0.0 if device == "cpu" else 1e-6)
    np.testing.assert_array_equal(np.array(me.vision_tower.calls, dtype=np.int64), g["tower_calls"])
    assert cfg.longvideo_kwargs["kvcache_compression_kwargs"]["compression_ratio"] == float(g["kv_ratio_after"])
    assert out[1].compression_ratio == float(g["cache_ratio"])


@pytest.mark.parametrize("name", sorted(QWEN_FWD))
def test_qwen2vl_forward_driver_matches_reference(name, monkeypatch):
    """G6 (qwen2_vl.py:522-764): per text segment / video chunk the language model must be handed exactly the tensors,
    cache flags and key-patch mask slices the reference hands it (frame-chunked vision tower, dynamic ratio included)."""
    import retake.qwen2_vl as q

    monkeypatch.setattr(q, "memory_bank_compress_keyframe", _oracle_keyframe)
    run_qwen_forward(name)


@pytest.mark.parametrize("name", sorted(LLAVA_FWD))
def test_llava_forward_driver_matches_reference(name, monkeypatch):
    """G6 for LLaVA-Video (llava_onevision.py:306-583): front trim of the attention mask (:261), the t*side^2 ->
    t*pooled+1 key-patch mask truncation (:486), the dropped image_newline slot, 2-D ids."""
    import retake.qwen2_vl as q

    monkeypatch.setattr(q, "memory_bank_compress_keyframe", _oracle_keyframe)
    run_llava_forward(name)


def test_llava_glue_functions_match_reference(monkeypatch):
    """G2-G5 LLaVA twins (llava_onevision.py:144-303) against the reference's outputs."""
    import glue_stubs as gs
    import retake.llava_onevision as lo
    import retake.qwen2_vl as q

    monkeypatch.setattr(q, "memory_bank_compress_keyframe", _oracle_keyframe)
    g = gu.load("glue_llava")
    cfg = gs.llava_config(ratio=0.5)
    kw, bank = gs.llava_inputs(T=12)
    me = gs.make_llava_model(lo, cfg, bank)
    seg = me.segment_input_ids(kw["input_ids"])
    assert [int(s) for s, _, _ in seg] == g["seg_s"].tolist() and [int(e) for _, e, _ in seg] == g["seg_e"].tolist()
    assert [t for _, _, t in seg] == g["seg_t"].tolist()
    assert me.get_chunk_size(cfg, kw["pixel_values_videos"]) == int(g["chunk_size"])
    out = me.compress_video_tokens(input_ids=kw["input_ids"].clone(), attention_mask=kw["attention_mask"].clone(),
                                   selected_video_feature=bank.clone(), position_ids=kw["position_ids"].clone(),
                                   cache_position=kw["cache_position"].clone(), labels=None)
    for n, o in zip(["ids", "am", "feat", "pos", "cp"], out[:5]):
        np.testing.assert_array_equal(o.numpy(), g["cvt_" + n], err_msg=n)
    assert int(out[5]) == int(g["cvt_tgt"])
    np.testing.assert_array_equal(out[6].numpy(), g["cvt_mask"])
    assert g["cvt_am"][0, 0] == 1 and kw["attention_mask"][0, 0] == 0      # the fixture does tell front from back trim
    S = kw["input_ids"].shape[1]
    ie = torch.arange(S * 2, dtype=torch.float32).reshape(1, S, 2)
    o = me.forge_input_chunks(5, 17, seg, kw["position_ids"], kw["cache_position"], kw["attention_mask"], None, ie)
    for n, t in zip(["pos", "cp", "am", "ie"], o[:4]):
        np.testing.assert_array_equal(t.numpy(), g["fic_" + n])
    assert o[4] is None
    cfg.longvideo_kwargs["kvcache_compression_kwargs"]["prompt_guided_compression"] = True
    o = me.forge_input_chunks(5, 17, seg, kw["position_ids"] + 100, kw["cache_position"], kw["attention_mask"], None, ie)
    for n, t in zip(["pos", "cp", "am", "ie"], o[:4]):
        np.testing.assert_array_equal(t.numpy(), g["ficp_" + n])
    assert int(o[4]) == int(g["ficp_prompt_length"])


class _StubAttention(torch.nn.Module):
    def __init__(self, hidden=64, heads=4, kv_heads=2):
        super().__init__()
        import synth

        self.num_heads, self.num_key_value_heads, self.head_dim = heads, kv_heads, hidden // heads
        self.num_key_value_groups = heads // kv_heads
        self.hidden_size, self.layer_idx, self.attention_dropout = hidden, 0, 0.0
        self.q_proj = torch.nn.Linear(hidden, hidden)
        self.k_proj = torch.nn.Linear(hidden, kv_heads * self.head_dim)
        self.v_proj = torch.nn.Linear(hidden, kv_heads * self.head_dim)
        self.o_proj = torch.nn.Linear(hidden, hidden, bias=False)
        self.rope_scaling = {"mrope_section": [2, 3, 3]}
        self.rotary_emb = synth.RotaryStub(synth.inv_freq(self.head_dim), 1.0)


def test_qwen2vl_attention_patch_eager_equals_sdpa_and_appends_to_cache():
    import retake.longvideo_cache as lc
    import retake.qwen2_vl as q

    torch.manual_seed(0)
    att = _StubAttention().eval()
    x = torch.randn(1, 6, 64)
    pos = torch.arange(6)[None, None].repeat(3, 1, 1)
    causal = torch.full((6, 6), float("-inf")).triu(1)[None, None]
    c1, c2 = lc.DynamicCache(), lc.DynamicCache()
    with torch.no_grad():
        o1, w1, _ = q.retake_Qwen2VLAttention_forward(att, x, causal, pos.clone(), c1, output_attentions=True,
                                                      cache_position=torch.arange(6))
        o2, w2, _ = q.retake_Qwen2VLSdpaAttention_forward(att, x, causal, pos.clone(), c2,
                                                          cache_position=torch.arange(6))
    assert torch.allclose(o1, o2, atol=1e-5) and w1.shape == (1, 4, 6, 6) and w2 is None
    assert c1.key_cache[0].shape == (1, 2, 6, 16) and torch.equal(c1.key_cache[0], c2.key_cache[0])
