"""Stub modules that stand in for the HuggingFace model parts around ReTaKe's glue (vision tower, projector,
language model, attention projections, rotary module).  Plain torch, independent of both the reference and the
product: the golden generator (tests/golden/gen_glue_golden.py) drives the REFERENCE's glue functions with them, the
tests drive the product's functions of the same names with the same stubs and compare what the language model /
the cache were handed.
"""
from __future__ import annotations

import math
import types

import numpy as np
import torch

import synth

VID, IMG, TXT = 151656, 151655, 7


class _Out(dict):
    """What the reference needs from a HF ModelOutput: item access by name and by position, slices, attributes."""

    def __init__(self, hidden, cache):
        super().__init__(last_hidden_state=hidden, past_key_values=cache)
        self.past_key_values, self.hidden_states, self.attentions = cache, None, None

    def __getitem__(self, k):
        if isinstance(k, (int, slice)):
            return (self["last_hidden_state"], self["past_key_values"])[k]
        return dict.__getitem__(self, k)


class StubLanguageModel:
    """Records every call: the tensors it was handed plus the cache flags at call time."""

    def __init__(self, hidden: int):
        self.hidden = hidden
        self.calls = []

    def embed_tokens(self, ids):
        return (ids.to(torch.float32)[..., None] % 97.0 / 97.0).repeat(1, 1, self.hidden)

    def __call__(self, **kw):
        cache = kw["past_key_values"]
        m = getattr(cache, "keypatches_mask_chunk", None)
        self.calls.append(dict(
            attention_mask=kw["attention_mask"].detach().cpu().clone(),
            position_ids=kw["position_ids"].detach().cpu().clone(),
            inputs_embeds=kw["inputs_embeds"].detach().cpu().clone(),
            cache_position=kw["cache_position"].detach().cpu().clone(),
            use_cache=bool(kw.get("use_cache")),
            kvcache_compression=bool(getattr(cache, "kvcache_compression", False)),
            mask=None if m is None else m.detach().cpu().clone()))
        return _Out(kw["inputs_embeds"] * 2.0, cache)


def calls_to_record(calls, prefix):
    rec = {prefix + "n": len(calls)}
    for i, c in enumerate(calls):
        p = f"{prefix}{i}_"
        for k in ("attention_mask", "position_ids", "inputs_embeds", "cache_position"):
            rec[p + k] = c[k].numpy()
        rec[p + "kvcache_compression"] = c["kvcache_compression"]
        rec[p + "use_cache"] = c["use_cache"]
        rec[p + "has_mask"] = c["mask"] is not None
        rec[p + "mask"] = c["mask"].numpy() if c["mask"] is not None else np.zeros(0, dtype=bool)
    return rec


def assert_calls_equal(calls, g, prefix, emb_tol=0.0):
    assert len(calls) == int(g[prefix + "n"])
    for i, c in enumerate(calls):
        p = f"{prefix}{i}_"
        for k in ("attention_mask", "position_ids", "cache_position"):
            np.testing.assert_array_equal(c[k].numpy(), g[p + k], err_msg=f"call {i} {k}")
        if emb_tol:
            np.testing.assert_allclose(c["inputs_embeds"].numpy(), g[p + "inputs_embeds"], rtol=0, atol=emb_tol)
        else:
            np.testing.assert_array_equal(c["inputs_embeds"].numpy(), g[p + "inputs_embeds"], err_msg=f"call {i} embeds")
        assert c["kvcache_compression"] == bool(g[p + "kvcache_compression"]), i
        assert c["use_cache"] == bool(g[p + "use_cache"]), i
        assert (c["mask"] is not None) == bool(g[p + "has_mask"]), i
        if c["mask"] is not None:
            np.testing.assert_array_equal(c["mask"].numpy(), g[p + "mask"], err_msg=f"call {i} key-patch mask")


# ---------------------------------------------------------------------------------------------------
# Qwen2-VL model stub (reference: qwen2_vl.py:522-764)
# ---------------------------------------------------------------------------------------------------
class StubVisual:
    """[grid_t*h*w, d] pixel rows -> [grid_t*h*w/4, C] merged embeddings, row-local (so frame chunking is exact)."""

    def __init__(self, d, C, seed=3):
        g = torch.Generator().manual_seed(seed)
        self.W = torch.randn(d, C, generator=g) / d ** 0.5
        self.calls = []

    def get_dtype(self):
        return torch.float32

    def __call__(self, pixel_values, grid_thw=None):
        self.calls.append(tuple(int(x) for x in grid_thw[0]))
        W = self.W.to(pixel_values.device)
        return pixel_values.reshape(-1, 4, pixel_values.shape[-1]).mean(1) @ W


def qwen_config(ratio=0.5, kv_ratio=0.5, chunk_frames=8, frame_chunk_size=None, sync=False, dynamic=None,
                prompt_guided=False):
    kv = {"compression_ratio": kv_ratio, "compression_method": "pivotkv", "pos_embed_reforge": True}
    if dynamic is not None:
        kv.update(dynamic_compression_ratio=True, max_input_length=dynamic)
    if prompt_guided:
        kv["prompt_guided_compression"] = True
    lk = {"chunked_prefill_frames": chunk_frames, "visual_compression": True,
          "visual_compression_kwargs": {"compression_ratio": ratio, "compression_method": "Keyframe",
                                        "patch_sync": sync, "return_keyframe_mask": True},
          "kvcache_compression": True, "kvcache_compression_kwargs": kv}
    if frame_chunk_size is not None:
        lk["frame_chunk_size"] = frame_chunk_size
    return types.SimpleNamespace(
        video_token_id=VID, image_token_id=IMG, vocab_size=1000,
        vision_config=types.SimpleNamespace(spatial_merge_size=2, temporal_patch_size=1),
        output_attentions=False, output_hidden_states=False, use_return_dict=True,
        hidden_size=64, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2,
        longvideo_kwargs=lk)


def make_qwen_model(mod, cfg, C=32, d=24):
    """`mod` = the module holding retake_Qwen2VLForConditionalGeneration_* (reference or product)."""
    me = types.SimpleNamespace(config=cfg, rope_deltas=None)
    for name in ("get_chunk_size", "segment_input_ids", "compress_video_tokens", "forge_input_chunks"):
        setattr(me, name, types.MethodType(getattr(mod, "retake_Qwen2VLForConditionalGeneration_" + name), me))
    me.visual = StubVisual(d, C)
    me.model = StubLanguageModel(C)
    me.lm_head = lambda h: h[..., :8]
    me.get_rope_index = None
    return me


def qwen_inputs(grid_t=16, gh=4, gw=4, n_pre=5, n_post=7, d=24, seed=77, device="cpu"):
    n_vid = grid_t * gh * gw // 4
    ids = torch.tensor([[TXT] * n_pre + [VID] * n_vid + [TXT + 1] * n_post])
    S = ids.shape[1]
    frames = synth.frames_video(seed, grid_t, gh * gw // 4, d)[0]                    # [T, N, d] video-like
    pix = torch.from_numpy(np.repeat(frames[:, :, None, :], 4, axis=2).reshape(-1, d).copy())   # 4 pixel rows per token
    t = torch.cat([torch.arange(n_pre), n_pre + torch.arange(grid_t).repeat_interleave(gh * gw // 4),
                   n_pre + grid_t + torch.arange(n_post)])
    h = torch.cat([torch.arange(n_pre), n_pre + torch.arange(gh // 2).repeat_interleave(gw // 2).repeat(grid_t),
                   n_pre + grid_t + torch.arange(n_post)])
    w = torch.cat([torch.arange(n_pre), n_pre + torch.arange(gw // 2).repeat(gh // 2 * grid_t),
                   n_pre + grid_t + torch.arange(n_post)])
    pos = torch.stack([t, h, w])[:, None]
    kw = dict(input_ids=ids, attention_mask=torch.ones(1, S, dtype=torch.long), position_ids=pos,
              pixel_values_videos=pix, video_grid_thw=torch.tensor([[grid_t, gh, gw]]), cache_position=torch.arange(S))
    return {k: v.to(device) for k, v in kw.items()}


def bare_hf_qwen2vl(cfg):
    """An instance of the installed transformers' Qwen2VLForConditionalGeneration WITHOUT running its __init__ (no
    weights): only `.config` set - what forge_input_chunks' prompt-guided branch asks of `self` (isinstance + config)."""
    from transformers.models.qwen2_vl.modeling_qwen2_vl import Qwen2VLForConditionalGeneration

    me = Qwen2VLForConditionalGeneration.__new__(Qwen2VLForConditionalGeneration)
    me.__dict__["config"] = cfg
    return me


def prompt_guided_case():
    """Inputs of the prompt-guided forge_input_chunks scenario: 3 text + 12 video + 5 text tokens, chunk [3, 7)."""
    S = 20
    seg = [(0, 3, "text"), (3, 15, "video"), (15, 20, "text")]
    pos = torch.arange(S)[None, None].repeat(3, 1, 1) + 100
    pos[1] += 7
    am = torch.ones(1, S, dtype=torch.long)
    am[0, 0] = 0
    return seg, torch.arange(S), pos, am, torch.arange(S * 2, dtype=torch.float32).reshape(1, S, 2)


def qwen_inputs_with_image(n_img=4, rows_per_token=4, device="cpu", **kw):
    """qwen_inputs with `n_img` image tokens inside the leading text segment and their pixel rows (the image branch of
    the forward, qwen2_vl.py:593-596, :631-645); rows_per_token != 4 makes features and tokens disagree."""
    out = qwen_inputs(n_pre=9, device=device, **kw)
    out["input_ids"][0, 2:2 + n_img] = IMG
    g = torch.Generator().manual_seed(5)
    out["pixel_values"] = torch.randn(n_img * rows_per_token, out["pixel_values_videos"].shape[-1], generator=g).to(device)
    out["image_grid_thw"] = torch.tensor([[1, 4, n_img * rows_per_token // 4]], device=device)
    return out


def qwen_generate_steps(mod, cfg, device="cpu", grid_t=24, seed=85, n_decode=2):
    """A `generate`-shaped call sequence on the stub model: the prefill forward WITHOUT position ids (they come from
    `get_rope_index`, stubbed to return the ids of qwen_inputs and their delta; qwen2_vl.py:573-590), then `n_decode`
    one-token forwards with the returned cache, no ids, cache_position continuing the UNCOMPRESSED prompt length like HF's
    generate does (ids = cache_position + rope_deltas), the last one with return_dict=True.  Returns (model, outputs)."""
    me = make_qwen_model(mod, cfg)
    kw = qwen_inputs(grid_t=grid_t, seed=seed, device=device)
    pos = kw.pop("position_ids")
    S = kw["input_ids"].shape[1]
    deltas = (pos.max() + 1 - S).reshape(1, 1)
    me.get_rope_index = lambda input_ids, image_grid_thw, video_grid_thw, attention_mask: (pos.clone(), deltas.clone())
    fwd = getattr(mod, "retake_Qwen2VLForConditionalGeneration_forward")
    outs = [fwd(me, return_dict=False, **kw)]
    cache = outs[0][1]
    for i in range(n_decode):
        ids = torch.tensor([[TXT + 2 + i]], device=device)
        am = torch.ones(1, S + i + 1, dtype=torch.long, device=device)
        outs.append(fwd(me, input_ids=ids, attention_mask=am, past_key_values=cache, use_cache=True,
                        cache_position=torch.tensor([S + i], device=device), return_dict=(i == n_decode - 1)))
    return me, outs


# ---------------------------------------------------------------------------------------------------
# LLaVA-OneVision model stub (reference: llava_onevision.py:306-583)
# ---------------------------------------------------------------------------------------------------
class StubVisionTower:
    """Frame f's pixels are the constant f; returns the f-th row block of a fixed [T, side*side, C] feature bank."""

    def __init__(self, bank):
        self.bank = bank
        self.calls = []

    def __call__(self, pixel_values, output_hidden_states=True):
        f = pixel_values[:, 0, 0, 0].round().long()
        self.calls.append(int(pixel_values.shape[0]))
        return types.SimpleNamespace(hidden_states=[None, self.bank.to(pixel_values.device)[f]])


def llava_config(ratio=0.5, kv_ratio=0.5, chunk_frames=4, frame_chunk_size=None, sync=False, side=4, dynamic=None):
    kv = {"compression_ratio": kv_ratio, "compression_method": "pivotkv", "pos_embed_reforge": True}
    if dynamic is not None:
        kv.update(dynamic_compression_ratio=True, max_input_length=dynamic)
    lk = {"chunked_prefill_frames": chunk_frames, "visual_compression": True,
          "visual_compression_kwargs": {"compression_ratio": ratio, "compression_method": "Keyframe",
                                        "patch_sync": sync, "return_keyframe_mask": True},
          "kvcache_compression": True, "kvcache_compression_kwargs": kv}
    if frame_chunk_size is not None:
        lk["frame_chunk_size"] = frame_chunk_size
    text = types.SimpleNamespace(hidden_size=64, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2)
    return types.SimpleNamespace(
        video_token_index=VID, image_token_index=IMG, text_config=text,
        vision_config=types.SimpleNamespace(image_size=14 * side, patch_size=14),
        vision_feature_layer=-1, vision_feature_select_strategy="full", vision_aspect_ratio="anyres_max_9",
        image_grid_pinpoints=None, output_attentions=False, output_hidden_states=False, use_return_dict=True,
        longvideo_kwargs=lk)


def make_llava_model(mod, cfg, bank):
    me = types.SimpleNamespace(config=cfg, pool_stride=2)
    for name in ("get_chunk_size", "segment_input_ids", "compress_video_tokens", "forge_input_chunks"):
        setattr(me, name, types.MethodType(getattr(mod, "retake_LlavaOnevisionForConditionalGeneration_" + name), me))
    C = bank.shape[-1]
    me.vision_tower = StubVisionTower(bank)
    me.multi_modal_projector = lambda x: x * 0.5
    side = cfg.vision_config.image_size // cfg.vision_config.patch_size

    def apply_pooling(x):   # [frames, side*side, C] -> [frames, ceil(side/2)^2, C], 2x2 mean (stand-in for HF's bilinear)
        f = x.shape[0]
        g = x.reshape(f, side, side, C).permute(0, 3, 1, 2)
        g = torch.nn.functional.avg_pool2d(g, 2, ceil_mode=True)
        return g.permute(0, 2, 3, 1).reshape(f, -1, C)

    me.apply_pooling = apply_pooling
    me.image_newline = torch.full((C,), 0.25)
    lm = StubLanguageModel(C)
    me.language_model = lm
    me.get_input_embeddings = lambda: lm.embed_tokens
    return me


def llava_inputs(T=12, side=4, n_pre=3, n_post=5, C=32, seed=78, device="cpu", pad_left=2):
    pooled = ((side + 1) // 2) ** 2
    n_vid = T * pooled + 1                       # frames * pooled tokens + the image_newline slot
    ids = torch.tensor([[TXT] * n_pre + [VID] * n_vid + [TXT + 1] * n_post])
    S = ids.shape[1]
    bank = torch.from_numpy(synth.frames_video(seed, T, side * side, C)[0].copy())
    pix = torch.arange(T, dtype=torch.float32)[None, :, None, None, None].expand(1, T, 3, 14 * side, 14 * side).contiguous()
    am = torch.ones(1, S, dtype=torch.long)
    am[0, :pad_left] = 0                         # distinguishes the reference's FRONT trim (:261) from a back trim
    kw = dict(input_ids=ids, attention_mask=am, position_ids=torch.arange(S)[None], pixel_values_videos=pix,
              cache_position=torch.arange(S))
    return {k: v.to(device) for k, v in kw.items()}, bank


def llava_generate_steps(mod, cfg, device="cpu", T=12, seed=86, n_decode=2):
    """A `generate`-shaped call sequence on the LLaVA stub model: the chunked prefill, then `n_decode` one-token forwards
    (input_ids.shape[1] == 1 selects the decode branch, llava_onevision.py:330-353, :548-560) with the returned cache, the
    ids HF's generate would hand over, the last one with return_dict=True.  Returns (model, outputs)."""
    kw, bank = llava_inputs(T=T, seed=seed, device=device)
    me = make_llava_model(mod, cfg, bank.to(device))
    me.image_newline = me.image_newline.to(device)
    S = kw["input_ids"].shape[1]
    fwd = getattr(mod, "retake_LlavaOnevisionForConditionalGeneration_forward")
    outs = [fwd(me, return_dict=False, **kw)]
    cache = outs[0][1]
    for i in range(n_decode):
        outs.append(fwd(me, input_ids=torch.tensor([[TXT + 2 + i]], device=device),
                        attention_mask=torch.ones(1, S + i + 1, dtype=torch.long, device=device),
                        position_ids=torch.tensor([[S + i]], device=device), past_key_values=cache, use_cache=True,
                        cache_position=torch.tensor([S + i], device=device), return_dict=(i == n_decode - 1)))
    return me, outs


# ---------------------------------------------------------------------------------------------------
# attention-module stubs (reference: qwen2_vl.py:42-122, llava_onevision.py:59-141)
# ---------------------------------------------------------------------------------------------------
class StubAttention(torch.nn.Module):
    """The attributes the patched attention forwards read, with explicit weights (stored in the fixtures)."""

    def __init__(self, layer_idx=0, hidden=64, heads=4, kv_heads=2, mrope=(2, 3, 3), scaling=1.0, weights=None, seed=0):
        super().__init__()
        self.num_heads, self.num_key_value_heads, self.head_dim = heads, kv_heads, hidden // heads
        self.num_key_value_groups = heads // kv_heads
        self.hidden_size, self.layer_idx, self.attention_dropout = hidden, layer_idx, 0.0
        self.scaling = self.head_dim ** -0.5
        self.is_causal = True
        self._flash_attn_uses_top_left_mask = False
        self.q_proj = torch.nn.Linear(hidden, hidden)
        self.k_proj = torch.nn.Linear(hidden, kv_heads * self.head_dim)
        self.v_proj = torch.nn.Linear(hidden, kv_heads * self.head_dim)
        self.o_proj = torch.nn.Linear(hidden, hidden, bias=False)
        if weights is None:
            g = torch.Generator().manual_seed(1000 + seed + layer_idx)
            with torch.no_grad():
                for p in self.parameters():
                    p.copy_(torch.randn(p.shape, generator=g) * (0.6 if p.ndim == 2 else 0.1))
        else:
            with torch.no_grad():
                for p, w in zip(self.parameters(), weights):
                    p.copy_(torch.as_tensor(w))
        self.rope_scaling = {"mrope_section": list(mrope)} if mrope else None
        self.rotary_emb = synth.RotaryStub(synth.inv_freq(self.head_dim, 1e4), scaling)
        self.config = types.SimpleNamespace(use_sliding_window=False, sliding_window=None, max_window_layers=0,
                                            _attn_implementation="eager")

    def weights(self):
        return [p.detach().cpu().numpy().copy() for p in self.parameters()]

    def to_device(self, dev):
        self.to(dev)
        self.rotary_emb.inv_freq = self.rotary_emb.inv_freq.to(dev)
        return self


def causal_mask(q_len, total, dtype=torch.float32):
    """[1,1,q_len,total] additive mask: query i (the last q_len of `total` positions) sees positions <= its own."""
    qpos = torch.arange(total - q_len, total)[:, None]
    m = torch.zeros(q_len, total, dtype=dtype)
    m[torch.arange(total)[None, :] > qpos] = float("-inf")
    return m[None, None]


def flash_attention_forward_stub(query_states, key_states, value_states, attention_mask, query_length, is_causal=True,
                                 dropout=0.0, sliding_window=None, use_top_left_mask=False, **kwargs):
    """Stand-in for transformers' `_flash_attention_forward` (third-party; needs the flash-attn package): the same
    contract on plain torch - inputs [b, s, heads, d], no padding mask (HF hands the FA2 path attention_mask=None for an
    all-ones mask), causal mask aligned bottom-right when the keys are longer than the queries, output
    [b, q_len, heads, d].  Used on BOTH sides: by the generator around the reference's FA2 patch and by the tests
    around this repo's."""
    assert attention_mask is None and sliding_window is None and not use_top_left_mask and dropout == 0.0
    q, k, v = (t.transpose(1, 2) for t in (query_states, key_states, value_states))
    ql, kl = q.shape[2], k.shape[2]
    assert ql == query_length
    mask = None
    if is_causal and ql > 1:
        qpos = torch.arange(kl - ql, kl, device=q.device)[:, None]
        mask = torch.arange(kl, device=q.device)[None, :] <= qpos
    w = torch.matmul(q, k.transpose(2, 3)) / math.sqrt(q.shape[-1])
    if mask is not None:
        w = w.masked_fill(~mask, float("-inf"))
    w = torch.softmax(w, dim=-1, dtype=torch.float32).to(q.dtype)
    return torch.matmul(w, v).transpose(1, 2)


INTERFACE_CALLS = []


def attention_interface_stub(module, query, key, value, attention_mask, dropout=0.0, scaling=None, sliding_window=None,
                             **kwargs):
    """An attention function to register in transformers' ALL_ATTENTION_FUNCTIONS under a test-only name: pins the
    non-eager dispatch of the LLaVA attention patch (llava_onevision.py:118-139) - what it is handed is recorded in
    INTERFACE_CALLS, the arithmetic is the 4.48 eager function's."""
    INTERFACE_CALLS.append((float(dropout), float(scaling), -1 if sliding_window is None else int(sliding_window),
                            sorted(kwargs)))
    return eager_attention_forward_448(module, query, key, value, attention_mask, scaling, dropout)


def eager_attention_forward_448(module, query, key, value, attention_mask, scaling, dropout=0.0, **kwargs):
    """transformers==4.48 `eager_attention_forward` of modeling_qwen2 (third-party, restated from its published source;
    the reference pins 4.48, environment.yaml:9): unlike 5.x it slices the 4-D mask to the key length, which is what
    makes the reference's compressed-cache attention run at all."""
    G = module.num_key_value_groups
    b, h, s, d = key.shape
    key_states = key[:, :, None].expand(b, h, G, s, d).reshape(b, h * G, s, d)
    value_states = value[:, :, None].expand(b, h, G, s, d).reshape(b, h * G, s, d)
    attn_weights = torch.matmul(query, key_states.transpose(2, 3)) * scaling
    if attention_mask is not None:
        attn_weights = attn_weights + attention_mask[:, :, :, : key_states.shape[-2]]
    attn_weights = torch.nn.functional.softmax(attn_weights, dim=-1, dtype=torch.float32).to(query.dtype)
    attn_weights = torch.nn.functional.dropout(attn_weights, p=dropout, training=module.training)
    attn_output = torch.matmul(attn_weights, value_states).transpose(1, 2).contiguous()
    return attn_output, attn_weights
