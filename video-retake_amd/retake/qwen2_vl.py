"""Qwen2-VL glue of ReTaKe on the MI355X build: attention patch, video-token compression and the
chunked-prefill forward.  Same function names and signatures as the reference's retake/qwen2_vl.py so
`retake.monkeypatch.patch_qwen2vl` rebinds the same HF attributes; the hot calls inside
(memory_bank_compress_keyframe, PivotKVCache.update) run as HIP kernels.

Targets the transformers==4.48 module layout the reference pins (environment.yaml:9); symbols that a
newer transformers no longer exports are resolved lazily so that importing this module never fails.
"""
from __future__ import annotations

import math
from typing import List, Optional, Tuple, Union

import torch
import torch.nn as nn

from . import _prefill
from .longvideo_cache import apply_multimodal_rotary_pos_emb, build_kvcache, repeat_kv
from .visual_compression import (memory_bank_compress_keyframe, memory_bank_compress_MALLM,
                                 memory_bank_compress_MALLM_hard, memory_bank_compress_MALLM_hard_to)

DEBUG_MODE = False

__all__ = [
    "retake_Qwen2VLAttention_forward", "retake_Qwen2VLSdpaAttention_forward",
    "retake_Qwen2VLFlashAttention2_forward", "retake_Qwen2VLForConditionalGeneration_compress_video_tokens",
    "retake_Qwen2VLForConditionalGeneration_segment_input_ids",
    "retake_Qwen2VLForConditionalGeneration_get_chunk_size",
    "retake_Qwen2VLForConditionalGeneration_forge_input_chunks", "retake_Qwen2VLForConditionalGeneration_forward",
]


# ---------------------------------------------------------------------------------------------------
# attention patch (reference: qwen2_vl.py:42-363).  The three HF attention classes share the prologue:
# projections -> temporal-id continuity shift -> RoPE computed inside the layer -> cache.update with the
# extra kwargs PivotKV needs.  Only the attention kernel itself differs.
# ---------------------------------------------------------------------------------------------------
def _qkv_and_cache_update(self, hidden_states, position_ids, past_key_value, cache_position):
    bsz, q_len, _ = hidden_states.size()
    query_states = self.q_proj(hidden_states).view(bsz, q_len, self.num_heads, self.head_dim).transpose(1, 2)
    key_states = self.k_proj(hidden_states).view(bsz, q_len, self.num_key_value_heads, self.head_dim).transpose(1, 2)
    value_states = self.v_proj(hidden_states).view(bsz, q_len, self.num_key_value_heads, self.head_dim).transpose(1, 2)

    reforge = past_key_value is not None and getattr(past_key_value, "pos_embed_reforge", False)
    if reforge and bsz == 1 and query_states.is_cuda and hasattr(past_key_value, "update_pre_rope"):
        # on the GPU the whole prologue is ONE kernel, straight from the projections (reference :68-86 + the cache's
        # update :238-259 / :319-321): shift, rotary tables, RoPE of q / k, the cache append, and - video chunks - the
        # operands of the deferred PivotKV scoring; text segments and decode steps append the shifted ids instead.
        # None = this call takes the op-by-op route below (nothing has been touched).
        fuse = past_key_value.update_pre_rope if getattr(past_key_value, "kvcache_compression", False) \
            else past_key_value.append_pre_rope
        fused = fuse(query_states, key_states, value_states, self.layer_idx, position_ids, self.rotary_emb,
                     self.rope_scaling["mrope_section"])
        if fused is not None:
            return fused

    # Position ids were reforged by the cache for earlier chunks: keep the temporal axis continuous
    # (reference :68-73).  In place, so later layers of this forward see the shifted ids too.
    if reforge:
        if position_ids.is_cuda and hasattr(past_key_value, "shift_temporal_ids_"):
            assert bsz == 1
            past_key_value.shift_temporal_ids_(position_ids, self.layer_idx)  # same rule on the device, no host sync
        else:
            prev_tempo_idx = past_key_value.get_prev_temporal_idx(self.layer_idx)
            if prev_tempo_idx + 1 != position_ids[0, 0, 0]:
                assert bsz == 1
                position_ids[0, 0, :] += prev_tempo_idx + 1 - position_ids[0, 0, 0]

    # RoPE is computed inside the layer so that reforged ids take effect (reference :75-79)
    cos, sin = self.rotary_emb(value_states, position_ids)
    mrope_section = self.rope_scaling["mrope_section"]
    query_states, key_states = apply_multimodal_rotary_pos_emb(query_states, key_states, cos, sin, mrope_section)

    if past_key_value is not None:
        cache_kwargs = {"sin": sin, "cos": cos, "cache_position": cache_position,
                        # PivotKV extras (reference :84-85)
                        "query_states": query_states, "position_ids": position_ids,
                        "rotary_emb": self.rotary_emb, "mrope_section": mrope_section}
        if reforge and hasattr(past_key_value, "shift_temporal_ids_"):
            # (build) this patch shifts `position_ids` in place for every layer anyway (above): the cache's update launch
            # may do the NEXT layer's shift on the way - not a key of the reference's protocol, ignored by other caches
            cache_kwargs["shift_next_position_ids"] = True
        key_states, value_states = past_key_value.update(key_states, value_states, self.layer_idx, cache_kwargs)
    return query_states, key_states, value_states


def retake_Qwen2VLAttention_forward(self, hidden_states, attention_mask=None, position_ids=None, past_key_value=None,
                                    output_attentions=False, use_cache=False, cache_position=None,
                                    position_embeddings=None):
    """Eager attention (reference: qwen2_vl.py:42-122)."""
    bsz, q_len, _ = hidden_states.size()
    query_states, key_states, value_states = _qkv_and_cache_update(self, hidden_states, position_ids, past_key_value,
                                                                   cache_position)
    key_states = repeat_kv(key_states, self.num_key_value_groups)
    value_states = repeat_kv(value_states, self.num_key_value_groups)
    attn_weights = torch.matmul(query_states, key_states.transpose(2, 3)) / math.sqrt(self.head_dim)
    if attention_mask is not None:  # whatever its length, slice it to the keys
        attn_weights = attn_weights + attention_mask[:, :, :, : key_states.shape[-2]]
    if query_states.dtype == torch.float16:  # fp16: +-inf logits would turn into NaN (reference :98-101)
        attn_weights = torch.where(torch.isinf(attn_weights), torch.zeros_like(attn_weights), attn_weights)
    attn_weights = nn.functional.softmax(attn_weights, dim=-1, dtype=torch.float32).to(query_states.dtype)
    attn_weights = nn.functional.dropout(attn_weights, p=self.attention_dropout, training=self.training)
    attn_output = torch.matmul(attn_weights, value_states)
    if attn_output.size() != (bsz, self.num_heads, q_len, self.head_dim):
        raise ValueError(f"`attn_output` should be of size {(bsz, self.num_heads, q_len, self.head_dim)}, but is"
                         f" {attn_output.size()}")
    attn_output = self.o_proj(attn_output.transpose(1, 2).contiguous().reshape(bsz, q_len, -1))
    return attn_output, (attn_weights if output_attentions else None), past_key_value


def retake_Qwen2VLSdpaAttention_forward(self, hidden_states, attention_mask=None, position_ids=None,
                                        past_key_value=None, output_attentions=False, use_cache=False,
                                        cache_position=None, position_embeddings=None):
    """SDPA attention (reference: qwen2_vl.py:125-221)."""
    if output_attentions:  # SDPA cannot return the weights: same fallback as the reference (:137-150)
        return retake_Qwen2VLAttention_forward(self, hidden_states, attention_mask, position_ids, past_key_value,
                                               output_attentions, use_cache, cache_position, position_embeddings)
    bsz, q_len, _ = hidden_states.size()
    query_states, key_states, value_states = _qkv_and_cache_update(self, hidden_states, position_ids, past_key_value,
                                                                   cache_position)
    key_states = repeat_kv(key_states, self.num_key_value_groups)
    value_states = repeat_kv(value_states, self.num_key_value_groups)
    causal_mask = attention_mask
    if attention_mask is not None:
        causal_mask = attention_mask[:, :, :, : key_states.shape[-2]]
    if query_states.device.type == "cuda" and attention_mask is not None:
        query_states, key_states, value_states = (t.contiguous() for t in (query_states, key_states, value_states))
    is_causal = causal_mask is None and q_len > 1
    attn_output = torch.nn.functional.scaled_dot_product_attention(
        query_states, key_states, value_states, attn_mask=causal_mask,
        dropout_p=self.attention_dropout if self.training else 0.0, is_causal=is_causal)
    attn_output = self.o_proj(attn_output.transpose(1, 2).contiguous().view(bsz, q_len, self.hidden_size))
    return attn_output, None, past_key_value


def retake_Qwen2VLFlashAttention2_forward(self, hidden_states, attention_mask=None, position_ids=None,
                                          past_key_value=None, output_attentions=False, use_cache=False,
                                          cache_position=None, position_embeddings=None):
    """FlashAttention-2 variant (reference: qwen2_vl.py:224-363): same prologue, HF's flash-attention
    helper for the kernel (third-party; present when flash-attn for ROCm is installed)."""
    from transformers.modeling_flash_attention_utils import _flash_attention_forward  # third-party

    bsz, q_len, _ = hidden_states.size()
    if past_key_value is not None and getattr(self.config, "use_sliding_window", False) \
            and getattr(self.config, "sliding_window", None) is not None:
        # sliding-window models (Qwen2-VL ships use_sliding_window = false): the reference trims the padding mask to the
        # window before the cache update when the cache already holds tokens (reference :268-294; the sliced past
        # keys / values themselves are not used there either, only their length is checked)
        prev_len = past_key_value.get_seq_length(self.layer_idx)
        if q_len + prev_len > self.config.sliding_window and prev_len > 0:
            slicing_tokens = 1 - self.config.sliding_window
            if min(prev_len, -slicing_tokens) != self.config.sliding_window - 1:
                raise ValueError("past key must have a shape of (`batch_size, num_heads, self.config.sliding_window-1, "
                                 f"head_dim`), got a past length of {prev_len}")
            if attention_mask is not None:
                attention_mask = attention_mask[:, slicing_tokens:]
                attention_mask = torch.cat([attention_mask, torch.ones_like(attention_mask[:, -1:])], dim=-1)
    query_states, key_states, value_states = _qkv_and_cache_update(self, hidden_states, position_ids, past_key_value,
                                                                   cache_position)
    key_states = repeat_kv(key_states, self.num_key_value_groups)
    value_states = repeat_kv(value_states, self.num_key_value_groups)
    dropout_rate = 0.0 if not self.training else self.attention_dropout
    input_dtype = query_states.dtype
    if input_dtype == torch.float32:  # flash-attn wants half precision (reference :318-333)
        if torch.is_autocast_enabled():
            target_dtype = torch.get_autocast_gpu_dtype()
        elif hasattr(self.config, "_pre_quantization_dtype"):
            target_dtype = self.config._pre_quantization_dtype
        else:
            target_dtype = self.q_proj.weight.dtype
        query_states, key_states, value_states = (t.to(target_dtype) for t in (query_states, key_states, value_states))
    query_states, key_states, value_states = (t.transpose(1, 2) for t in (query_states, key_states, value_states))
    sliding_window = None
    if (getattr(self.config, "use_sliding_window", False) and getattr(self.config, "sliding_window", None) is not None
            and self.layer_idx >= self.config.max_window_layers):
        sliding_window = self.config.sliding_window
    attn_output = _flash_attention_forward(query_states, key_states, value_states, attention_mask, q_len,
                                           dropout=dropout_rate, sliding_window=sliding_window,
                                           is_causal=self.is_causal,
                                           use_top_left_mask=getattr(self, "_flash_attn_uses_top_left_mask", False))
    attn_output = self.o_proj(attn_output.reshape(bsz, q_len, self.hidden_size).contiguous())
    return attn_output, None, past_key_value


# ---------------------------------------------------------------------------------------------------
# video-token compression (reference: qwen2_vl.py:366-442)
# ---------------------------------------------------------------------------------------------------
def _visual_compression_settings(config):
    kwargs = getattr(config, "longvideo_kwargs", None)
    if kwargs is None or not kwargs.get("visual_compression", False):
        return None
    ck = kwargs["visual_compression_kwargs"]
    return ck.get("compression_ratio"), ck.get("compression_method"), ck.get("patch_sync"), ck.get("return_keyframe_mask")


def _compress_memory_bank(bank, tgt_len, method, patch_sync, return_mask):
    """Dispatch on `compression_method` (reference: qwen2_vl.py:402-416)."""
    if method == "MA-LLM":
        size = torch.ones_like(bank[:, :, :, 0])
        while bank.shape[1] > tgt_len:
            bank, size = memory_bank_compress_MALLM(bank, size, sync=patch_sync)
        return bank, None
    if method == "MA-LLM-hard":
        if bank.is_cuda:   # the reference's loop of single steps (:406-408) in one pass over the bank, same values
            return memory_bank_compress_MALLM_hard_to(bank, tgt_len, sync=patch_sync), None
        while bank.shape[1] > tgt_len:
            bank = memory_bank_compress_MALLM_hard(bank, sync=patch_sync)
        return bank, None
    if method == "Keyframe":
        bank, mask = memory_bank_compress_keyframe(bank, tgt_len, 3, sync=patch_sync)  # DPSelect (HIP)
        return bank, (mask if return_mask else None)
    raise NotImplementedError


def retake_Qwen2VLForConditionalGeneration_compress_video_tokens(self, input_ids=None, attention_mask=None,
                                                                 video_embeds=None, cache_position=None,
                                                                 position_ids=None, labels=None, video_grid_thw=None):
    """DPSelect on the video embeddings, then splice ids / mask / cache_position / position ids down to
    the kept tokens.  Returns (input_ids, attention_mask, video_embeds, cache_position, position_ids,
    labels, keypatches_mask)."""
    settings = _visual_compression_settings(self.config)
    if settings is None:
        return input_ids, attention_mask, video_embeds, cache_position, position_ids, labels, None
    ratio, method, patch_sync, return_mask = settings
    assert labels is None
    assert video_grid_thw.shape[0] <= 1, "Currently, interleaved videos are not supported"
    assert input_ids.shape[0] == 1, "Currently, only inference are supported"
    video_positions = torch.where(input_ids[0] == self.config.video_token_id)[0]
    s_index, e_index = video_positions[0], video_positions[-1]
    grid_t = video_grid_thw[0][0]
    grid_hw = video_embeds.shape[0] // grid_t
    ori_seq_len = input_ids.shape[1]
    tgt_mem_len = max(1, round(ratio * grid_t.item()))  # python banker's rounding, as the reference (:397)
    num_frame_diff = grid_t - tgt_mem_len

    bank, keypatches_mask = _compress_memory_bank(video_embeds.reshape(1, grid_t, grid_hw, -1), tgt_mem_len, method,
                                                  patch_sync, return_mask)
    video_embeds = bank.flatten(1, 2)[0]
    tgt_seq_len = video_embeds.shape[0]

    input_ids = torch.cat([input_ids[:, :s_index], input_ids[:, s_index:e_index + 1][:, :tgt_seq_len],
                           input_ids[:, e_index + 1:]], dim=1)
    num_token_diff = ori_seq_len - input_ids.shape[1]
    if num_token_diff and attention_mask is not None:
        attention_mask = attention_mask[:, :-num_token_diff]
    if num_token_diff and cache_position is not None:
        cache_position = cache_position[:-num_token_diff]
    if position_ids is not None:
        position_ids = torch.cat([position_ids[..., :s_index], position_ids[..., s_index:e_index + 1][..., :tgt_seq_len],
                                  position_ids[..., e_index + 1:]], dim=2)
        position_ids[:, :, s_index + tgt_seq_len:] -= num_frame_diff  # everything after the video moves up
    return input_ids, attention_mask, video_embeds, cache_position, position_ids, labels, keypatches_mask


def retake_Qwen2VLForConditionalGeneration_segment_input_ids(self, input_ids):
    """[(s, e, 'video' | 'text')], end exclusive, sorted (reference: qwen2_vl.py:444-475)."""
    return _prefill.segment_token_runs(input_ids[0] == self.config.video_token_id)


def retake_Qwen2VLForConditionalGeneration_get_chunk_size(self, config, video_grid_thw) -> Optional[int]:
    """Tokens per prefill chunk = min(chunk_frames, T)*H*W // (merge^2 * temporal_patch) (reference :477-491)."""
    kwargs = getattr(config, "longvideo_kwargs", None)
    chunk_frames = kwargs.get("chunked_prefill_frames", None) if kwargs else None
    if chunk_frames is None:
        return None
    T, H, W = video_grid_thw[0]
    t_factor = config.vision_config.spatial_merge_size ** 2 * config.vision_config.temporal_patch_size
    return int((min(chunk_frames, T) * H * W // t_factor).item())


def retake_Qwen2VLForConditionalGeneration_forge_input_chunks(self, ss, ee, modality_segments, cache_position,
                                                              position_ids, attention_mask, past_key_values,
                                                              inputs_embeds):
    """Slices of one prefill chunk; the attention mask and cache_position run from 0 to `ee`
    (reference: qwen2_vl.py:493-519).  Prompt-guided mode appends the trailing text segment."""
    cache_position_chunk = cache_position[:ee]
    position_ids_chunk = position_ids[:, :, ss:ee]
    attention_mask_chunk = attention_mask[:, :ee]
    inputs_embeds_chunk = inputs_embeds[:, ss:ee]
    prompt_length = None
    if _prefill.prompt_guided(self.config):
        s_p, e_p, t_p = modality_segments[-1]
        # only '<|vision_pad|>' counts as vision for Qwen2-VL, so the trailing segment starts at '<|vision_end|>'; any
        # other model class is refused (reference :506-511)
        from transformers.models.qwen2_vl.modeling_qwen2_vl import Qwen2VLForConditionalGeneration  # third-party

        if not isinstance(self, Qwen2VLForConditionalGeneration):
            raise NotImplementedError
        assert t_p == "text"
        pos_offset = position_ids[0, 0, s_p] - position_ids_chunk[0, 0, -1] - 1
        position_ids_chunk = torch.cat([position_ids_chunk, position_ids[:, :, s_p:e_p] - pos_offset], dim=2)
        attention_mask_chunk = torch.cat([attention_mask_chunk, attention_mask[:, s_p:e_p]], dim=1)
        inputs_embeds_chunk = torch.cat([inputs_embeds_chunk, inputs_embeds[:, s_p:e_p]], dim=1)
        prompt_length = e_p - s_p
        cache_position_chunk = cache_position[:ee + prompt_length]
    return cache_position_chunk, position_ids_chunk, attention_mask_chunk, inputs_embeds_chunk, prompt_length


# ---------------------------------------------------------------------------------------------------
# model forward with chunked prefill (reference: qwen2_vl.py:522-764)
# ---------------------------------------------------------------------------------------------------
def _encode_video(self, pixel_values_videos, video_grid_thw):
    """Vision tower in frame chunks of `frame_chunk_size` (reference: qwen2_vl.py:597-617)."""
    pixel_values_videos = pixel_values_videos.type(self.visual.get_dtype())
    grid_t, grid_h, grid_w = video_grid_thw[0]
    frame_chunk_size = (getattr(self.config, "longvideo_kwargs", None) or {}).get("frame_chunk_size", 1000000000)
    if grid_t < frame_chunk_size:
        return self.visual(pixel_values_videos, grid_thw=video_grid_thw)
    d = pixel_values_videos.shape[-1]
    frames = pixel_values_videos.reshape(grid_t, grid_h * grid_w, d)
    pieces = []
    for i in range(0, grid_t, frame_chunk_size):
        chunk = frames[i:i + frame_chunk_size]
        thw = video_grid_thw.clone()
        thw[0, 0] = chunk.shape[0]
        pieces.append(self.visual(chunk.reshape(-1, d), grid_thw=thw))
    return torch.cat(pieces)


def _scatter_features(inputs_embeds, input_ids, token_id, features, what):
    n_tokens = (input_ids == token_id).sum().item()
    if n_tokens != features.shape[0]:
        raise ValueError(f"{what} features and {what.lower()} tokens do not match: tokens: {n_tokens}, "
                         f"features {features.shape[0]}")
    mask = (input_ids == token_id).unsqueeze(-1).expand_as(inputs_embeds).to(inputs_embeds.device)
    return inputs_embeds.masked_scatter(mask, features.to(inputs_embeds.device, inputs_embeds.dtype)), mask


def retake_Qwen2VLForConditionalGeneration_forward(
        self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None, inputs_embeds=None,
        labels=None, use_cache=None, output_attentions=None, output_hidden_states=None, return_dict=None,
        pixel_values=None, pixel_values_videos=None, image_grid_thw=None, video_grid_thw=None, rope_deltas=None,
        cache_position=None):
    assert input_ids.shape[0] == 1, "Batch inference of long video is not supported yet!"
    is_prefill = cache_position is not None and cache_position[0] == 0
    chunk_size, modality_segments = None, None
    if is_prefill:
        chunk_size = self.get_chunk_size(self.config, video_grid_thw)
        _prefill.apply_dynamic_compression_ratio(self.config, input_ids.shape[1])
        if chunk_size is not None:
            modality_segments = self.segment_input_ids(input_ids)
            past_key_values = build_kvcache(self.config, reserve_tokens=_prefill.expected_cache_tokens(
                self.config, input_ids.shape[1], chunk_size))
            use_cache = True

    output_attentions = output_attentions if output_attentions is not None else self.config.output_attentions
    output_hidden_states = (output_hidden_states if output_hidden_states is not None
                            else self.config.output_hidden_states)
    return_dict = return_dict if return_dict is not None else self.config.use_return_dict

    if position_ids is None and (attention_mask is None or attention_mask.ndim == 2):
        if is_prefill or self.rope_deltas is None:  # RoPE index once per generation
            position_ids, rope_deltas = self.get_rope_index(input_ids, image_grid_thw, video_grid_thw, attention_mask)
            self.rope_deltas = rope_deltas
        else:  # decode: continue from the stored deltas
            batch_size, seq_length = input_ids.shape
            delta = cache_position[0] + self.rope_deltas if cache_position is not None else 0
            position_ids = torch.arange(seq_length, device=input_ids.device).view(1, -1).expand(batch_size, -1)
            if cache_position is not None:
                delta = delta.repeat_interleave(batch_size // delta.shape[0], dim=0)
            position_ids = position_ids.add(delta).unsqueeze(0).expand(3, -1, -1)

    keypatches_mask = None
    if inputs_embeds is None:
        image_embeds = video_embeds = None
        if pixel_values is not None:
            image_embeds = self.visual(pixel_values.type(self.visual.get_dtype()), grid_thw=image_grid_thw)
        if pixel_values_videos is not None:
            video_embeds = _encode_video(self, pixel_values_videos, video_grid_thw)
            (input_ids, attention_mask, video_embeds, cache_position, position_ids, labels,
             keypatches_mask) = self.compress_video_tokens(
                input_ids=input_ids, attention_mask=attention_mask, video_embeds=video_embeds,
                cache_position=cache_position, position_ids=position_ids, labels=labels,
                video_grid_thw=video_grid_thw)
        inputs_embeds = self.model.embed_tokens(input_ids)
        if image_embeds is not None:
            inputs_embeds, _ = _scatter_features(inputs_embeds, input_ids, self.config.image_token_id, image_embeds,
                                                 "Image")
        if video_embeds is not None:
            inputs_embeds, video_mask = _scatter_features(inputs_embeds, input_ids, self.config.video_token_id,
                                                          video_embeds, "Video")
            if keypatches_mask is not None:  # key-patch flags at their token positions (reference :662-663)
                keypatches_mask = torch.zeros_like(input_ids).bool().masked_scatter(video_mask[:, :, 0],
                                                                                   keypatches_mask)
        if attention_mask is not None:
            attention_mask = attention_mask.to(inputs_embeds.device)
        if position_ids is not None:
            position_ids = position_ids.to(inputs_embeds.device)

    common = dict(input_ids=None, use_cache=True, output_attentions=output_attentions,
                  output_hidden_states=output_hidden_states, return_dict=return_dict)
    if is_prefill and chunk_size is not None:
        assert past_key_values is not None
        cache = past_key_values

        def run_text(s, e):
            return self.model(position_ids=position_ids[:, :, s:e], attention_mask=attention_mask[:, :e],
                              past_key_values=cache, inputs_embeds=inputs_embeds[:, s:e],
                              cache_position=cache_position[:e], **common)

        def run_video_chunk(ss, ee):
            cp, pos, am, emb, prompt_length = self.forge_input_chunks(ss, ee, modality_segments, cache_position,
                                                                      position_ids, attention_mask, cache,
                                                                      inputs_embeds)
            if hasattr(cache, "before_forward"):
                cache.before_forward(prompt_length=prompt_length)
            out = self.model(position_ids=pos, attention_mask=am, past_key_values=cache, inputs_embeds=emb,
                             cache_position=cp, **common)
            if hasattr(cache, "after_forward"):
                cache.after_forward()
            return out

        outputs = _prefill.run_chunked_prefill(modality_segments, chunk_size, cache, keypatches_mask, run_text,
                                               run_video_chunk)
    else:  # decode / ordinary prefill
        common["use_cache"] = use_cache
        outputs = self.model(position_ids=position_ids, attention_mask=attention_mask,
                             past_key_values=past_key_values, inputs_embeds=inputs_embeds,
                             cache_position=cache_position, **common)

    hidden_states = outputs[0]
    logits = self.lm_head(hidden_states)
    loss = None
    if labels is not None:
        logits = logits.float()
        shift_logits = logits[..., :-1, :].contiguous().view(-1, self.config.vocab_size)
        shift_labels = labels[..., 1:].contiguous().view(-1).to(shift_logits.device)
        loss = nn.CrossEntropyLoss()(shift_logits, shift_labels)
    if not return_dict:
        output = (logits,) + outputs[1:]
        return (loss,) + output if loss is not None else output
    from transformers.models.qwen2_vl.modeling_qwen2_vl import Qwen2VLCausalLMOutputWithPast  # third-party

    return Qwen2VLCausalLMOutputWithPast(loss=loss, logits=logits, past_key_values=outputs.past_key_values,
                                         hidden_states=outputs.hidden_states, attentions=outputs.attentions,
                                         rope_deltas=self.rope_deltas)
