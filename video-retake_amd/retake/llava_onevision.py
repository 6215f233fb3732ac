"""LLaVA-OneVision / LLaVA-Video glue of ReTaKe on the MI355X build (reference: retake/llava_onevision.py).

Same roles as the Qwen2-VL glue, for `Qwen2Attention` (2-D position ids, plain RoPE, a rotary module per
layer) and `LlavaOnevisionForConditionalGeneration` (DPSelect runs on the PRE-projector SigLIP features,
2x2 pooling afterwards).  Names and signatures follow the reference so `patch_llava_onevision` rebinds
the same attributes.  Reference quirks that are reproduced, not fixed:
  * the key-patch mask has one entry per pre-pool patch (t*729) but is scattered over t*196+1 token
    slots; `masked_scatter` silently consumes only the first entries (reference :486);
  * the attention mask is trimmed from the FRONT after compression (reference :261).
"""
from __future__ import annotations

import math
from typing import Callable, Optional

import torch
import torch.nn as nn

from . import _prefill
from .longvideo_cache import apply_rotary_pos_emb, build_kvcache
from .qwen2_vl import _compress_memory_bank, _visual_compression_settings

DEBUG_MODE = False

__all__ = [
    "retake_Qwen2Attention_init", "retake_Qwen2Attention_forward",
    "retake_LlavaOnevisionForConditionalGeneration_get_chunk_size",
    "retake_LlavaOnevisionForConditionalGeneration_segment_input_ids",
    "retake_LlavaOnevisionForConditionalGeneration_compress_video_tokens",
    "retake_LlavaOnevisionForConditionalGeneration_forge_input_chunks",
    "retake_LlavaOnevisionForConditionalGeneration_forward",
]

try:  # captured at import, before any patching, exactly like the reference (llava_onevision.py:48)
    from transformers.models.qwen2.modeling_qwen2 import Qwen2Attention as _Qwen2Attention

    Qwen2Attention_original_init = _Qwen2Attention.__init__
except Exception:  # noqa: BLE001  (transformers missing or too old: patch_llava_onevision will fail loudly later)
    Qwen2Attention_original_init = None


def retake_Qwen2Attention_init(self, config, layer_idx: Optional[int] = None):
    """Every attention layer gets its own rotary module so RoPE can be recomputed from reforged ids
    (reference: llava_onevision.py:48-56)."""
    from transformers.models.qwen2.modeling_qwen2 import Qwen2RotaryEmbedding  # third-party

    Qwen2Attention_original_init(self, config, layer_idx)
    self.rotary_emb = Qwen2RotaryEmbedding(config=self.config)


def retake_Qwen2Attention_forward(self, hidden_states, position_embeddings, attention_mask, past_key_value=None,
                                  cache_position=None, **kwargs):
    """Qwen2 attention with PivotKV hooks (reference: llava_onevision.py:59-141)."""
    from transformers.modeling_utils import ALL_ATTENTION_FUNCTIONS  # third-party
    from transformers.models.qwen2.modeling_qwen2 import eager_attention_forward  # third-party

    input_shape = hidden_states.shape[:-1]
    hidden_shape = (*input_shape, -1, self.head_dim)
    query_states = self.q_proj(hidden_states).view(hidden_shape).transpose(1, 2)
    key_states = self.k_proj(hidden_states).view(hidden_shape).transpose(1, 2)
    value_states = self.v_proj(hidden_states).view(hidden_shape).transpose(1, 2)

    # current chunk's ids follow the reforged ids of the earlier chunks (reference :76-88); the ids are
    # cloned first so the shift of one layer does not leak into the next one
    position_ids = None
    fused = None
    reforge = past_key_value is not None and getattr(past_key_value, "pos_embed_reforge", False)
    if reforge and query_states.is_cuda and query_states.shape[0] == 1 and kwargs.get("position_ids") is not None \
            and hasattr(past_key_value, "update_pre_rope"):
        # on the GPU the whole prologue is one kernel (see qwen2_vl._qkv_and_cache_update): video chunks feed the
        # deferred scoring, text segments and decode steps just append; the ids the caller handed over stay as they
        # are, like the reference's clone
        fuse = past_key_value.update_pre_rope if getattr(past_key_value, "kvcache_compression", False) \
            else past_key_value.append_pre_rope
        fused = fuse(query_states, key_states, value_states, self.layer_idx, kwargs["position_ids"], self.rotary_emb, None,
                     shift_ids_in_place=False)
    if fused is not None:
        query_states, key_states, value_states = fused
    elif reforge:
        position_ids = kwargs.get("position_ids")
        if position_ids.is_cuda and hasattr(past_key_value, "shift_temporal_ids_"):
            # same rule on the device, no host sync (a zero shift leaves the clone equal to the ids)
            position_ids = past_key_value.shift_temporal_ids_(position_ids.clone(), self.layer_idx)
        else:
            prev_tempo_idx = past_key_value.get_prev_temporal_idx(self.layer_idx)
            cur_tempo_idx = position_ids[0, 0]
            if prev_tempo_idx + 1 != cur_tempo_idx:
                position_ids = position_ids.clone()
                position_ids[0, :] += prev_tempo_idx + 1 - cur_tempo_idx
        position_embeddings = None  # must be recomputed from the shifted ids
    if fused is None:
        if position_embeddings is None:
            cos, sin = self.rotary_emb(value_states, position_ids)
        else:
            cos, sin = position_embeddings
        query_states, key_states = apply_rotary_pos_emb(query_states, key_states, cos, sin)

    if fused is None and past_key_value is not None:
        cache_kwargs = {"sin": sin, "cos": cos, "cache_position": cache_position,
                        "query_states": query_states, "position_ids": position_ids, "rotary_emb": self.rotary_emb}
        key_states, value_states = past_key_value.update(key_states, value_states, self.layer_idx, cache_kwargs)

    sliding_window = None
    if (self.config.use_sliding_window and getattr(self.config, "sliding_window", None) is not None
            and self.layer_idx >= self.config.max_window_layers):
        sliding_window = self.config.sliding_window
    attention_interface: Callable = eager_attention_forward
    if self.config._attn_implementation != "eager":
        if not (self.config._attn_implementation == "sdpa" and kwargs.get("output_attentions", False)):
            attention_interface = ALL_ATTENTION_FUNCTIONS[self.config._attn_implementation]
    if attention_mask is not None and attention_mask.ndim == 4 and attention_mask.shape[-1] != key_states.shape[-2]:
        # the compressed cache is shorter than the mask HF built from the uncompressed prompt: transformers 4.48's
        # attention functions slice the mask to the key length themselves (what the reference relies on), later
        # versions do not - slicing here gives the 4.48 behaviour on both
        attention_mask = attention_mask[:, :, :, : key_states.shape[-2]]
    attn_output, attn_weights = attention_interface(
        self, query_states, key_states, value_states, attention_mask,
        dropout=0.0 if not self.training else self.attention_dropout, scaling=self.scaling,
        sliding_window=sliding_window, **kwargs)
    attn_output = self.o_proj(attn_output.reshape(*input_shape, -1).contiguous())
    return attn_output, attn_weights


def retake_LlavaOnevisionForConditionalGeneration_get_chunk_size(self, config, pixel_values_videos) -> Optional[int]:
    """min(chunk_frames, T) * ceil(H//patch / pool) * ceil(W//patch / pool) (reference :144-161)."""
    kwargs = getattr(config, "longvideo_kwargs", None)
    chunk_frames = kwargs.get("chunked_prefill_frames", None) if kwargs else None
    if chunk_frames is None:
        return None
    T, _, H, W = pixel_values_videos[0].shape
    H = math.ceil(H // self.config.vision_config.patch_size / self.pool_stride)
    W = math.ceil(W // self.config.vision_config.patch_size / self.pool_stride)
    return min(chunk_frames, T) * H * W


def retake_LlavaOnevisionForConditionalGeneration_segment_input_ids(self, input_ids):
    """[(s, e, 'video' | 'text')] (reference :164-198)."""
    return _prefill.segment_token_runs(input_ids[0] == self.config.video_token_index)


def retake_LlavaOnevisionForConditionalGeneration_compress_video_tokens(self, input_ids=None, attention_mask=None,
                                                                        selected_video_feature=None,
                                                                        position_ids=None, cache_position=None,
                                                                        labels=None):
    """DPSelect on the pre-projector SigLIP features [T, 729, 1152], then splice the token-level tensors
    to `tgt_grid_t * pooled_hw` video tokens (reference :201-269).  Returns (input_ids, attention_mask,
    selected_video_feature, position_ids, cache_position, tgt_grid_t, keypatches_mask)."""
    grid_t, grid_hw = selected_video_feature.shape[:2]
    settings = _visual_compression_settings(self.config)
    if settings is None:
        return input_ids, attention_mask, selected_video_feature, position_ids, cache_position, grid_t, None
    ratio, method, patch_sync, return_mask = settings
    assert labels is None
    assert input_ids.shape[0] == 1, "Currently, only inference are supported"
    video_positions = torch.where(input_ids[0] == self.config.video_token_index)[0]
    s_index, e_index = video_positions[0], video_positions[-1]
    side = self.config.vision_config.image_size // self.config.vision_config.patch_size
    pooled_hw = math.ceil(side / self.pool_stride) * math.ceil(side / self.pool_stride)
    ori_seq_len = input_ids.shape[1]
    tgt_grid_t = max(1, round(ratio * grid_t))

    bank, keypatches_mask = _compress_memory_bank(selected_video_feature.reshape(1, grid_t, grid_hw, -1), tgt_grid_t,
                                                  method, patch_sync, return_mask)
    selected_video_feature = bank[0]
    mem_len_after = tgt_grid_t * pooled_hw

    input_ids = torch.cat([input_ids[:, :s_index], input_ids[:, s_index:e_index + 1][:, :mem_len_after],
                           input_ids[:, e_index + 1:]], dim=1)
    num_token_diff = ori_seq_len - input_ids.shape[1]
    if num_token_diff and attention_mask is not None:
        attention_mask = attention_mask[:, num_token_diff:]   # from the front (reference :261)
    if num_token_diff and position_ids is not None:
        position_ids = position_ids[:, :-num_token_diff]
    if num_token_diff and cache_position is not None:
        cache_position = cache_position[:-num_token_diff]
    return input_ids, attention_mask, selected_video_feature, position_ids, cache_position, tgt_grid_t, keypatches_mask


def retake_LlavaOnevisionForConditionalGeneration_forge_input_chunks(self, ss, ee, modality_segments, position_ids,
                                                                     cache_position, attention_mask, past_key_values,
                                                                     inputs_embeds):
    """Per-chunk slices with 2-D position ids (reference :272-303)."""
    position_ids_chunk = position_ids[:, ss:ee]
    cache_position_chunk = cache_position[:ee]
    attention_mask_chunk = attention_mask[:, :ee]
    inputs_embeds_chunk = inputs_embeds[:, ss:ee]
    prompt_length = None
    if _prefill.prompt_guided(self.config):
        s_p, e_p, t_p = modality_segments[-1]
        assert t_p == "text"
        pos_offset = position_ids[0, s_p] - position_ids_chunk[0, -1] - 1
        position_ids_chunk = torch.cat([position_ids_chunk, position_ids[:, s_p:e_p] - pos_offset], dim=1)
        cache_position_chunk = torch.cat([cache_position_chunk, cache_position[s_p:e_p] - pos_offset], dim=0)
        attention_mask_chunk = torch.cat([attention_mask_chunk, attention_mask[:, s_p:e_p]], dim=1)
        inputs_embeds_chunk = torch.cat([inputs_embeds_chunk, inputs_embeds[:, s_p:e_p]], dim=1)
        prompt_length = e_p - s_p
    return position_ids_chunk, cache_position_chunk, attention_mask_chunk, inputs_embeds_chunk, prompt_length


def _vision_hidden_states(self, pixel_values_videos, vision_feature_layer):
    """SigLIP tower over frames in chunks of `frame_chunk_size` (reference :421-439)."""
    frame_chunk_size = (getattr(self.config, "longvideo_kwargs", None) or {}).get("frame_chunk_size", 1000000000)
    n = pixel_values_videos.shape[0]
    if n < frame_chunk_size:
        return self.vision_tower(pixel_values_videos, output_hidden_states=True).hidden_states[vision_feature_layer]
    pieces = [self.vision_tower(pixel_values_videos[i:i + frame_chunk_size],
                                output_hidden_states=True).hidden_states[vision_feature_layer]
              for i in range(0, n, frame_chunk_size)]
    return torch.cat(pieces)


def _scatter(inputs_embeds, input_ids, token_index, features):
    mask = (input_ids == token_index).unsqueeze(-1).expand_as(inputs_embeds).to(inputs_embeds.device)
    return inputs_embeds.masked_scatter(mask, features.to(inputs_embeds.device, inputs_embeds.dtype)), mask


def retake_LlavaOnevisionForConditionalGeneration_forward(
        self, input_ids=None, pixel_values=None, image_sizes=None, pixel_values_videos=None, image_sizes_videos=None,
        attention_mask=None, position_ids=None, past_key_values=None, inputs_embeds=None, vision_feature_layer=None,
        vision_feature_select_strategy=None, vision_aspect_ratio=None, labels=None, use_cache=None,
        output_attentions=None, output_hidden_states=None, return_dict=None, cache_position=None, logits_to_keep=0):
    """Chunked-prefill forward (reference: llava_onevision.py:306-583)."""
    from transformers.models.llava_onevision.modeling_llava_onevision import (  # third-party
        LlavaOnevisionCausalLMOutputWithPast, image_size_to_num_patches)

    assert input_ids.shape[0] == 1, "Batch inference of long video is not supported yet!"
    self.pool_stride = 2
    is_prefill = input_ids.shape[1] > 1
    chunk_size, modality_segments = None, None
    if is_prefill:
        chunk_size = self.get_chunk_size(self.config, pixel_values_videos)
        _prefill.apply_dynamic_compression_ratio(self.config, input_ids.shape[1])
        if chunk_size is not None:
            modality_segments = self.segment_input_ids(input_ids)
            past_key_values = build_kvcache(self.config, reserve_tokens=_prefill.expected_cache_tokens(
                self.config, input_ids.shape[1], chunk_size))
            use_cache = True

    cfg = self.config
    output_attentions = output_attentions if output_attentions is not None else cfg.output_attentions
    output_hidden_states = output_hidden_states if output_hidden_states is not None else cfg.output_hidden_states
    return_dict = return_dict if return_dict is not None else cfg.use_return_dict
    vision_feature_layer = vision_feature_layer if vision_feature_layer is not None else cfg.vision_feature_layer
    vision_feature_select_strategy = (vision_feature_select_strategy if vision_feature_select_strategy is not None
                                      else cfg.vision_feature_select_strategy)
    vision_aspect_ratio = vision_aspect_ratio if vision_aspect_ratio is not None else cfg.vision_aspect_ratio
    if (input_ids is None) ^ (inputs_embeds is not None):
        raise ValueError("You cannot specify both input_ids and inputs_embeds at the same time, and must specify "
                         "either one")
    if (pixel_values is not None or pixel_values_videos is not None) and inputs_embeds is not None:
        raise ValueError("You cannot specify both pixel_values/pixel_values_videos and inputs_embeds at the same "
                         "time, and must specify either one")

    image_features = None
    if pixel_values is not None:  # images: anyres patches (unchanged HF behaviour, reference :382-415)
        image_num_patches = [image_size_to_num_patches(image_size=s, grid_pinpoints=cfg.image_grid_pinpoints,
                                                       patch_size=cfg.vision_config.image_size) for s in image_sizes]
        if pixel_values.dim() == 5:
            pixel_values = torch.cat([pv[:n] for pv, n in zip(pixel_values, image_num_patches)], dim=0)
        elif pixel_values.dim() != 4:
            raise ValueError(f"pixel_values of shape {pixel_values.shape}, expect to be of 4 or 5 dimensions")
        feat = self.vision_tower(pixel_values, output_hidden_states=True).hidden_states[vision_feature_layer]
        if vision_feature_select_strategy == "default":
            feat = feat[:, 1:]
        feat = torch.split(self.multi_modal_projector(feat), image_num_patches, dim=0)
        image_features, _ = self.pack_image_features(feat, image_sizes, image_newline=self.image_newline,
                                                     vision_aspect_ratio=vision_aspect_ratio)

    keypatches_mask = None
    video_features = None
    if pixel_values_videos is not None:
        batch_size, frames, channels, height, width = pixel_values_videos.shape
        selected = _vision_hidden_states(self, pixel_values_videos.view(batch_size * frames, channels, height, width),
                                         vision_feature_layer)
        (input_ids, attention_mask, selected, position_ids, cache_position, frames,
         keypatches_mask) = self.compress_video_tokens(input_ids=input_ids, attention_mask=attention_mask,
                                                       selected_video_feature=selected, position_ids=position_ids,
                                                       cache_position=cache_position, labels=labels)
        if vision_feature_select_strategy == "default":
            selected = selected[:, 1:]
        video_features = self.apply_pooling(self.multi_modal_projector(selected))
        video_features = video_features.reshape(batch_size, frames * video_features.shape[1], -1)
        newline = self.image_newline[None, None, :].repeat(batch_size, 1, 1).to(video_features.device)
        video_features = torch.cat((video_features, newline), dim=1).flatten(0, 1)

    if inputs_embeds is None:
        inputs_embeds = self.get_input_embeddings()(input_ids)
    if image_features is not None:
        inputs_embeds, _ = _scatter(inputs_embeds, input_ids, cfg.image_token_index, image_features)
    if video_features is not None:
        inputs_embeds, vmask = _scatter(inputs_embeds, input_ids, cfg.video_token_index, video_features)
        if keypatches_mask is not None:  # NB: t*729 flags into t*196+1 slots, truncated (reference :486)
            keypatches_mask = torch.zeros_like(input_ids).bool().masked_scatter(vmask[:, :, 0], keypatches_mask)

    common = dict(use_cache=True, output_attentions=output_attentions, output_hidden_states=output_hidden_states,
                  return_dict=return_dict, logits_to_keep=logits_to_keep)
    if is_prefill and chunk_size is not None:
        assert past_key_values is not None
        cache = past_key_values

        def run_text(s, e):
            return self.language_model(attention_mask=attention_mask[:, :e], position_ids=position_ids[:, s:e],
                                       past_key_values=cache, inputs_embeds=inputs_embeds[:, s:e],
                                       cache_position=cache_position[:e], **common)

        def run_video_chunk(ss, ee):
            pos, cp, am, emb, prompt_length = self.forge_input_chunks(ss, ee, modality_segments, position_ids,
                                                                      cache_position, attention_mask, cache,
                                                                      inputs_embeds)
            if hasattr(cache, "before_forward"):
                cache.before_forward(prompt_length=prompt_length)
            out = self.language_model(attention_mask=am, position_ids=pos, past_key_values=cache, inputs_embeds=emb,
                                      cache_position=cp, **common)
            if hasattr(cache, "after_forward"):
                cache.after_forward()
            return out

        outputs = _prefill.run_chunked_prefill(modality_segments, chunk_size, cache, keypatches_mask, run_text,
                                               run_video_chunk)
    else:
        common["use_cache"] = use_cache
        outputs = self.language_model(attention_mask=attention_mask, position_ids=position_ids,
                                      past_key_values=past_key_values, inputs_embeds=inputs_embeds,
                                      cache_position=cache_position, **common)

    logits = outputs[0]
    loss = None
    if labels is not None:
        if attention_mask is not None:
            keep = attention_mask[..., 1:]
            shift_logits = logits[..., :-1, :][keep.to(logits.device) != 0].contiguous()
            shift_labels = labels[..., 1:][keep.to(labels.device) != 0].contiguous()
        else:
            shift_logits = logits[..., :-1, :].contiguous()
            shift_labels = labels[..., 1:].contiguous()
        loss = nn.CrossEntropyLoss()(shift_logits.view(-1, shift_logits.size(-1)),
                                     shift_labels.view(-1).to(shift_logits.device))
    if not return_dict:
        output = (logits,) + outputs[1:]
        return (loss,) + output if loss is not None else output
    return LlavaOnevisionCausalLMOutputWithPast(
        loss=loss, logits=logits, past_key_values=outputs.past_key_values, hidden_states=outputs.hidden_states,
        attentions=outputs.attentions, image_hidden_states=image_features if pixel_values is not None else None,
        video_hidden_states=video_features if pixel_values_videos is not None else None)
