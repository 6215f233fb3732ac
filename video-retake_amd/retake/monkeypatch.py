"""Plugin surface: rebinds HF model methods to the ReTaKe versions (reference: retake/monkeypatch.py).

`patch_qwen2vl("retake")` / `patch_llava_onevision("retake")` keep the reference's names, its
"retake"-only method check (NotImplementedError otherwise) and its config patchers (YaRN rope scaling,
`config.longvideo_kwargs`).  HF attention class names differ between transformers releases (the
reference pins 4.48); classes that do not exist in the installed release are skipped.
"""
from __future__ import annotations

import importlib

from .llava_onevision import (
    retake_LlavaOnevisionForConditionalGeneration_compress_video_tokens,
    retake_LlavaOnevisionForConditionalGeneration_forge_input_chunks,
    retake_LlavaOnevisionForConditionalGeneration_forward,
    retake_LlavaOnevisionForConditionalGeneration_get_chunk_size,
    retake_LlavaOnevisionForConditionalGeneration_segment_input_ids,
    retake_Qwen2Attention_forward,
    retake_Qwen2Attention_init,
)
from .qwen2_vl import (
    retake_Qwen2VLAttention_forward,
    retake_Qwen2VLFlashAttention2_forward,
    retake_Qwen2VLForConditionalGeneration_compress_video_tokens,
    retake_Qwen2VLForConditionalGeneration_forge_input_chunks,
    retake_Qwen2VLForConditionalGeneration_forward,
    retake_Qwen2VLForConditionalGeneration_get_chunk_size,
    retake_Qwen2VLForConditionalGeneration_segment_input_ids,
    retake_Qwen2VLSdpaAttention_forward,
)


def patch_qwen2vl_config(config, exp_configs):
    """YaRN rope scaling (factor = scaling_factor, beta_fast 32, beta_slow 1) + longvideo_kwargs
    (reference: monkeypatch.py:24-34)."""
    if "scaling_factor" in exp_configs:
        config.rope_scaling.pop("type", None)
        config.rope_scaling["rope_type"] = "yarn"
        config.rope_scaling["factor"] = exp_configs["scaling_factor"]
        config.rope_scaling["beta_fast"] = 32.0
        config.rope_scaling["beta_slow"] = 1.0
    config.longvideo_kwargs = exp_configs.get("longvideo_kwargs", {})
    return config


def patch_llava_onevision_config(config, exp_configs):
    """Same for the text config of LLaVA-OneVision (reference: monkeypatch.py:37-48)."""
    if "scaling_factor" in exp_configs:
        config.text_config.rope_scaling = {"rope_type": "yarn", "factor": exp_configs["scaling_factor"],
                                           "beta_fast": 32.0, "beta_slow": 1.0}
    config.longvideo_kwargs = exp_configs.get("longvideo_kwargs", {})
    return config


def _rebind(module_name: str, class_name: str, **attrs) -> bool:
    module = importlib.import_module(module_name)
    cls = getattr(module, class_name, None)
    if cls is None:
        return False
    for name, fn in attrs.items():
        setattr(cls, name, fn)
    return True


def patch_qwen2vl(method):
    if method != "retake":
        raise NotImplementedError
    print("Using ReTaKe for Qwen2VLForConditionalGeneration!")
    m = "transformers.models.qwen2_vl.modeling_qwen2_vl"
    _rebind(m, "Qwen2VLAttention", forward=retake_Qwen2VLAttention_forward)
    _rebind(m, "Qwen2VLSdpaAttention", forward=retake_Qwen2VLSdpaAttention_forward)
    _rebind(m, "Qwen2VLFlashAttention2", forward=retake_Qwen2VLFlashAttention2_forward)
    _rebind(m, "Qwen2VLForConditionalGeneration",
            compress_video_tokens=retake_Qwen2VLForConditionalGeneration_compress_video_tokens,
            segment_input_ids=retake_Qwen2VLForConditionalGeneration_segment_input_ids,
            get_chunk_size=retake_Qwen2VLForConditionalGeneration_get_chunk_size,
            forge_input_chunks=retake_Qwen2VLForConditionalGeneration_forge_input_chunks,
            forward=retake_Qwen2VLForConditionalGeneration_forward)


def patch_llava_onevision(method):
    if method != "retake":
        raise NotImplementedError
    print("Using ReTaKe for LlavaOnevisionForConditionalGeneration!")
    _rebind("transformers.models.qwen2.modeling_qwen2", "Qwen2Attention", __init__=retake_Qwen2Attention_init,
            forward=retake_Qwen2Attention_forward)
    _rebind("transformers.models.llava_onevision.modeling_llava_onevision", "LlavaOnevisionForConditionalGeneration",
            get_chunk_size=retake_LlavaOnevisionForConditionalGeneration_get_chunk_size,
            segment_input_ids=retake_LlavaOnevisionForConditionalGeneration_segment_input_ids,
            compress_video_tokens=retake_LlavaOnevisionForConditionalGeneration_compress_video_tokens,
            forge_input_chunks=retake_LlavaOnevisionForConditionalGeneration_forge_input_chunks,
            forward=retake_LlavaOnevisionForConditionalGeneration_forward)
