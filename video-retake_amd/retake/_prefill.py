"""Model-independent pieces of ReTaKe's chunked prefill (shared by the Qwen2-VL and LLaVA-Video glue).

Reference: retake/qwen2_vl.py:444-475 (segment_input_ids), :548-557 (dynamic ratio), :670-720 (chunked
prefill loop); retake/llava_onevision.py:164-198, :337-347, :499-545 are the same logic for LLaVA.
"""
from __future__ import annotations

import math
from typing import Callable, List, Optional, Tuple

import torch

Segment = Tuple[int, int, str]


def segment_token_runs(is_video: torch.Tensor) -> List[Segment]:
    """Run-length segmentation of a 1-D bool tensor into sorted (start, end, 'video' | 'text') runs,
    end exclusive (reference: qwen2_vl.py:444-475)."""
    flags = is_video.to(torch.bool).cpu().tolist()
    segments: List[Segment] = []
    start = 0
    for i in range(1, len(flags) + 1):
        if i == len(flags) or flags[i] != flags[start]:
            segments.append((start, i, "video" if flags[start] else "text"))
            start = i
    return segments


def apply_dynamic_compression_ratio(config, input_length: int) -> None:
    """`dynamic_compression_ratio`: ratio = max_input_length / input_length when the prompt is longer than
    the budget, else 1.  Written back into the shared config dict (reference: qwen2_vl.py:548-557)."""
    kwargs = getattr(config, "longvideo_kwargs", None)
    if not kwargs or not kwargs.get("kvcache_compression", False):
        return
    comp = kwargs["kvcache_compression_kwargs"]
    if comp.get("dynamic_compression_ratio", False):
        max_len = comp["max_input_length"]
        comp["compression_ratio"] = 1 if input_length <= max_len else max_len / input_length


def expected_cache_tokens(config, input_length: int, chunk_size: Optional[int]) -> Optional[int]:
    """Capacity hint for the pre-allocated PivotKV cache (not in the reference): the prompt compressed at the configured
    ratio, plus one uncompressed chunk in flight, plus room for generation.  An over-long answer only costs a regrowth."""
    kwargs = getattr(config, "longvideo_kwargs", None)
    if not kwargs or not kwargs.get("kvcache_compression", False) or chunk_size is None:
        return None
    ratio = float(kwargs["kvcache_compression_kwargs"].get("compression_ratio", 1.0))
    return int(math.ceil(min(1.0, ratio) * input_length)) + int(chunk_size) + 2048


def prompt_guided(config) -> bool:
    kwargs = getattr(config, "longvideo_kwargs", None)
    if not kwargs or not kwargs.get("kvcache_compression", False):
        return False
    comp = kwargs["kvcache_compression_kwargs"]
    return bool(comp.get("prompt_guided_compression", False) and comp.get("compression_ratio", 1) < 1.0)


def run_chunked_prefill(segments: List[Segment], chunk_size: int, cache, keypatches_mask,
                        run_text: Callable[[int, int], object], run_video_chunk: Callable[[int, int], object]):
    """The prefill driver (reference: qwen2_vl.py:670-720).

    Text segments are prefilled in one call with compression off; each video segment is cut into
    `chunk_size`-token chunks, the cache gets the chunk's key-patch mask (the callback fires the
    `before_forward` / `after_forward` hooks around the model call), and compression is switched off
    again afterwards so that decoding appends normally.  Returns the last model output.
    """
    compression_on = getattr(cache, "kvcache_compression", False)
    outputs = None
    for (s, e, kind) in segments:
        if kind == "text":
            cache.kvcache_compression = False
            outputs = run_text(s, e)
        elif kind == "video":
            cache.kvcache_compression = compression_on
            for idx in range(math.ceil((e - s) / chunk_size)):
                ss = s + idx * chunk_size
                ee = min(s + (idx + 1) * chunk_size, e)
                if keypatches_mask is not None:
                    cache.keypatches_mask_chunk = keypatches_mask[0, ss:ee]
                outputs = run_video_chunk(ss, ee)
            cache.keypatches_mask_chunk = None
            cache.kvcache_compression = False  # turned off for decoding
        else:
            raise ValueError(kind)
        try:  # the model hands the cache back through its output (reference: `past_key_values = outputs[...]`)
            cache = outputs["past_key_values"]
        except (KeyError, TypeError, IndexError):
            pass
    return outputs
