"""retake — MI355X-native DPSelect + PivotKV behind the reference's plugin surface.

Module paths, function names, argument order and return values follow SCZwangxiao/video-ReTaKe:
    retake.visual_compression.memory_bank_compress_keyframe
    retake.longvideo_cache.{PivotKVCache, build_kvcache, apply_*rotary_pos_emb, ...}
    retake.monkeypatch.{patch_qwen2vl, patch_qwen2vl_config, patch_llava_onevision, ...}
The arithmetic runs in hand-written gfx950 HIP kernels (libretake_hip.so, C ABI in
include/retake_hip.h); torch only provides device memory and streams.
"""
__version__ = "0.1.0"
