#!/usr/bin/env python3
"""BASELINE configs[4] on ONE MI355X: the DPSelect + PivotKV share of a 2048-frame LLaVA-Video-Qwen2-7B prefill.

    python tools/bench_llava.py [--frames 2048 --steps 2 --warmup 1]

Not the driver's bench (bench.py is; it times configs[2]).  Same plugin surface, LLaVA-Video geometry:
  DPSelect (ratio 1.0, patch_sync False) on the SigLIP patch embeddings [1, T, 729, 1152] bf16 (3.4 GB at T = 2048);
  the key-patch mask [T * 729] truncated to the first T * 196 + 1 entries (the reference's quirk, llava_onevision.py:486);
  PivotKV on T / 32 chunks x 28 layers of L = 6272 pooled tokens, plain RoPE with [1, L] ids (no M-RoPE), YaRN factor 4,
  pos_embed_reforge, `dynamic_compression_ratio` with max_input_length 40000: ratio 40000 / (T * 196 + 1), keep = 624 at 2048
  frames (retake_llava-video_*.yaml).
Prints one JSON line: frames/s, the score kernels' average launch durations (HIP events on the launch stream) and the
self-check of the last chunk (batched launches == one-unit launches, bitwise)."""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "video-retake_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, _p)
import torch

import bench as B

N_SIGLIP, C_SIGLIP, N_POOLED = 729, 1152, 196


class PlainRotary:
    """inv_freq * position with YaRN attention_scaling for [1, L] ids (what Qwen2's rotary module computes)."""

    def __init__(self, device):
        self.inv_freq = (1.0 / (1e6 ** (torch.arange(0, B.D, 2, dtype=torch.int64).float() / B.D))).to(device)
        self.attention_scaling = B.A_SCALE

    def __call__(self, x, position_ids):
        freqs = position_ids[:, :, None].float() * self.inv_freq[None, None, :]
        emb = torch.cat((freqs, freqs), dim=-1)
        return (emb.cos() * self.attention_scaling).to(x.dtype), (emb.sin() * self.attention_scaling).to(x.dtype)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=2048)
    ap.add_argument("--layers", type=int, default=B.LAYERS)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--pool", type=int, default=48)
    args = ap.parse_args()
    import retake._native as nv
    import retake.longvideo_cache as lc
    import retake.visual_compression as vc
    import unit_check as uc
    from retake import _prefill

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    T, layers = args.frames, args.layers
    L = B.FRAMES_PER_CHUNK * N_POOLED
    n_chunks = T // B.FRAMES_PER_CHUNK
    g = torch.Generator(device=dev).manual_seed(4242)
    frames = torch.randn((1, T, N_SIGLIP, C_SIGLIP), generator=g, device=dev, dtype=torch.float32).bfloat16()
    pool = [B.pool_set(i, dev, torch.bfloat16) for i in range(min(args.pool, n_chunks * layers))]
    rotary = PlainRotary(dev)
    llm = types.SimpleNamespace(hidden_size=B.Hq * B.D, num_hidden_layers=layers, num_attention_heads=B.Hq,
                                num_key_value_heads=B.Hkv)
    kwc = {"compression_ratio": 0.5, "compression_method": "pivotkv", "pos_embed_reforge": True, "native_rope": True,
           "dynamic_compression_ratio": True, "max_input_length": 40000}
    n_visual = T * N_POOLED + 1

    def run():
        cfg = types.SimpleNamespace(text_config=llm, longvideo_kwargs={"kvcache_compression": True,
                                                                       "kvcache_compression_kwargs": dict(kwc)})
        _prefill.apply_dynamic_compression_ratio(cfg, n_visual)          # what the model forward does (llava_onevision.py)
        out, mask = vc.memory_bank_compress_keyframe(frames, T, 3, sync=False)
        mask = mask[:n_visual]                                            # the reference's truncation, not a re-pooling
        cache = lc.build_kvcache(cfg)
        keep = max(1, int(cache.compression_ratio * L))
        call = 0
        for c in range(n_chunks):
            cache.keypatches_mask_chunk = mask[c * L:(c + 1) * L]
            cache.kvcache_compression = True
            pos = (torch.arange(L, device=dev) + 16 + c * L)[None].contiguous()
            for layer in range(layers):
                q, k, v = pool[call % len(pool)]
                call += 1
                p_l = cache.shift_temporal_ids_(pos.clone(), layer)      # LLaVA's patch shifts a clone per layer
                cache.update(k, v, layer, {"query_states": q, "position_ids": p_l, "rotary_emb": rotary})
            cache.after_forward()
        return cache, mask, keep

    for _ in range(args.warmup):
        run()
    ids = nv.profile_kernel_ids()
    nv.check(nv.lib.rtk_profile_reset(), "profile_reset")
    nv.check(nv.lib.rtk_profile_enable_mask((1 << ids["score_pass1"]) | (1 << ids["score_pass2"]) |
                                            (1 << ids["dpselect_dis"]) | (1 << ids["gather_frames"])), "profile_enable")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        cache, mask, keep = run()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    nv.check(nv.lib.rtk_profile_enable(0), "profile_enable")
    kern = {k: {"launches": n, "avg_us": ms / n * 1e3} for k, (n, ms) in nv.profile_read().items()}
    # untimed: the last chunk's first / middle / last layer against one-unit launches (bitwise)
    c = n_chunks - 1
    lay = sorted({0, layers // 2, layers - 1})
    inputs = {l: pool[(c * layers + l) % len(pool)][:2] for l in lay}
    uc.check_batch_against_units(cache, lay, inputs, {l: mask[c * L:(c + 1) * L] for l in lay}, keep, rotary.inv_freq,
                                 B.A_SCALE, None)
    flops = 2.0 * B.Hq * L * L * B.D * layers
    out = {"metric": "frames/sec through DPSelect+PivotKV, LLaVA-Video geometry (BASELINE configs[4], 1 GPU share)",
           "value": T * args.steps / dt, "unit": "frames/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": dt / args.steps * 1e3, "dtype": "bf16", "data": "synthetic",
           "config": {"workload": f"DPSelect on [1,{T},{N_SIGLIP},{C_SIGLIP}] bf16 + PivotKV (plain RoPE, dynamic ratio "
                                  f"{cache.compression_ratio:.5f}, keep {keep}) on {n_chunks} chunks x {layers} layers, L={L}",
                      "keep": keep, "assembled_cache_tokens": int(cache.key_cache[0].shape[2])},
           "kernels_timed_region": kern,
           # pass 1 (every key) priced against the dense bf16 peak; pass 2 computes the unmasked columns only
           "score_pass1_frac_of_2.5PF": flops / (kern["score_pass1"]["avg_us"] * 1e-6) / 2.5e15,
           "key_patch_mask_rate": float(mask[: n_chunks * L].float().mean().item()),
           "dpselect_dis_GBps": (T * N_SIGLIP * C_SIGLIP * 2 + 4 * T * N_SIGLIP) / (kern["dpselect_dis"]["avg_us"] * 1e-6) / 1e9,
           "self_check": "batched score / keep_idx / new ids == one-unit launches (bitwise), last chunk, layers %s" % lay}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
