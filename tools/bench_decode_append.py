#!/usr/bin/env python3
"""Decode steps after a compressed prefill: what one generated token costs in cache maintenance.

The reference's cache appends with `torch.cat` (longvideo_cache.py:238 through DynamicCache.update): every decode step
copies the layer's whole K and V - O(cache) bytes per layer and token.  This package appends into the layer's
pre-allocated buffer.  Measured here on the cache a 2048-frame prefill leaves behind (100 352 tokens per
layer at ratio 0.25): 28 layers x `steps` decode updates of one token each, both ways, same tensors.

    python tools/bench_decode_append.py [--frames 2048 --steps 64]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "video-retake_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, _p)
import torch

import bench as B


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=2048)
    ap.add_argument("--steps", type=int, default=64)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    td = torch.bfloat16
    T, layers = args.frames, B.LAYERS
    n_chunks = T // B.FRAMES_PER_CHUNK
    frames = torch.cat([B.chunk_frames(c, dev, td) for c in range(n_chunks)])[None]
    pool = [B.pool_set(i, dev, td) for i in range(48)]
    pos_base = [B.chunk_position_ids(c, dev) for c in range(n_chunks)]
    rotary = B.Rotary(dev)
    _, cache, _ = B.run_video(frames, pool, None, pos_base, rotary, layers, td)
    cache.kvcache_compression = False          # generation: plain appends (qwen2_vl.py:715-716 resets it after the video)
    P0 = cache.key_cache[0].shape[2]
    g = torch.Generator(device=dev).manual_seed(1)
    toks = [tuple((1.7 * torch.randn((1, h, 1, B.D), generator=g, device=dev)).to(td) for h in (B.Hq, B.Hkv, B.Hkv))
            for _ in range(8)]
    last = int(cache.position_cache[0].reshape(-1, cache.position_cache[0].shape[-1])[0, -1].item())

    def ours(steps):
        for s in range(steps):
            pos = torch.full((3, 1, 1), last + 1 + s, dtype=torch.int64, device=dev)
            for l in range(layers):
                q, k, v = toks[(s * layers + l) % len(toks)]
                cache.update(k, v, l, {"query_states": q, "position_ids": pos, "rotary_emb": rotary, "mrope_section": B.MROPE})

    # the reference's way on copies of the same caches
    ref_k = [cache.key_cache[l].clone() for l in range(layers)]
    ref_v = [cache.value_cache[l].clone() for l in range(layers)]

    def cat(steps):
        for s in range(steps):
            for l in range(layers):
                _, k, v = toks[(s * layers + l) % len(toks)]
                ref_k[l] = torch.cat([ref_k[l], k], dim=-2)
                ref_v[l] = torch.cat([ref_v[l], v], dim=-2)

    res = {}
    for name, fn in (("preallocated_append", ours), ("torch_cat", cat)):
        fn(2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn(args.steps)
        torch.cuda.synchronize()
        res[name] = (time.perf_counter() - t0) / args.steps * 1e3
    assert cache.key_cache[0].shape[2] == P0 + 2 + args.steps == ref_k[0].shape[2]
    assert torch.equal(cache.key_cache[3], ref_k[3]) and torch.equal(cache.value_cache[27], ref_v[27])
    print(json.dumps({"cache_tokens_per_layer": P0, "layers": layers, "decode_steps": args.steps,
                      "ms_per_token_preallocated_append": res["preallocated_append"], "ms_per_token_torch_cat": res["torch_cat"],
                      "cat_bytes_per_token": 2 * 2 * layers * B.Hkv * P0 * B.D * 2,
                      "checked": "both caches hold the same K / V afterwards"}))


if __name__ == "__main__":
    main()
