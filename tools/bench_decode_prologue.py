#!/usr/bin/env python3
"""Decode-time cost of the attention patch's prologue (qwen2_vl.py:68-86 + the cache's else-branch :319-321) for the 28
layers of one generated token, after a compressed prefill: the op-by-op route (continuity shift, rotary module,
apply_multimodal_rotary_pos_emb, PivotKVCache.update - what the reference's patch runs, ~25 launches per layer) against
the fused one (PivotKVCache.append_pre_rope: one kernel + the in-place id shift).  Wall time per token with the GPU
drained at both ends, and the text-prefill form (a 64-token segment).  One JSON line.

    python tools/bench_decode_prologue.py [--tokens 200]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "video-retake_amd"))

import torch  # noqa: E402

import bench  # noqa: E402


def main():
    import retake.longvideo_cache as lc

    ap = argparse.ArgumentParser()
    ap.add_argument("--tokens", type=int, default=200)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    td = torch.bfloat16
    layers, Hq, Hkv, D = bench.LAYERS, bench.Hq, bench.Hkv, bench.D
    rot = bench.Rotary(dev)
    out = {"layers": layers, "dtype": "bf16", "prefix_tokens_per_layer": 40000}

    def proj(n):
        return tuple(torch.randn((1, n, h, D), device=dev, dtype=torch.float32).to(td).transpose(1, 2) for h in (Hq, Hkv, Hkv))

    for n, label in ((1, "decode_token"), (64, "text_segment_64")):
        res = {}
        for fused in (True, False):
            cache = lc.build_kvcache(bench.make_cache_config(layers), reserve_tokens=40000 + args.tokens * n + 64)
            cache.kvcache_compression = False
            # a prefix as a long compressed prompt leaves it: 40 000 cached rows per layer with their ids
            pre = 40000
            for l in range(layers):
                st = cache.reserve(l, pre, torch.empty((1, Hkv, 1, D), dtype=td, device=dev))
                st.length = pre
                cache._pos_reserve(st, 3, 3, pre, dev)
                st.pos[:, :pre] = torch.arange(pre, device=dev)
                st.pos_len = pre
            cache._pos_layers = layers
            qkv = [proj(n) for _ in range(4)]

            def token(t):
                pos = (torch.arange(n, device=dev) + 50000 + t * n).view(1, 1, n).repeat(3, 1, 1)
                for l in range(layers):
                    q, k, v = qkv[(t + l) % 4]
                    if fused:
                        r = cache.append_pre_rope(q, k, v, l, pos, rot, bench.MROPE)
                        assert r is not None
                    else:
                        cache.shift_temporal_ids_(pos, l)
                        cos, sin = rot(v, pos)
                        qr, kr = lc.apply_multimodal_rotary_pos_emb(q, k, cos, sin, bench.MROPE)
                        cache.update(kr, v, l, {"sin": sin, "cos": cos, "query_states": qr, "position_ids": pos,
                                                "rotary_emb": rot, "mrope_section": bench.MROPE})

            for t in range(5):
                token(t)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for t in range(args.tokens):
                token(5 + t)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            res["fused" if fused else "op_by_op"] = {"enqueue_us_per_step": (t1 - t0) / args.tokens * 1e6,
                                                       "wall_us_per_step": (t2 - t0) / args.tokens * 1e6}
            del cache
            torch.cuda.empty_cache()
        res["speedup_wall"] = res["op_by_op"]["wall_us_per_step"] / res["fused"]["wall_us_per_step"]
        out[label] = res
    print(json.dumps(out))


if __name__ == "__main__":
    main()
