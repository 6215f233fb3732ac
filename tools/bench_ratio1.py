#!/usr/bin/env python3
"""compression_ratio 1: the ratio `dynamic_compression_ratio` gives every prompt that fits max_input_length
(reference qwen2_vl.py:548-557; every shipped config sets it, max_input_length 32000 for Qwen2-VL).

Times bench.py's step on a SHORT video of the real Qwen2-VL geometry (13 chunks x 2304 tokens = 29 952 video tokens,
the longest prompt that still runs at ratio 1) and of the BASELINE geometry (5 chunks x 6272), with the default
(no scoring: topk(k = L) + sort is the identity) and with `score_when_keeping_all` (the reference's route: score every
chunk, keep everything).  One JSON line.  Not a driver line.

    python tools/bench_ratio1.py [--steps 3]
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


def measure(geometry, chunks, scored, steps, dev):
    bench.set_geometry(geometry)
    bench.RATIO = 1
    base = bench.make_cache_config

    def cfg(layers):
        c = base(layers)
        c.longvideo_kwargs["kvcache_compression_kwargs"]["score_when_keeping_all"] = scored
        return c

    bench.make_cache_config = cfg
    try:
        tdtype = torch.bfloat16
        rows = chunks * bench.FRAMES_PER_CHUNK
        L = bench.FRAMES_PER_CHUNK * bench.N_PATCH
        frames = torch.cat([bench.chunk_frames(c, dev, tdtype) for c in range(chunks)])[None]
        pool = [bench.pool_set(i, dev, tdtype) for i in range(min(48, chunks * bench.LAYERS))]
        pos_base = [bench.chunk_position_ids(c, dev) for c in range(chunks)]
        rotary = bench.Rotary(dev)
        bench.run_video(frames, pool, None, pos_base, rotary, bench.LAYERS, tdtype)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            _, cache, _ = bench.run_video(frames, pool, None, pos_base, rotary, bench.LAYERS, tdtype)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        assert cache.key_cache[0].shape[2] == chunks * L
        fingerprint = bench.cache_checksum([cache.key_cache[l] for l in range(bench.LAYERS)],
                                           [cache.value_cache[l] for l in range(bench.LAYERS)],
                                           [cache.position_cache[l] for l in range(bench.LAYERS)])
        del cache
        torch.cuda.empty_cache()
        return {"ms_per_video": dt * 1e3, "frames_per_s": rows * bench.FRAMES_PER_ROW / dt, "chunks": chunks,
                "chunk_tokens": L, "video_tokens": chunks * L}, fingerprint
    finally:
        bench.make_cache_config = base


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    out = {}
    for geometry, chunks in (("qwen448", 13), ("baseline", 5)):
        skipped, fa = measure(geometry, chunks, False, args.steps, dev)
        scored, fb = measure(geometry, chunks, True, args.steps, dev)
        assert fa == fb, (fa, fb)   # same cache either way: ids, V bits and |K| sums
        out[geometry] = {"keep_all_unscored": skipped, "keep_all_scored": scored,
                         "speedup": scored["ms_per_video"] / skipped["ms_per_video"], "same_cache": True}
    print(json.dumps({"compression_ratio": 1, "dtype": "bf16", "layers": bench.LAYERS, **out}))


if __name__ == "__main__":
    main()
