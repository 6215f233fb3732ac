#!/usr/bin/env python3
"""The fused prepare kernel on the two q / k / v layouts it meets:
  head-major   [1, H, L, D] contiguous - what bench.py's resident pool holds;
  HF layout    the [1, L, H, D] projection output viewed as [1, H, L, D] (`.view(b, L, H, D).transpose(1, 2)`, then the
               element-wise RoPE keeps that permutation) - what the patched attention forwards hand to `update`.
Same values, same cache calls (28 layers x a few chunks, native RoPE, bf16); the kernel's HIP-event time per launch.

    python tools/bench_prepare_layout.py
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "video-retake_amd"))

import torch  # noqa: E402

import bench  # noqa: E402


def run(geometry, hf_layout, chunks, dev):
    import retake._native as nv
    import retake.longvideo_cache as lc

    bench.set_geometry(geometry)
    L = bench.FRAMES_PER_CHUNK * bench.N_PATCH
    layers = bench.LAYERS
    pool = [bench.pool_set(i, dev, torch.bfloat16) for i in range(layers)]
    if hf_layout:
        pool = [tuple(t.transpose(1, 2).contiguous().transpose(1, 2) for t in s) for s in pool]
        assert not pool[0][0].is_contiguous()
    rotary = bench.Rotary(dev)
    ids = nv.profile_kernel_ids()
    out = None
    for rep in range(2):   # first pass warms the allocator, second is read
        cache = lc.build_kvcache(bench.make_cache_config(layers), reserve_tokens=chunks * max(1, int(bench.RATIO * L)) + L)
        nv.check(nv.lib.rtk_profile_reset(), "reset")
        nv.check(nv.lib.rtk_profile_enable_mask(1 << ids["unrotate_pack"]), "enable")
        for c in range(chunks):
            cache.keypatches_mask_chunk = None
            cache.kvcache_compression = True
            pos = bench.chunk_position_ids(c, dev)
            for l in range(layers):
                q, k, v = pool[l]
                cache.shift_temporal_ids_(pos, l)
                cache.update(k, v, l, {"query_states": q, "position_ids": pos, "rotary_emb": rotary,
                                       "mrope_section": bench.MROPE})
            cache.after_forward()
        torch.cuda.synchronize()
        nv.check(nv.lib.rtk_profile_enable(0), "disable")
        n, ms = nv.profile_read()["unrotate_pack"]
        out = {"launches": n, "avg_us": ms / n * 1e3}
        del cache
    return out


def main():
    dev = torch.device("cuda:0")
    res = {}
    for geometry, chunks in (("qwen448", 6), ("baseline", 3)):
        res[geometry] = {"head_major": run(geometry, False, chunks, dev), "hf_layout": run(geometry, True, chunks, dev)}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
