#!/usr/bin/env python3
"""Small driver for rocprofv3: runs each hot-path kernel a few times at BASELINE geometry.

    rocprofv3 --kernel-trace --stats -d gpurun_out/prof -- python3 tools/prof_driver.py --what score --iters 20
"""
import argparse
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "video-retake_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch

import bench


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--what", default="all", choices=["all", "score", "dpselect", "update"])
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--frames", type=int, default=2048)
    a = ap.parse_args()
    import retake.longvideo_cache as lc
    import retake.visual_compression as vc

    dev = torch.device("cuda:0")
    td = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    g = torch.Generator(device=dev).manual_seed(0)
    L = bench.FRAMES_PER_CHUNK * bench.N_PATCH
    if a.what in ("all", "dpselect"):
        x = torch.randn((1, a.frames, bench.N_PATCH, bench.C_EMB), generator=g, device=dev).to(td)
        for _ in range(max(1, a.iters // 4)):
            vc.memory_bank_compress_keyframe(x, a.frames, 3, sync=False)
        del x
    if a.what in ("all", "score", "update"):
        sets = []
        for i in range(8):
            sets.append(tuple((1.7 * torch.randn((1, h, L, bench.D), generator=g, device=dev)).to(td)
                              for h in (bench.Hq, bench.Hkv, bench.Hkv)))
        rot = bench.Rotary(dev)
        cache = lc.build_kvcache(bench.make_cache_config(1))
        pos = bench.chunk_position_ids(0, dev)
        mask = torch.rand(L, generator=g, device=dev) < 0.33
        for i in range(a.iters):
            q, k, v = sets[i % len(sets)]
            cache.keypatches_mask_chunk = mask
            cache.update(k, v, 0, {"query_states": q, "position_ids": pos, "rotary_emb": rot,
                                   "mrope_section": bench.MROPE})
        cache.after_forward()
    torch.cuda.synchronize()
    print("done")


if __name__ == "__main__":
    main()
