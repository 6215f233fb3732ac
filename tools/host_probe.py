#!/usr/bin/env python3
"""Host-side cost of the PivotKV update path: enqueue time vs wall time, and a cProfile of the enqueue loop.
    python tools/host_probe.py [--frames 256] [--profile]"""
import argparse
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "video-retake_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch

import bench as B


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=256)
    ap.add_argument("--profile", action="store_true")
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--streams", type=int, default=0)
    a = ap.parse_args()
    B.OVERLAP_STREAMS = a.streams
    dev = torch.device("cuda:0")
    td = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    g = torch.Generator(device=dev).manual_seed(0)
    L = B.FRAMES_PER_CHUNK * B.N_PATCH
    frames = torch.randn((1, a.frames, B.N_PATCH, B.C_EMB), generator=g, device=dev).to(td)
    pool = [tuple((1.7 * torch.randn((1, h, L, B.D), generator=g, device=dev)).to(td) for h in (B.Hq, B.Hkv, B.Hkv))
            for _ in range(24)]
    n_chunks = a.frames // B.FRAMES_PER_CHUNK
    pos_base = [B.chunk_position_ids(c, dev) for c in range(n_chunks)]
    rot = B.Rotary(dev)
    B.run_video(frames, pool, None, pos_base, rot, B.LAYERS, td)
    torch.cuda.synchronize()
    for rep in range(2):
        t0 = time.perf_counter()
        B.run_video(frames, pool, None, pos_base, rot, B.LAYERS, td)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        n = n_chunks * B.LAYERS
        print(f"rep {rep}: enqueue {1e6 * (t1 - t0) / n:.1f} us/update, wall {1e6 * (t2 - t0) / n:.1f} us/update "
              f"({n} updates)")
    if a.profile:
        pr = cProfile.Profile()
        pr.enable()
        B.run_video(frames, pool, None, pos_base, rot, B.LAYERS, td)
        pr.disable()
        torch.cuda.synchronize()
        pstats.Stats(pr).sort_stats("cumulative").print_stats(25)


if __name__ == "__main__":
    main()
