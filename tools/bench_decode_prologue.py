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
    ap = argparse.ArgumentParser()
    ap.add_argument("--tokens", type=int, default=200)
    args = ap.parse_args()
    print(json.dumps(bench.decode_prologue_measurement(torch.device("cuda:0"), args.tokens)))


if __name__ == "__main__":
    main()
