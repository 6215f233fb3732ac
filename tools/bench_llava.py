#!/usr/bin/env python3
"""BASELINE configs[4] on ONE MI355X: the DPSelect + PivotKV share of a 2048-frame LLaVA-Video-Qwen2-7B prefill.

    python tools/bench_llava.py [--frames 2048 --steps 2 --warmup 1]

The measurement itself lives in bench.py (`llava_measurement`; the driver's line carries it as the `llava_workload`
companion); this is its stand-alone entry point.  Prints one JSON line."""
from __future__ import annotations

import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "video-retake_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, _p)
import torch

import bench as B


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=2048)
    ap.add_argument("--layers", type=int, default=B.LAYERS)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--pool", type=int, default=48)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    print(json.dumps(B.llava_measurement(dev, args.frames, args.layers, args.steps, args.warmup, args.pool)), flush=True)


if __name__ == "__main__":
    main()
