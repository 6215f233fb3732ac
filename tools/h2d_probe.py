#!/usr/bin/env python3
"""Host-to-device rate for the DPSelect input of BASELINE configs[2] ([1, 2048, 196, 1280] bf16 = 1.03 GB) from pinned
host memory: what a host-resident caller would pay in front of the path (never part of bench.py's `value`)."""
import time

import torch

dev = torch.device("cuda:0")
h = torch.empty((1, 2048, 196, 1280), dtype=torch.bfloat16).pin_memory()
d = torch.empty_like(h, device=dev)
d.copy_(h, non_blocking=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    d.copy_(h, non_blocking=True)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
print("H2D %.3f GB in %.2f ms = %.1f GB/s" % (h.numel() * 2 / 1e9, dt * 1e3, h.numel() * 2 / dt / 1e9))
