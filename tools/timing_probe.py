#!/usr/bin/env python3
"""Reads the s_memtime phase counters of an RTK_TIMING build (tools/variants.sh timing "-DRTK_TIMING")."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "video-retake_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch

import retake._native as nv

dev = torch.device("cuda:0")
Hq, Hkv, D, L = 28, 4, 128, 6272
g = torch.Generator(device=dev).manual_seed(0)
q = (1.7 * torch.randn((1, Hq, L, D), generator=g, device=dev)).bfloat16()
k = (1.7 * torch.randn((1, Hkv, L, D), generator=g, device=dev)).bfloat16()
dt = 1
wsb = nv.lib.rtk_pivotkv_score_workspace_bytes(Hq, Hkv, L, D, dt)
ws = torch.empty(wsb + 256, dtype=torch.uint8, device=dev)
wsp = (ws.data_ptr() + 255) & ~255
score = torch.empty(L, dtype=torch.float32, device=dev)
lib = C.CDLL(nv.LIB_PATH)
out = (C.c_ulonglong * 8)()
for it in range(3):
    nv.check(nv.lib.rtk_pivotkv_score(nv.ptr(q), q.stride(1), q.stride(2), nv.ptr(k), k.stride(1), k.stride(2), Hq, Hkv, L, D,
                                      dt, None, None, 1.0, nv.ptr(score), None, C.c_void_p(wsp), wsb, nv.stream()), "score")
    torch.cuda.synchronize()
    lib.rtk_debug_read_timing(out, 1)
    v = list(out)
    tiles = max(v[7], 1)
    names = ["issue loads", "LDS reads + MFMAs", "softmax VALU", "LDS store", "barrier"]
    print("iter", it, "tiles sampled", tiles)
    tot = sum(v[:5])
    if tot == 0:
        print('   (score kernels of this build carry no phase counters)')
        break
    for n, x in zip(names, v[:5]):
        print(f"   {n:20s} {x / tiles:9.1f} cycles/tile  ({100 * x / tot:5.1f} %)")
    print(f"   total per tile {tot / tiles:9.1f}")

# ---- select kernel phases (single workgroup) ----
if hasattr(lib, "rtk_debug_read_select_timing"):
    keep = L // 4
    mask = torch.rand(L, device=dev) < 0.3
    pos = torch.arange(L, device=dev)[None].repeat(3, 1).contiguous()
    keep_idx = torch.empty(keep, dtype=torch.int64, device=dev)
    rank = torch.empty(L, dtype=torch.int32, device=dev)
    pos_out = torch.empty((3, keep), dtype=torch.int64, device=dev)
    for it in range(3):
        nv.check(nv.lib.rtk_pivotkv_select(nv.ptr(score), nv.ptr(mask), L, keep, nv.ptr(pos), 3, 1, nv.ptr(keep_idx), nv.ptr(rank),
                                           nv.ptr(pos_out), keep, None, 0, nv.stream()), "select")
        torch.cuda.synchronize()
        lib.rtk_debug_read_select_timing(out)
        v = list(out)
        print("select phases (cycles): load", v[1] - v[0], "radix", v[2] - v[1], "scan+min", v[3] - v[2], "emit", v[4] - v[3],
              "total", v[4] - v[0])
