#!/usr/bin/env python3
"""Micro-benchmark of rtk_pivotkv_score (and friends) at BASELINE geometry; prints per-kernel avg µs.
    python tools/bench_score.py [--dtype bf16|fp32] [--L 6272] [--iters 20]"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "video-retake_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch

import retake._native as nv


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--L", type=int, default=6272)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--refround", action="store_true", help="bf16 payloads with the reference's bf16 score chain")
    ap.add_argument("--units", type=int, default=0, help="time rtk_pivotkv_score_passes_batched over this many units")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    td = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    Hq, Hkv, D, L = 28, 4, 128, a.L
    g = torch.Generator(device=dev).manual_seed(0)
    sets = [tuple((1.7 * torch.randn((1, h, L, D), generator=g, device=dev)).to(td) for h in (Hq, Hkv)) for _ in range(6)]
    cos = torch.rand((L, D), generator=g, device=dev)
    sin = (1 - cos * cos).sqrt()
    dt = nv.dtype_code(sets[0][0])
    if a.refround:
        dt = nv.RTK_BF16_REFROUND
    wsb = nv.lib.rtk_pivotkv_score_workspace_bytes(Hq, Hkv, L, D, dt)
    ws = torch.empty(wsb + 256, dtype=torch.uint8, device=dev)
    wsp = (ws.data_ptr() + 255) & ~255
    score = torch.empty(L, dtype=torch.float32, device=dev)
    kun = torch.empty((Hkv, L, D), dtype=td, device=dev)
    st = nv.stream()

    def run(q, k):
        nv.check(nv.lib.rtk_pivotkv_score(nv.ptr(q), q.stride(1), q.stride(2), nv.ptr(k), k.stride(1), k.stride(2), Hq, Hkv,
                                          L, D, dt, nv.ptr(cos), nv.ptr(sin), 1.1386, nv.ptr(score), nv.ptr(kun),
                                          C.c_void_p(wsp), wsb, st), "score")
    if a.units:
        # the chunk-batched launches bench.py times: one launch per kernel over `units` prepared workspaces
        stride = (wsb + 255) & ~255
        big = torch.empty(a.units * stride + 256, dtype=torch.uint8, device=dev)
        base = (big.data_ptr() + 255) & ~255
        kuns = torch.empty((a.units, Hkv, L, D), dtype=td, device=dev)
        rs_n = C.c_int(0)
        pf = nv.lib.rtk_pivotkv_score_partials(Hq, Hkv, L, D, dt, C.byref(rs_n))
        parts = torch.empty((a.units, pf), dtype=torch.float32, device=dev)
        for u in range(a.units):
            q, k = sets[u % len(sets)]
            nv.check(nv.lib.rtk_pivotkv_score_stages(nv.ptr(q), q.stride(1), q.stride(2), nv.ptr(k), k.stride(1), k.stride(2),
                                                     Hq, Hkv, L, D, dt, nv.ptr(cos), nv.ptr(sin), 1.1386, nv.ptr(score),
                                                     nv.ptr(kuns[u]), C.c_void_p(base + u * stride), wsb, nv.SCORE_PREPARE,
                                                     None, st), "prepare")

        def runb():
            nv.check(nv.lib.rtk_pivotkv_score_passes_batched(C.c_void_p(base), stride, nv.ptr(kuns), Hkv * L * D * kuns.element_size(),
                                                             nv.ptr(parts), pf, a.units, Hq, Hkv, L, D, dt, None, None, st), "batched")
        for _ in range(2):
            runb()
        torch.cuda.synchronize()
        nv.lib.rtk_profile_reset()
        nv.lib.rtk_profile_enable(1)
        for _ in range(a.iters):
            runb()
        torch.cuda.synchronize()
        nv.lib.rtk_profile_enable(0)
        prof = nv.profile_read()
        flops = 2.0 * Hq * L * L * D * a.units
        for k, (n, ms) in prof.items():
            us = ms / n * 1e3
            extra = f"  {flops / (us * 1e-6) / 1e12:7.1f} TFLOP/s" if k.startswith("score_pass") else ""
            print(f"{k:16s} n={n:4d} avg={us:9.1f} us{extra}")
        G = Hq // Hkv
        if a.refround:   # per-head partials: the fingerprint below is for the default layout only
            return
        sc = (parts.view(a.units, Hkv, rs_n.value, L).sum(2) / G).mean(1)
        print("units", a.units, "score mean %.7f" % float(sc.mean()), "checksum %.9e" % float((sc.double() * torch.arange(1, L + 1, device=dev).double()).sum()))
        return
    for i in range(3):
        run(*sets[i % len(sets)])
    torch.cuda.synchronize()
    nv.lib.rtk_profile_reset()
    nv.lib.rtk_profile_enable(1)
    for i in range(a.iters):
        run(*sets[i % len(sets)])
    torch.cuda.synchronize()
    nv.lib.rtk_profile_enable(0)
    prof = nv.profile_read()
    flops = 2.0 * Hq * L * L * D
    for k, (n, ms) in prof.items():
        us = ms / n * 1e3
        extra = f"  {flops / (us * 1e-6) / 1e12:7.1f} TFLOP/s" if k.startswith("score_pass") else ""
        print(f"{k:16s} n={n:4d} avg={us:9.1f} us{extra}")
    print("score mean", float(score.mean()), "min", float(score.min()), "max", float(score.max()))


if __name__ == "__main__":
    main()
