#!/usr/bin/env python3
"""MA-LLM / MA-LLM-hard at size (visual_compression.py:5-83 looped as qwen2_vl.py:402-410): a [1, T, 196, 1280] bf16
bank merged down to T/2.  Times the reference-shaped loop of single steps (every step is a pass over the bank, as in
the reference) for both variants and the one-pass chain of the hard variant, and checks chain == loop.  One JSON line.

    python tools/bench_mallm.py [--frames 2048] [--loop-frames 256]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "video-retake_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402

N, C = 196, 1280


def bank(T, dev):
    import synth

    g = torch.Generator(device=dev).manual_seed(T)
    base = torch.randn((T, N, C), generator=g, device=dev)
    # temporally correlated frames (a random walk per patch position) so that the merge order has structure
    x = torch.cumsum(0.35 * base, dim=0) + torch.randn((1, N, C), generator=g, device=dev)
    return x.bfloat16()[None].contiguous()


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    return out, time.perf_counter() - t0


def main():
    import retake.visual_compression as vc

    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=2048)
    ap.add_argument("--loop-frames", type=int, default=512, help="bank length the step-per-call loops are timed on in full")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    res = {"bank": [1, args.frames, N, C], "dtype": "bf16"}
    for sync in (False, True):
        key = "sync" if sync else "async"
        # chain vs loop, equal and timed, on the short bank
        Ts = args.loop_frames
        x = bank(Ts, dev)
        vc.memory_bank_compress_MALLM_hard_to(x, Ts // 2, sync=sync)     # warm-up (module load, LDS opt-in)

        def loop_hard():
            b = x
            while b.shape[1] > Ts // 2:
                b = vc.memory_bank_compress_MALLM_hard(b, sync=sync)
            return b

        def loop_soft():
            b, s = x, torch.ones_like(x[:, :, :, 0])
            while b.shape[1] > Ts // 2:
                b, s = vc.memory_bank_compress_MALLM(b, s, sync=sync)
            return b

        lh, t_lh = timed(loop_hard)
        ch, t_ch = timed(lambda: vc.memory_bank_compress_MALLM_hard_to(x, Ts // 2, sync=sync))
        _, t_ls = timed(loop_soft)
        assert torch.equal(lh, ch)
        r = {"short_bank_frames": Ts, "hard_loop_ms": t_lh * 1e3, "hard_chain_ms": t_ch * 1e3, "soft_loop_ms": t_ls * 1e3,
             "chain_equals_loop": True, "speedup_short": t_lh / t_ch}
        # the full bank: the chain in full; the loops by their first 32 steps (a step's cost falls linearly with the bank)
        T = args.frames
        x = bank(T, dev)
        _, t_chain = timed(lambda: vc.memory_bank_compress_MALLM_hard_to(x, T // 2, sync=sync))
        _, t_chain = timed(lambda: vc.memory_bank_compress_MALLM_hard_to(x, T // 2, sync=sync))

        def first_steps(hard, n=32):
            b, s = x, torch.ones_like(x[:, :, :, 0])
            for _ in range(n):
                if hard:
                    b = vc.memory_bank_compress_MALLM_hard(b, sync=sync)
                else:
                    b, s = vc.memory_bank_compress_MALLM(b, s, sync=sync)
            return b

        first_steps(True, 2)
        _, t_h32 = timed(lambda: first_steps(True))
        _, t_s32 = timed(lambda: first_steps(False))
        steps = T - T // 2
        # step k works on T - k frames: sum_{k < steps} (T - k) / T relative to the first step's cost
        scale = sum((T - k) / T for k in range(steps)) / sum((T - k) / T for k in range(32))
        es = 2
        r.update({"frames": T, "target": T // 2, "hard_chain_ms_full": t_chain * 1e3,
                  "hard_loop_ms_full_extrapolated": t_h32 * scale * 1e3, "soft_loop_ms_full_extrapolated": t_s32 * scale * 1e3,
                  "hard_step_ms_at_T": t_h32 / 32 * 1e3, "soft_step_ms_at_T": t_s32 / 32 * 1e3,
                  # one step reads the bank once for the cosines and once for the merge, and writes it once
                  "hard_step_GBps": 3 * T * N * C * es / (t_h32 / 32) / 1e9,
                  "soft_step_GBps": 3 * T * N * C * es / (t_s32 / 32) / 1e9,
                  "speedup_full": t_h32 * scale / t_chain,
                  # the chain moves the bank once for the cosines, (T - t) * 2 rows per patch for the new pairs, 2 x the kept rows
                  "chain_algorithmic_GB": (T * N * C * es + 2 * steps * N * C * es + 2 * (T // 2) * N * C * es) / 1e9})
        res[key] = r
        del x
        torch.cuda.empty_cache()
    print(json.dumps(res))


if __name__ == "__main__":
    main()
