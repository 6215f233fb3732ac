#!/usr/bin/env python3
"""SURVEY §8(f)1 - "fuse PivotKV scoring into the chunk's attention": what could the two contractions share?

CPU-only evidence behind DESIGN.md §8 (no reference code, no GPU):

  1. logits.   With `pos_embed_reforge` (every shipped config: configs/retake_demo.yaml:21, configs/qwen2_vl/*.yaml) the
     score contracts the UN-rotated q~ k~ (longvideo_cache.py:248-264) while the layer's attention contracts the rotated
     q k (qwen2_vl.py:224-363).  For one chunk of M-RoPE ids this script builds both logit matrices from the same
     pre-RoPE contents and reports their correlation and rms difference: if they are not the same matrix there is no
     Q K^T to share, only the operand reads.
  2. flops.    scoring = 2 passes x 2 Hq L^2 D per (layer, chunk); attention over [prefix | chunk] with a causal mask =
     2 (QK^T, PV) x 2 Hq L (P + L/2) D with P compressed prefix tokens.  Printed per chunk and over a whole video.
  3. reforge off: the logits coincide, but attention is causal over the prefix and the chunk and normalised over that
     range, the score is the UNMASKED chunk-local softmax: only the lower triangle of ONE of the score's two
     contractions is reusable (the row normaliser still needs the upper triangle) -> the reusable share is printed.

    python tools/f1_logit_overlap.py [--grids 8 --gh 14 --gw 14 --seed 0]
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth  # noqa: E402  (plain numpy / torch-CPU input generators; independent of reference and product)


def logit_overlap(grids=8, gh=14, gw=14, D=128, seed=0, scale=1.7, a=synth.YARN_FACTOR4_ATTENTION_SCALING,
                  mrope=(16, 24, 24), t0=16):
    """One head, one chunk: (corr, rms difference, logit std) between the un-rotated and the rotated logit matrices."""
    import torch

    L = grids * gh * gw
    q0, k0, _ = synth.qkv_chunk(seed, 1, 1, L, D, scale)
    pos = torch.from_numpy(synth.mrope_position_ids(t0, grids, gh, gw, hw0=t0))
    rot = synth.RotaryStub(synth.inv_freq(D), a)
    q = synth.rope_forward(torch.from_numpy(q0), pos, rot, list(mrope))[0, 0].double().numpy()
    k = synth.rope_forward(torch.from_numpy(k0), pos, rot, list(mrope))[0, 0].double().numpy()
    s_unrot = q0[0, 0].astype(np.float64) @ k0[0, 0].astype(np.float64).T / math.sqrt(D)   # what the score contracts
    s_rot = q @ k.T / math.sqrt(D) / (a * a)          # what the attention contracts, YaRN's a^2 divided out for the comparison
    corr = float(np.corrcoef(s_unrot.ravel(), s_rot.ravel())[0, 1])
    rms = float(np.sqrt(np.mean((s_unrot - s_rot) ** 2)))
    # the decision the score feeds: would the rotated logits keep the same tokens?
    def colmass(s):
        p = np.exp(s - s.max(1, keepdims=True))
        return (p / p.sum(1, keepdims=True)).sum(0)
    keep = max(1, L // 4)
    a_set, b_set = np.argsort(-colmass(s_unrot))[:keep], np.argsort(-colmass(s_rot))[:keep]
    return {"L": L, "D": D, "corr": corr, "rms_diff": rms, "logit_std": float(s_unrot.std()),
            "kept_set_overlap_at_ratio_0.25": float(np.intersect1d(a_set, b_set).size / keep)}


def flop_shares(L=6272, keep=1568, chunks=64, Hq=28, D=128):
    """Scoring flops relative to the layer's attention flops, per chunk index and over the video."""
    score = 2 * 2 * Hq * L * L * D
    rows = []
    tot_s = tot_a = 0.0
    for c in range(chunks):
        P = c * keep
        attn = 2 * 2 * Hq * L * (P + L / 2) * D
        rows.append(score / attn)
        tot_s += score
        tot_a += attn
    # reforge off: the lower triangle of one of the two score contractions equals the chunk-local part of the causal QK^T
    reusable = 0.25
    return {"L": L, "keep": keep, "chunks": chunks, "score_over_attention_first_chunk": rows[0],
            "score_over_attention_last_chunk": rows[-1], "score_over_attention_video": tot_s / tot_a,
            "reusable_share_of_scoring_reforge_off": reusable,
            "ceiling_saving_vs_attention_reforge_off": reusable * tot_s / tot_a, "ceiling_saving_reforge_on": 0.0}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grids", type=int, default=8)
    ap.add_argument("--gh", type=int, default=14)
    ap.add_argument("--gw", type=int, default=14)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    out = {"logits_reforge_on": logit_overlap(a.grids, a.gh, a.gw, seed=a.seed),
           "flops_baseline_geometry": flop_shares(),
           "flops_qwen448_geometry": flop_shares(L=2304, keep=576)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
