"""Fixture loading + input regeneration shared by the oracle tests (CPU) and the HIP parity tests (GPU)."""
from __future__ import annotations

import glob
import os

import numpy as np

import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def names(prefix: str):
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, prefix + "*.npz")))


def load(name: str):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def dpselect_input(g) -> np.ndarray:
    """[1,T,N,C] float32, or uint16 (bf16 bits) for bf16 fixtures; verified against the stored crc."""
    if "x" in g.files:
        x = g["x"]
        return x.view(np.uint16) if str(g["dtype"]) == "bf16" else x
    kind, seed = str(g["kind"]), int(g["seed"])
    T, N, C = int(g["T"]), int(g["N"]), int(g["C"])
    if kind == "torch0":
        import torch

        torch.manual_seed(seed)
        x = torch.randn(1, T, N, C).numpy()
    else:
        x = synth.make_frames(kind, seed, T, N, C)
    assert synth.checksum(x) == int(g["x_crc"]), "regenerated input differs from the fixture's"
    return x


def pivotkv_chunk_inputs(g, c: int):
    """(q, k, v, pos, mask) for chunk c: rotated q,k as handed to update. Regenerated for big cases."""
    import torch

    pre = f"c{c}_"
    pos = g[pre + "pos"]
    mask = g[pre + "mask"]
    mask = mask if mask.size else None
    if bool(g["raw"]):
        return g[pre + "q"], g[pre + "k"], g[pre + "v"], pos, mask
    Hq, Hkv, D, L = (int(g[k]) for k in ("Hq", "Hkv", "D", "L"))
    q0, k0, v = synth.qkv_chunk(int(g["seed"]) * 100 + c, Hq, Hkv, L, D)
    rotary = synth.RotaryStub(g["inv_freq"], float(g["attention_scaling"]))
    sec = [int(s) for s in g["mrope_section"]] or None
    q = synth.rope_forward(torch.from_numpy(q0), torch.from_numpy(pos), rotary, sec).numpy()
    k = synth.rope_forward(torch.from_numpy(k0), torch.from_numpy(pos), rotary, sec).numpy()
    assert synth.checksum(q) == int(g[pre + "q_crc"]) and synth.checksum(k) == int(g[pre + "k_crc"])
    assert synth.checksum(v) == int(g[pre + "v_crc"])
    return q, k, v, pos, mask


def pivotkv_bf16_chunk_inputs(g, c: int):
    """(q, k, v) as uint16 bf16 bit patterns [1,H,L,D], pos [3,1,L], mask [L] of a bf16 fixture (the reference run on
    a bf16 model).  Regenerated from the seed for the big cases and verified against the stored crc."""
    import torch

    pre = f"c{c}_"
    pos, mask = g[pre + "pos"], g[pre + "mask"]
    if bool(g["raw"]):
        return g[pre + "q_bits"], g[pre + "k_bits"], g[pre + "v_bits"], pos, mask
    Hq, Hkv, D, L = (int(g[k]) for k in ("Hq", "Hkv", "D", "L"))
    q0, k0, v = synth.qkv_chunk(int(g["seed"]) * 100 + c, Hq, Hkv, L, D)
    rotary = synth.RotaryStub(g["inv_freq"], float(g["attention_scaling"]))
    sec = [int(s) for s in g["mrope_section"]]

    def bits(t):
        return t.bfloat16().contiguous().view(torch.int16).numpy().view(np.uint16)

    q = bits(synth.rope_forward(torch.from_numpy(q0), torch.from_numpy(pos), rotary, sec))
    k = bits(synth.rope_forward(torch.from_numpy(k0), torch.from_numpy(pos), rotary, sec))
    vb = bits(torch.from_numpy(v))
    assert synth.checksum(q) == int(g[pre + "q_crc"]) and synth.checksum(k) == int(g[pre + "k_crc"])
    assert synth.checksum(vb) == int(g[pre + "v_crc"])
    return q, k, vb, pos, mask


def bf16_ulp(x: np.ndarray) -> np.ndarray:
    """Spacing of bf16 numbers at |x| (x fp32 holding bf16 values)."""
    e = np.floor(np.log2(np.maximum(np.abs(x.astype(np.float64)), 2.0 ** -126)))
    return 2.0 ** (e - 7)
