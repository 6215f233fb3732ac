"""Deterministic synthetic inputs shared by the golden generator, the tests and bench.py.

Everything here is plain numpy / torch-CPU and independent of both the reference and the
product code.  Inputs are regenerated from (kind, seed, shape) so that large cases do not
have to be committed; fixtures carry a checksum of the regenerated array to detect drift.
"""
from __future__ import annotations

import math
import zlib

import numpy as np

YARN_FACTOR4_ATTENTION_SCALING = 0.1 * math.log(4.0) + 1.0  # HF YaRN mscale for factor 4 (=1.1386...)


def checksum(a: np.ndarray) -> int:
    """crc32 of the raw bytes (contiguous) — bit-exact input identity."""
    return zlib.crc32(np.ascontiguousarray(a).view(np.uint8).tobytes()) & 0xFFFFFFFF


def frames_iid(seed: int, T: int, N: int, C: int) -> np.ndarray:
    """i.i.d. N(0,1) frame embeddings [1,T,N,C] fp32 (BASELINE.json config shape)."""
    rng = np.random.default_rng(seed)
    return rng.standard_normal((1, T, N, C), dtype=np.float32)


def frames_video(seed: int, T: int, N: int, C: int, cut_prob: float = 0.08) -> np.ndarray:
    """Video-like embeddings: AR(1) drift per patch with random per-frame correlation and scene cuts.

    x[t] = rho[t,n] * x[t-1] + sqrt(1-rho^2) * noise;  rho~U(0.55,0.98), cut => rho=0.
    Adjacent-frame cosine then spreads over (0,1) with clear local maxima of the distance.
    """
    rng = np.random.default_rng(seed)
    x = np.empty((1, T, N, C), dtype=np.float32)
    x[0, 0] = rng.standard_normal((N, C), dtype=np.float32)
    for t in range(1, T):
        rho = rng.uniform(0.55, 0.98, size=(N, 1)).astype(np.float32)
        if rng.uniform() < cut_prob:
            rho[:] = 0.0
        noise = rng.standard_normal((N, C), dtype=np.float32)
        x[0, t] = rho * x[0, t - 1] + np.sqrt(1.0 - rho * rho) * noise
    return x


def make_frames(kind: str, seed: int, T: int, N: int, C: int) -> np.ndarray:
    if kind == "iid":
        return frames_iid(seed, T, N, C)
    if kind == "video":
        return frames_video(seed, T, N, C)
    raise ValueError(kind)


def qkv_chunk(seed: int, Hq: int, Hkv: int, L: int, D: int, scale: float = 1.7):
    """Pre-RoPE q,k and v for one (layer, chunk): scale*N(0,1), fp32. SURVEY §8(d) cfg 2."""
    rng = np.random.default_rng(seed)
    q0 = scale * rng.standard_normal((1, Hq, L, D), dtype=np.float32)
    k0 = scale * rng.standard_normal((1, Hkv, L, D), dtype=np.float32)
    v = scale * rng.standard_normal((1, Hkv, L, D), dtype=np.float32)
    return q0, k0, v


def mrope_position_ids(t0: int, n_grid_t: int, gh: int, gw: int, hw0: int = 0) -> np.ndarray:
    """Qwen2-VL style M-RoPE ids for a chunk of n_grid_t temporal grids of gh x gw tokens: [3,1,L] int64.

    temporal id = t0 + grid index, height/width ids = hw0 + row / col.
    """
    L = n_grid_t * gh * gw
    t = np.repeat(np.arange(n_grid_t), gh * gw) + t0
    h = np.tile(np.repeat(np.arange(gh), gw), n_grid_t) + hw0
    w = np.tile(np.arange(gw), n_grid_t * gh) + hw0
    return np.stack([t, h, w]).reshape(3, 1, L).astype(np.int64)


def inv_freq(D: int, theta: float = 1e6) -> np.ndarray:
    """Default RoPE inverse frequencies, fp32, [D/2] (HF: 1/theta^(arange(0,D,2)/D))."""
    return (1.0 / (theta ** (np.arange(0, D, 2, dtype=np.int64).astype(np.float32) / D))).astype(np.float32)


class RotaryStub:
    """Stand-in for HF Qwen2VLRotaryEmbedding / Qwen2RotaryEmbedding (third-party): a callable
    (x, position_ids) -> (cos, sin) carrying .attention_scaling and .inv_freq, torch-CPU or GPU.

    position_ids [3,B,L] (M-RoPE) -> cos/sin [3,B,L,D];  [B,L] -> [B,L,D].
    cos = cos(pos * inv_freq) * attention_scaling, computed in fp32 then cast to x.dtype.
    """

    def __init__(self, inv_freq_np: np.ndarray, attention_scaling: float = 1.0, device="cpu"):
        import torch

        self.inv_freq = torch.from_numpy(np.asarray(inv_freq_np, dtype=np.float32)).to(device)
        self.attention_scaling = float(attention_scaling)

    def __call__(self, x, position_ids):
        import torch

        with torch.no_grad():
            inv = self.inv_freq.to(position_ids.device).float()
            if position_ids.ndim == 3:
                B = position_ids.shape[1]
                inv_e = inv[None, None, :, None].expand(3, B, -1, 1)
                pos_e = position_ids[:, :, None, :].float()
                freqs = (inv_e @ pos_e).transpose(2, 3)
            else:
                B = position_ids.shape[0]
                inv_e = inv[None, :, None].expand(B, -1, 1)
                pos_e = position_ids[:, None, :].float()
                freqs = (inv_e @ pos_e).transpose(1, 2)
            emb = torch.cat((freqs, freqs), dim=-1)
            cos = emb.cos() * self.attention_scaling
            sin = emb.sin() * self.attention_scaling
        return cos.to(dtype=x.dtype), sin.to(dtype=x.dtype)


def rope_forward(x0, position_ids, rotary, mrope_section=None):
    """Forward (M-)RoPE of pre-RoPE x0 [1,H,L,D] (torch) at position_ids, the way the HF attention
    modules do it (third-party formula): x*cos + rotate_half(x)*sin with cos/sin from `rotary`.
    Manufactures the *rotated* q/k the attention patch hands to `PivotKVCache.update`."""
    import torch

    cos, sin = rotary(x0, position_ids)
    if mrope_section:
        sec = list(mrope_section) * 2
        cos = torch.cat([m[i % 3] for i, m in enumerate(cos.split(sec, dim=-1))], dim=-1).unsqueeze(1)
        sin = torch.cat([m[i % 3] for i, m in enumerate(sin.split(sec, dim=-1))], dim=-1).unsqueeze(1)
    else:
        cos, sin = cos.unsqueeze(1), sin.unsqueeze(1)
    D = x0.shape[-1]
    rot = torch.cat((-x0[..., D // 2:], x0[..., : D // 2]), dim=-1)
    return x0 * cos + rot * sin
