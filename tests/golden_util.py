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
        return x.view(np.uint16) if str(g["dtype"]) in ("bf16", "fp16") else x
    kind, seed = str(g["kind"]), int(g["seed"])
    T, N, C = int(g["T"]), int(g["N"]), int(g["C"])
    if kind == "torch0":
        import torch

        torch.manual_seed(seed)
        x = torch.randn(1, T, N, C).numpy()
    else:
        x = synth.make_frames(kind, seed, T, N, C)
    if str(g["dtype"]) in ("bf16", "fp16"):   # the generator cast the fp32 frames with torch's round-to-nearest-even
        import torch

        xt = torch.from_numpy(x)
        x = (xt.bfloat16() if str(g["dtype"]) == "bf16" else xt.half()).view(torch.int16).numpy().view(np.uint16)
    assert synth.checksum(x) == int(g["x_crc"]), "regenerated input differs from the fixture's"
    return x


def _stencil3(rows: np.ndarray) -> np.ndarray:
    """argrelmax of visual_compression.py:121-123 / :153-156 for window 3 on rows [R,T]: the first index of a tied
    window wins, borders padded with -inf  <=>  d[i] > d[i-1] and d[i] >= d[i+1]."""
    ninf = np.full((rows.shape[0], 1), -np.inf, dtype=rows.dtype)
    return (rows > np.concatenate([ninf, rows[:, :-1]], 1)) & (rows >= np.concatenate([rows[:, 1:], ninf], 1))


def _canonical_topk(keys: np.ndarray, k: int) -> np.ndarray:
    """Larger key first, lower index first among equal keys; returned ascending (DESIGN.md §2 'Ties')."""
    return np.sort(np.lexsort((np.arange(keys.size), -keys.astype(np.float64)))[:k])


def check_dpselect_bf16(g, dis: np.ndarray, idx: np.ndarray, mask_flat: np.ndarray, eps_sync: float = 2e-6) -> dict:
    """Per-row, margin-aware comparison of a bf16 DPSelect result (dis [T,N] fp32, idx [t] / [t,N], mask [t*N]) with
    the reference's bf16 run recorded in fixture g (visual_compression.py:100-106, :142-175).

    A "row" is one decision sequence: a patch position (async) or the patch mean (sync).  Three classes:
      exact    the row's distances equal the reference's bit for bit and its k-th boundary is not an exact tie
               -> idx AND mask must equal the reference's;
      tied     distances equal, but the reference's top-k boundary falls inside exactly tied keys (bf16 distances tie
               massively; torch.topk's pick among ties is backend-specific) -> every key above the
               threshold kept by both, same key multiset, mask[pos] == peak flag of the index picked at pos;
      relaxed  some distance of the row differs (the fp32 summation order inside the bf16 rounding chain can flip the
               last bf16 bit) -> peak flags may differ ONLY at sites one of whose two stencil comparisons has a
               reference gap <= the perturbation of the compared pair, keys may differ only at perturbed entries and
               those sites, and an index outside the common picks must be such a site or lie within 2^-7 (one bf16 ulp
               at 1) of the reference's threshold key.
    In every class the result must also be the canonical top-k of ITS OWN keys (what the select kernel promises).
    Returns the class counts (the tests print them)."""
    sync = bool(g["sync"])
    t = int(g["tgt"])
    ref_dis = g["dis32"].astype(np.float32)
    T, N = ref_dis.shape
    ref_idx, ref_mask = g["idx"], g["mask"].reshape(t, N)
    mask = np.asarray(mask_flat).reshape(t, N)
    dis = np.asarray(dis, dtype=np.float32)
    if sync:
        rows_m = dis.mean(1, dtype=np.float32)[None]          # :110 dis.mean(1), fp32
        rows_r = ref_dis.mean(1, dtype=np.float32)[None]
        delta = np.abs(rows_m.astype(np.float64) - rows_r)
        flipped_src = (dis != ref_dis).any(1)[None]
        delta = np.where(flipped_src, delta + eps_sync, np.where(delta > 0, delta + eps_sync, 0.0))
        idx_rows, ref_idx_rows = idx[None], ref_idx[None]
        for n in range(1, N):                                  # :138-140: the one frame set serves every patch position
            np.testing.assert_array_equal(mask[:, n], mask[:, 0])
        mask_rows, ref_mask_rows = mask[:, 0][None], ref_mask[:, 0][None]
    else:
        rows_m, rows_r = dis.T.copy(), ref_dis.T.copy()
        delta = np.abs(rows_m.astype(np.float64) - rows_r)
        idx_rows, ref_idx_rows = idx.T, ref_idx.T
        mask_rows, ref_mask_rows = mask.T, ref_mask.T
    pk_m, pk_r = _stencil3(rows_m), _stencil3(rows_r)
    two = np.float32(2.0)
    keys_m = np.where(pk_m, rows_m + two, rows_m).astype(np.float32)   # :132-133 / :159-160, fp32 add
    keys_r = np.where(pk_r, rows_r + two, rows_r).astype(np.float32)
    stats = {"rows": rows_m.shape[0], "exact": 0, "tied": 0, "relaxed": 0, "flipped_entries": int((dis != ref_dis).sum()),
             "peak_flags_differing": 0, "indices_differing": 0}
    for r in range(rows_m.shape[0]):
        mine, theirs = idx_rows[r], ref_idx_rows[r]
        assert (np.diff(mine) > 0).all(), f"row {r}: indices not ascending / distinct"
        # (0) the product's own contract: canonical top-k of its own keys, mask = its own peak flags at the picks
        if not sync:   # the sync row's own keys come from the kernel's fp32 mean, which this numpy mean only approximates
            np.testing.assert_array_equal(mine, _canonical_topk(keys_m[r], t), err_msg=f"row {r}: not the top-k of its keys")
            np.testing.assert_array_equal(mask_rows[r], pk_m[r][mine], err_msg=f"row {r}: mask is not the peak flag")
        np.testing.assert_array_equal(ref_mask_rows[r], pk_r[r][theirs], err_msg=f"row {r}: fixture self-consistency")
        if not (delta[r] > 0).any():
            srt = np.sort(keys_r[r])[::-1]
            tie = t < T and srt[t - 1] == srt[t]
            if not tie:
                np.testing.assert_array_equal(mine, theirs, err_msg=f"row {r} (exact class): indices differ")
                np.testing.assert_array_equal(mask_rows[r], ref_mask_rows[r], err_msg=f"row {r} (exact class): mask differs")
                stats["exact"] += 1
            else:
                thr = srt[t - 1]
                above = np.nonzero(keys_r[r] > thr)[0]
                assert np.isin(above, mine).all() and np.isin(above, theirs).all(), f"row {r}: a key above the tie dropped"
                np.testing.assert_array_equal(np.sort(keys_r[r][mine]), np.sort(keys_r[r][theirs]))
                np.testing.assert_array_equal(mask_rows[r], pk_r[r][mine], err_msg=f"row {r} (tied class): mask")
                stats["tied"] += 1
                stats["indices_differing"] += int(np.setxor1d(mine, theirs).size)
            continue
        # relaxed class
        stats["relaxed"] += 1
        d = delta[r]
        gap = np.abs(np.diff(rows_r[r].astype(np.float64)))            # gap[i] = |r[i] - r[i+1]|
        may_flip = gap <= d[:-1] + d[1:]                                # comparison (i, i+1) may come out differently
        site_ok = np.zeros(T, dtype=bool)
        site_ok[:-1] |= may_flip
        site_ok[1:] |= may_flip
        bad = (pk_m[r] != pk_r[r]) & ~site_ok if not sync else np.zeros(T, dtype=bool)
        assert not bad.any(), f"row {r}: peak flag differs at {np.nonzero(bad)[0]} with no perturbed comparison there"
        P = pk_m[r] != pk_r[r] if not sync else site_ok
        stats["peak_flags_differing"] += int((pk_m[r] != pk_r[r]).sum())
        if not sync:
            moved = keys_m[r] != keys_r[r]
            assert not (moved & ~(P | (d > 0))).any(), f"row {r}: a key differs where nothing was perturbed"
        xor = np.setxor1d(mine, theirs)
        stats["indices_differing"] += int(xor.size)
        if xor.size:
            thr = np.sort(keys_r[r])[::-1][min(t, T) - 1]
            near = np.abs(keys_r[r][xor].astype(np.float64) - thr) <= 2.0 ** -7
            own = P[xor] | (d[xor] > 0)
            assert (near | own).all(), f"row {r}: index {xor[~(near | own)]} changed sides far from the threshold"
            assert xor.size <= 2 * int(P.sum() + (d > 0).sum()), f"row {r}: {xor.size} picks differ for {int((d > 0).sum())} flips"
        # common picks carry the same peak flag unless the site itself is perturbed
        common = np.intersect1d(mine, theirs)
        mm = dict(zip(mine.tolist(), mask_rows[r].tolist()))
        rm = dict(zip(theirs.tolist(), ref_mask_rows[r].tolist()))
        for j in common.tolist():
            assert mm[j] == rm[j] or P[j], f"row {r}: mask differs at frame {j}, not a perturbed stencil site"
    return stats


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


def pivotkv_fp16_chunk_inputs(g):
    """(q, k, v) as numpy float16 [1,H,L,D], pos [3,1,L], mask [L] of an fp16 fixture (the reference run on a float16
    model, one chunk).  Regenerated from the seed for the big cases and verified against the stored crc."""
    import torch

    pos, mask = g["c0_pos"], g["c0_mask"]
    if bool(g["raw"]):
        return tuple(g["c0_" + n + "_bits"].view(np.float16) for n in ("q", "k", "v")) + (pos, mask)
    Hq, Hkv, D, L = (int(g[k]) for k in ("Hq", "Hkv", "D", "L"))
    q0, k0, v = synth.qkv_chunk(int(g["seed"]) * 100, Hq, Hkv, L, D)
    rotary = synth.RotaryStub(g["inv_freq"], float(g["attention_scaling"]))
    sec = [int(s) for s in g["mrope_section"]]

    def bits(t):
        return t.half().contiguous().view(torch.int16).numpy().view(np.uint16)

    q = bits(synth.rope_forward(torch.from_numpy(q0), torch.from_numpy(pos), rotary, sec))
    k = bits(synth.rope_forward(torch.from_numpy(k0), torch.from_numpy(pos), rotary, sec))
    vb = bits(torch.from_numpy(v))
    assert synth.checksum(q) == int(g["c0_q_crc"]) and synth.checksum(k) == int(g["c0_k_crc"])
    assert synth.checksum(vb) == int(g["c0_v_crc"])
    return q.view(np.float16), k.view(np.float16), vb.view(np.float16), pos, mask


def fp16_ulp(x: np.ndarray) -> np.ndarray:
    """Spacing of fp16 numbers at |x| (normal range)."""
    e = np.floor(np.log2(np.maximum(np.abs(np.asarray(x, dtype=np.float64)), 2.0 ** -14)))
    return 2.0 ** (e - 10)


# ---------------------------------------------------------------------------------------------------
# round 5: fixtures of the reference's whole attention-side chain on a bf16 model, from the bf16 PRE-RoPE projections
# (gen_golden.py --only pivotkv_prerope_bf16)
# ---------------------------------------------------------------------------------------------------
def pivotkv_prerope_chunk_inputs(g, c: int):
    """(q0, k0, v) as uint16 bf16 bit patterns [1,H,L,D] - the PRE-RoPE projections of a bf16 model -, pos_in (the ids the
    caller hands to the attention layer), pos (after the continuity shift: what the reference rotates with), mask [L]."""
    import torch

    pre = f"c{c}_"
    pos_in, pos, mask = g[pre + "pos_in"], g[pre + "pos"], g[pre + "mask"]
    if bool(g["raw"]):
        return g[pre + "q0_bits"], g[pre + "k0_bits"], g[pre + "v_bits"], pos_in, pos, mask
    Hq, Hkv, D, L = (int(g[k]) for k in ("Hq", "Hkv", "D", "L"))

    tdt = torch.float16 if str(g["dtype"]) == "fp16" else torch.bfloat16

    def bits(a):
        return torch.from_numpy(a).to(tdt).contiguous().view(torch.int16).numpy().view(np.uint16)

    q0, k0, v = (bits(a) for a in synth.qkv_chunk(int(g["seed"]) * 100 + c, Hq, Hkv, L, D))
    assert synth.checksum(q0) == int(g[pre + "q0_crc"]) and synth.checksum(k0) == int(g[pre + "k0_crc"])
    assert synth.checksum(v) == int(g[pre + "v_crc"])
    return q0, k0, v, pos_in, pos, mask


def rotate_like_a_bf16_model(g, c: int, q0_bits: np.ndarray, k0_bits: np.ndarray):
    """The rotated q, k a bf16 (fp16: fixtures with dtype "fp16") HF model hands to PivotKVCache.update (uint16 bits): the
    rotary module's tables rounded to the model dtype, then (x*cos) + (rotate_half(x)*sin) with one rounding per torch op (longvideo_cache.py:68-81 / :109-114).
    Verified against the crc of the tensors the reference's own helper produced at generation time."""
    import torch

    pre = f"c{c}_"
    pos = torch.from_numpy(g[pre + "pos"])
    sec = [int(s) for s in g["mrope_section"]]
    rotary = synth.RotaryStub(g["inv_freq"], float(g["attention_scaling"]))

    tdt = torch.float16 if str(g["dtype"]) == "fp16" else torch.bfloat16

    def bf(bits):
        return torch.from_numpy(bits.view(np.int16)).view(tdt)

    q0, k0 = bf(q0_bits), bf(k0_bits)
    cos, sin = rotary(k0, pos)
    if sec:
        s2 = sec * 2
        cos = torch.cat([m[i % 3] for i, m in enumerate(cos.split(s2, dim=-1))], dim=-1).unsqueeze(1)
        sin = torch.cat([m[i % 3] for i, m in enumerate(sin.split(s2, dim=-1))], dim=-1).unsqueeze(1)
    else:
        cos, sin = cos.unsqueeze(1), sin.unsqueeze(1)

    def rot_half(x):
        h = x.shape[-1] // 2
        return torch.cat((-x[..., h:], x[..., :h]), dim=-1)

    out = []
    for x, name in ((q0, "q_rot_crc"), (k0, "k_rot_crc")):
        y = ((x * cos) + (rot_half(x) * sin)).contiguous().view(torch.int16).numpy().view(np.uint16)
        assert synth.checksum(y) == int(g[pre + name]), "restated bf16 rotation differs from the reference helper's output"
        out.append(y)
    return out[0], out[1]
