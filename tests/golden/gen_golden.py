#!/usr/bin/env python3
"""Golden-vector generator for the DPSelect + PivotKV hot path.

Runs ONLY in the build container: it imports the Python reference from /root/reference
(read-only, never copied) and records inputs/outputs as small .npz fixtures in this
directory.  The GPU box and the test-suite never import the reference; they read the
fixtures.  Usage:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden.py [--only dpselect|pivotkv|glue]

Shims (generator-side only):
  * transformers.cache_utils.DynamicCache is replaced *before import* by a stand-in with
    transformers-4.48 semantics (key_cache/value_cache python lists, update = append or
    torch.cat(dim=-2)) because the container ships transformers 5.x whose DynamicCache
    is layer-object based (reference pins 4.48, environment.yaml:9).
  * rotary_emb is tests/synth.RotaryStub (any callable (x,pos)->(cos,sin) with
    .attention_scaling is accepted by longvideo_cache.py:248-259).

Every fixture stores, next to the reference outputs, an fp64 recomputation of the decision
variables and the smallest decision margins, and the generator REFUSES to write a fixture
that contains a fragile decision (margin < FRAGILE) so that index parity is well defined.
"""
from __future__ import annotations

import argparse
import os
import sys
import types

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))  # tests/
import synth  # noqa: E402

REF = "/root/reference"
FRAGILE = 2e-6  # absolute margin (fp64) below which a decision is considered fragile in fp32

torch.set_num_threads(8)


# --------------------------------------------------------------------------------------
# reference import with shims
# --------------------------------------------------------------------------------------
class DynamicCache448:
    """transformers==4.48 DynamicCache public behaviour (third-party), minimal restatement."""

    def __init__(self, *a, **k):
        self.key_cache = []
        self.value_cache = []
        self._seen_tokens = 0

    def update(self, key_states, value_states, layer_idx, cache_kwargs=None):
        if layer_idx == 0:
            self._seen_tokens += key_states.shape[-2]
        if len(self.key_cache) <= layer_idx:
            for _ in range(len(self.key_cache), layer_idx):
                self.key_cache.append([])
                self.value_cache.append([])
            self.key_cache.append(key_states)
            self.value_cache.append(value_states)
        elif len(self.key_cache[layer_idx]) == 0:
            self.key_cache[layer_idx] = key_states
            self.value_cache[layer_idx] = value_states
        else:
            self.key_cache[layer_idx] = torch.cat([self.key_cache[layer_idx], key_states], dim=-2)
            self.value_cache[layer_idx] = torch.cat([self.value_cache[layer_idx], value_states], dim=-2)
        return self.key_cache[layer_idx], self.value_cache[layer_idx]

    # what the FlashAttention-2 patch asks the cache (qwen2_vl.py:245-252, :270): 4.48's DynamicCache answers
    def get_seq_length(self, layer_idx=0):
        if len(self.key_cache) <= layer_idx or len(self.key_cache[layer_idx]) == 0:
            return 0
        return self.key_cache[layer_idx].shape[-2]

    def get_max_cache_shape(self):
        return None

    def __getitem__(self, layer_idx):
        return self.key_cache[layer_idx], self.value_cache[layer_idx]

    def get_usable_length(self, new_seq_length, layer_idx=0):
        return self.get_seq_length(layer_idx)   # no maximum length: the previous length


def import_reference():
    import transformers.cache_utils as cu

    cu.DynamicCache = DynamicCache448
    sys.path.insert(0, REF)
    import importlib

    vc = importlib.import_module("retake.visual_compression")
    lc = importlib.import_module("retake.longvideo_cache")
    assert lc.__file__.startswith(REF), lc.__file__
    return vc, lc


def import_reference_glue():
    """retake/qwen2_vl.py needs two symbols that transformers 5.x no longer has."""
    import builtins

    import transformers.models.qwen2_vl.modeling_qwen2_vl as m

    if not hasattr(m, "Qwen2VLSdpaAttention"):
        m.Qwen2VLSdpaAttention = type("Qwen2VLSdpaAttention", (), {})
    if not hasattr(m, "Qwen2VLCausalLMOutputWithPast"):
        m.Qwen2VLCausalLMOutputWithPast = type("Qwen2VLCausalLMOutputWithPast", (), {})
    if not hasattr(m, "repeat_kv"):
        m.repeat_kv = lambda x, n: x
    if not hasattr(m, "apply_multimodal_rotary_pos_emb"):
        m.apply_multimodal_rotary_pos_emb = lambda *a, **k: None
    builtins.FlashAttentionKwargs = dict
    import importlib

    return importlib.import_module("retake.qwen2_vl")


# --------------------------------------------------------------------------------------
# DPSelect
# --------------------------------------------------------------------------------------
def dis_matrices(x: torch.Tensor):
    """dis [T,N] exactly as the reference's first three statements compute it (fp32 path of
    the input dtype), plus an fp64 recomputation for margins."""
    import torch.nn.functional as F

    sim = F.cosine_similarity(x[:, :-1, :], x[:, 1:, :], dim=-1)
    d = 1 - sim[0].type(torch.float)
    d32 = torch.cat([torch.ones_like(d[:1]), d], dim=0)
    x64 = x.double()
    sim64 = F.cosine_similarity(x64[:, :-1, :], x64[:, 1:, :], dim=-1)
    d64 = torch.cat([torch.ones_like(sim64[0][:1]), 1 - sim64[0]], dim=0)
    return d32.numpy(), d64.numpy()


def peak_margins(d64_rows: np.ndarray) -> float:
    """Smallest |d[i]-d[i±1]| over all stencil comparisons (rows = independent sequences)."""
    if d64_rows.shape[1] < 2:
        return np.inf
    diff = np.abs(np.diff(d64_rows, axis=1))
    return float(diff.min())


def topk_margin(keys64_rows: np.ndarray, k: int) -> float:
    """Smallest gap between the k-th and (k+1)-th largest key over rows."""
    n = keys64_rows.shape[1]
    if k >= n:
        return np.inf
    s = -np.sort(-keys64_rows, axis=1)
    return float((s[:, k - 1] - s[:, k]).min())


def recover_frame_idx(x: np.ndarray, out: np.ndarray) -> np.ndarray:
    """idx[t',n] such that out[0,t',n,:] == x[0,idx,n,:] bit-exactly (inputs have unique rows)."""
    T, N = x.shape[1], x.shape[2]
    t = out.shape[1]
    idx = np.empty((t, N), dtype=np.int64)
    for n in range(N):
        xs = x[0, :, n, :]
        os_ = out[0, :, n, :]
        for tt in range(t):
            m = np.nonzero((xs[:, :4] == os_[tt, :4]).all(axis=1))[0]
            m = [i for i in m if np.array_equal(xs[i], os_[tt])]
            assert len(m) == 1, (n, tt, m)
            idx[tt, n] = m[0]
    return idx


def gen_dpselect(vc, outdir):
    cases = []
    # (name, kind, seed, T, N, C, tgt, sync, dtype, store_raw)
    # BASELINE.json configs[0]: 64 random 1280-d frame embeddings, sync (N=1 async is a reference crash)
    cases += [("cfg1_t32", "torch0", 0, 64, 1, 1280, 32, True, "fp32", True),
              ("cfg1_t64", "torch0", 0, 64, 1, 1280, 64, True, "fp32", True),
              ("cfg1_t16", "torch0", 0, 64, 1, 1280, 16, True, "fp32", True)]
    cases += [("iid64x196_sync_r50", "iid", 11, 64, 196, 1280, 32, True, "fp32", False),
              ("iid64x196_async_r50", "iid", 11, 64, 196, 1280, 32, False, "fp32", False),
              ("iid64x196_async_r100", "iid", 11, 64, 196, 1280, 64, False, "fp32", False),
              ("iid64x196_async_r25", "iid", 12, 64, 196, 1280, 16, False, "fp32", False)]
    cases += [("video64x16x64_async_r50", "video", 21, 64, 16, 64, 32, False, "fp32", True),
              ("video64x16x64_async_r25", "video", 22, 64, 16, 64, 16, False, "fp32", True),
              ("video64x16x64_sync_r25", "video", 23, 64, 16, 64, 16, True, "fp32", True),
              ("video37x5x36_async_t9", "video", 24, 37, 5, 36, 9, False, "fp32", True),
              ("video8x4x32_async_t1", "video", 25, 8, 4, 32, 1, False, "fp32", True),
              ("video2x3x32_async_t1", "video", 26, 2, 3, 32, 1, False, "fp32", True),
              ("video2x3x32_sync_t2", "video", 27, 2, 3, 32, 2, True, "fp32", True),
              ("video128x144x3584_async_r50", "video", 31, 128, 144, 3584, 64, False, "fp32", False),
              ("video256x196x1280_async_r100", "video", 32, 256, 196, 1280, 256, False, "fp32", False),
              ("video256x196x1280_sync_r50", "video", 33, 256, 196, 1280, 128, True, "fp32", False),
              ("llava48x729x1152_async_r50", "video", 34, 48, 729, 1152, 24, False, "fp32", False)]
    cases += [("video64x16x64_async_r50_bf16", "video", 41, 64, 16, 64, 32, False, "bf16", True),
              ("video64x16x64_sync_r50_bf16", "video", 42, 64, 16, 64, 32, True, "bf16", True)]
    # production dtype at production shapes (round 3): the key-patch mask of the async ratio-1.0 call is DPSelect's only
    # product in the shipped configs; BASELINE patch geometry 196 x 1280 and Qwen2-VL's 144 x 3584
    cases += [("video64x196x1280_async_r100_bf16", "video", 43, 64, 196, 1280, 64, False, "bf16", False),
              ("video64x196x1280_async_r50_bf16", "video", 44, 64, 196, 1280, 32, False, "bf16", False),
              ("video32x144x3584_async_r100_bf16", "video", 45, 32, 144, 3584, 32, False, "bf16", False),
              ("video32x144x3584_async_r50_bf16", "video", 46, 32, 144, 3584, 16, False, "bf16", False),
              ("video64x196x1280_sync_r50_bf16", "video", 47, 64, 196, 1280, 32, True, "bf16", False)]

    for (name, kind, seed, T, N, C, tgt, sync, dtype, raw) in cases:
        for attempt in range(50):
            sd = seed + 1000 * attempt
            if kind == "torch0":
                torch.manual_seed(sd)
                x = torch.randn(1, T, N, C)
            else:
                x = torch.from_numpy(synth.make_frames(kind, sd, T, N, C))
            if dtype == "bf16":
                x = x.bfloat16()
            d32, d64 = dis_matrices(x)
            out, mask = vc.memory_bank_compress_keyframe(x.clone(), tgt, 3, sync=sync)
            xin = x.float().numpy()
            outn = out.float().numpy()
            if sync:
                idx = recover_frame_idx(xin[:, :, :1], outn[:, :, :1])[:, 0]
                rows32 = d32.mean(1, keepdims=True).T  # [1,T] (fp32 mean, informative only)
                rows64 = d64.mean(1, keepdims=True).T
            else:
                idx = recover_frame_idx(xin, outn)
                rows64 = d64.T
            # margins on the fp64 decision variables
            pm = peak_margins(rows64)
            pk = (rows64 > np.concatenate([np.full((rows64.shape[0], 1), -np.inf), rows64[:, :-1]], 1)) & \
                 (rows64 >= np.concatenate([rows64[:, 1:], np.full((rows64.shape[0], 1), -np.inf)], 1))
            keys64 = rows64 + 2.0 * pk
            tm = topk_margin(keys64, tgt)
            if dtype == "fp32" and min(pm, tm) < FRAGILE:
                print(f"  [{name}] seed {sd}: fragile (peak gap {pm:.2e}, topk gap {tm:.2e}) -> reseed")
                continue
            break
        else:
            raise RuntimeError(name)
        rec = dict(kind=kind, seed=sd, T=T, N=N, C=C, tgt=tgt, sync=sync, window=3, dtype=dtype,
                   x_crc=synth.checksum(x.view(torch.int16).numpy() if dtype == "bf16" else xin),
                   idx=idx, mask=mask.numpy(), dis32=d32, dis64=d64,
                   out_crc=synth.checksum(out.view(torch.int16).numpy() if dtype == "bf16" else outn),
                   min_peak_gap=pm, min_topk_gap=tm)
        if raw:
            rec["x"] = x.view(torch.int16).numpy() if dtype == "bf16" else xin
        np.savez_compressed(os.path.join(outdir, f"dpselect_{name}.npz"), **rec)
        print(f"dpselect_{name}: seed {sd} t={tgt} sync={sync} peak_gap={pm:.2e} topk_gap={tm:.2e} "
              f"mask_rate={mask.float().mean():.3f}")

    # ---- hand-built edge cases (ties / zeros): compared by the documented rules ----
    # plateau ties in dis (first-index rule): frames repeat in runs, each copy scaled by a
    # distinct power of two so frames stay unique (indices recoverable) while the normalised vectors,
    # hence the cosines, are bit-identical between pairs with the same pattern -> exact ties in dis.
    rng = np.random.default_rng(5)
    base = rng.standard_normal((6, 3, 16)).astype(np.float32)
    order = [0, 1, 1, 2, 3, 3, 3, 4, 0, 0, 5, 1, 2, 2, 4, 5, 3, 3, 0, 0]
    T = len(order)
    scale = (2.0 ** (np.arange(T) - 10)).astype(np.float32)
    x = torch.from_numpy(base[order] * scale[:, None, None])[None]
    x[0, 3, 1] = 0.0  # a zero vector: cos = 0 -> dis = 1
    for sync in (True, False):
        for tgt in (5, T):
            d32, d64 = dis_matrices(x)
            out, mask = vc.memory_bank_compress_keyframe(x.clone(), tgt, 3, sync=sync)
            xin, outn = x.numpy(), out.numpy()
            if sync:
                idx = recover_frame_idx(xin[:, :, :1], outn[:, :, :1])[:, 0]
            else:
                # patch 1 of frame 3 is all-zero and unique, fine
                idx = recover_frame_idx(xin, outn)
            np.savez_compressed(os.path.join(outdir, f"dpselect_edge_plateau_{'sync' if sync else 'async'}_t{tgt}.npz"),
                                kind="raw", T=T, N=3, C=16, tgt=tgt, sync=sync, window=3, dtype="fp32",
                                x=xin, out=outn, idx=idx, mask=mask.numpy(), dis32=d32, dis64=d64)
            print(f"dpselect_edge_plateau sync={sync} t={tgt}: mask_sum={int(mask.sum())} "
                  f"ties_in_dis={int((np.diff(np.sort(d32, axis=0), axis=0) == 0).sum())}")

    # N=1 async is a reference crash - record the exception type
    try:
        vc.memory_bank_compress_keyframe(torch.randn(1, 8, 1, 16), 4, 3, sync=False)
        crash = "none"
    except Exception as e:  # noqa: BLE001
        crash = type(e).__name__
    np.savez(os.path.join(outdir, "dpselect_edge_n1_async.npz"), exception=crash)
    print("dpselect N=1 async ->", crash)


def gen_dpselect_degenerate(vc, outdir):
    """Degenerate calls of visual_compression.py: which exception class each one dies with, and what the calls that do
    return hand back (tgt_mem_len 0, batch size 2).  One fixture: dpselect_edge_degenerate.npz."""
    def outcome(f):
        try:
            f()
            return "none"
        except Exception as e:  # noqa: BLE001
            return type(e).__name__

    torch.manual_seed(7)
    x1 = torch.randn(1, 1, 4, 8)
    x = torch.randn(1, 6, 4, 8)
    rec = dict(
        keyframe_T1_sync=outcome(lambda: vc.memory_bank_compress_keyframe(x1.clone(), 1, 3, True)),
        keyframe_T1_async=outcome(lambda: vc.memory_bank_compress_keyframe(x1.clone(), 1, 3, False)),
        keyframe_T1_N1_async=outcome(lambda: vc.memory_bank_compress_keyframe(x1[:, :, :1].clone(), 1, 3, False)),
        keyframe_tgt_gt_T_sync=outcome(lambda: vc.memory_bank_compress_keyframe(x.clone(), 7, 3, True)),
        keyframe_tgt_gt_T_async=outcome(lambda: vc.memory_bank_compress_keyframe(x.clone(), 7, 3, False)),
        keyframe_tgt_neg_sync=outcome(lambda: vc.memory_bank_compress_keyframe(x.clone(), -1, 3, True)),
        keyframe_tgt_gt_T_N1_async=outcome(lambda: vc.memory_bank_compress_keyframe(x[:, :, :1].clone(), 7, 3, False)),
        mallm_T1=outcome(lambda: vc.memory_bank_compress_MALLM(x1.clone(), torch.ones(1, 1, 4))),
        mallm_hard_T1=outcome(lambda: vc.memory_bank_compress_MALLM_hard(x1.clone())))
    for sync in (True, False):
        tag = "sync" if sync else "async"
        o, m = vc.memory_bank_compress_keyframe(x.clone(), 0, 3, sync)
        rec[f"tgt0_{tag}_out_shape"] = np.array(o.shape)
        rec[f"tgt0_{tag}_mask_shape"] = np.array(m.shape)
        rec[f"tgt0_{tag}_mask_dtype"] = str(m.dtype)
    xb = torch.from_numpy(synth.make_frames("video", 71, 12, 5, 24)).repeat(2, 1, 1, 1)
    xb[1] = torch.from_numpy(synth.make_frames("video", 72, 12, 5, 24))[0]
    rec["xb"] = xb.numpy()
    for sync in (True, False):
        tag = "sync" if sync else "async"
        o, m = vc.memory_bank_compress_keyframe(xb.clone(), 5, 3, sync)
        rec[f"b2_{tag}_out"] = o.numpy()
        rec[f"b2_{tag}_mask"] = m.numpy()
    np.savez_compressed(os.path.join(outdir, "dpselect_edge_degenerate.npz"), **rec)
    for k, v in rec.items():
        print("dpselect_edge_degenerate", k, v if isinstance(v, str) else getattr(v, "shape", v))


# --------------------------------------------------------------------------------------
# PivotKV
# --------------------------------------------------------------------------------------
class Cfg(types.SimpleNamespace):
    pass


def make_config(Hq, Hkv, D, layers, ratio, reforge, llava=False):
    llm = Cfg(hidden_size=Hq * D, num_hidden_layers=layers, num_attention_heads=Hq, num_key_value_heads=Hkv)
    kw = {"kvcache_compression": True,
          "kvcache_compression_kwargs": {"compression_ratio": ratio, "compression_method": "pivotkv",
                                         "pos_embed_reforge": reforge}}
    if llava:
        return Cfg(text_config=llm, longvideo_kwargs=kw)
    llm.longvideo_kwargs = kw
    return llm


def score_fp64(q, k, Hkv):
    """fp64 restatement of longvideo_cache.py:260-270 on the tensors handed to the matmul."""
    q = q.double()[0]
    k = k.double()[0]
    Hq, L, D = q.shape
    G = Hq // Hkv
    kr = k[:, None].expand(Hkv, G, L, D).reshape(Hq, L, D)
    s = torch.matmul(q, kr.transpose(1, 2)) / np.sqrt(D)
    p = torch.softmax(s, dim=-1)
    w = p.sum(1).reshape(Hkv, G, L).mean(1).mean(0)
    return w.numpy()


def gen_pivotkv(lc, outdir, only=None):
    # name, Hq, Hkv, D, gh, gw, grids/chunk, chunks, ratio, reforge, mrope, a, mask_rate, seed, layer, raw
    S = synth.YARN_FACTOR4_ATTENTION_SCALING
    cases = [
        ("small_mrope_reforge", 4, 2, 32, 4, 4, 4, 3, 0.25, True, [4, 6, 6], S, 0.3, 101, 0, True),
        ("small_mrope_reforge_nomask", 4, 2, 32, 4, 4, 4, 3, 0.5, True, [4, 6, 6], 1.0, 0.0, 102, 0, True),
        ("small_mrope_noreforge", 4, 2, 32, 4, 4, 4, 3, 0.25, False, [4, 6, 6], S, 0.3, 103, 0, True),
        ("small_rope1d_reforge", 4, 2, 32, 4, 4, 4, 3, 0.25, True, None, S, 0.3, 104, 0, True),
        ("small_rope1d_noreforge", 4, 2, 32, 4, 4, 4, 2, 0.4, False, None, 1.0, 0.2, 105, 0, True),
        ("small_layer2_first", 4, 2, 32, 4, 4, 4, 2, 0.25, True, [4, 6, 6], S, 0.3, 106, 2, True),
        ("small_gqa7_ragged", 14, 2, 64, 3, 5, 5, 3, 0.3, True, [8, 12, 12], S, 0.25, 107, 0, True),
        ("small_keep1", 4, 2, 32, 2, 2, 2, 2, 0.01, True, [4, 6, 6], S, 0.0, 108, 0, True),
        # all key patches kept (mask score 1.0 beats the unmasked tokens below 1.0), tie-free boundary
        ("small_masked_kept", 4, 2, 32, 4, 4, 4, 2, 0.8, True, [4, 6, 6], S, 0.2, 109, 0, True),
        # boundary falls inside the exact 1.0 ties created by masked_fill_: name contains 'tie'
        ("small_tie_boundary", 4, 2, 32, 4, 4, 4, 2, 0.5, True, [4, 6, 6], S, 0.6, 110, 0, True),
        ("qwen_L256", 28, 4, 128, 8, 8, 4, 2, 0.25, True, [16, 24, 24], S, 0.3, 111, 0, False),
        ("qwen_L2304", 28, 4, 128, 9, 16, 16, 1, 0.25, True, [16, 24, 24], S, 0.3, 112, 0, False),
        ("llava_L392", 28, 4, 128, 14, 14, 2, 2, 0.25, True, None, S, 0.3, 113, 0, False),
        # compression_ratio 1 - what `dynamic_compression_ratio` sets for every prompt within max_input_length
        # (qwen2_vl.py:553-554; all shipped configs): the whole chunk is kept, K still goes through the un-rotate /
        # re-rotate round trip, the id rescale multiplies by 1.0
        ("small_ratio1_mrope_reforge", 4, 2, 32, 4, 4, 4, 3, 1, True, [4, 6, 6], S, 0.3, 114, 0, True),
        ("small_ratio1_rope1d_noreforge", 4, 2, 32, 4, 4, 4, 2, 1, False, None, 1.0, 0.2, 115, 0, True),
        ("qwen_L576_ratio1", 28, 4, 128, 12, 12, 4, 2, 1, True, [16, 24, 24], S, 0.3, 116, 0, False),
        # round 5: chunks over the L >= 512 gate of the attention prologue (PivotKVCache.update_pre_rope takes them), two
        # chunks each, both RoPE flavours, and the synthetic headline geometry L = 6272
        ("qwen_L1568", 28, 4, 128, 14, 14, 8, 2, 0.25, True, [16, 24, 24], S, 0.3, 117, 0, False),
        ("llava_L1568", 28, 4, 128, 14, 14, 8, 2, 0.25, True, None, S, 0.3, 118, 0, False),
        ("qwen_L6272", 28, 4, 128, 14, 14, 32, 1, 0.25, True, [16, 24, 24], S, 0.3, 119, 0, False),
    ]
    if only:
        cases = [c for c in cases if c[0] in only]
    for (name, Hq, Hkv, D, gh, gw, gpc, nch, ratio, reforge, mrope, a, mrate, seed, layer, raw) in cases:
        for attempt in range(50):
            sd = seed + 1000 * attempt
            rec = run_pivotkv_case(lc, Hq, Hkv, D, gh, gw, gpc, nch, ratio, reforge, mrope, a, mrate, sd, layer, raw,
                                   allow_tie=("tie" in name))
            if rec is not None:
                break
            print(f"  [{name}] seed {sd}: fragile -> reseed")
        else:
            raise RuntimeError(name)
        np.savez_compressed(os.path.join(outdir, f"pivotkv_{name}.npz"), **rec)
        print(f"pivotkv_{name}: seed {sd} L={gpc * gh * gw} keep={rec['keep']} min_topk_gap={rec['min_topk_gap']:.2e} "
              f"masked_kept={rec['masked_kept']}")


def run_pivotkv_case(lc, Hq, Hkv, D, gh, gw, gpc, nch, ratio, reforge, mrope, a, mrate, seed, layer, raw,
                     allow_tie=False):
    L = gpc * gh * gw
    inv_f = synth.inv_freq(D, 1e6)
    rotary = synth.RotaryStub(inv_f, a)
    cfg = make_config(Hq, Hkv, D, layer + 1, ratio, reforge, llava=(mrope is None))
    cache = lc.PivotKVCache(cfg)
    rec = dict(Hq=Hq, Hkv=Hkv, D=D, L=L, gh=gh, gw=gw, grids_per_chunk=gpc, n_chunks=nch, ratio=ratio,
               reforge=reforge, mrope_section=np.array(mrope if mrope else [], dtype=np.int64),
               attention_scaling=a, inv_freq=inv_f, seed=seed, layer=layer, raw=raw, theta=1e6)
    min_gap = np.inf
    masked_kept = 0
    rng = np.random.default_rng(seed + 7)
    captured = {}
    orig_topk = torch.Tensor.topk

    def spy_topk(self, *args, **kw):
        captured["score"] = self.detach().clone()
        return orig_topk(self, *args, **kw)

    for c in range(nch):
        q0, k0, v = synth.qkv_chunk(seed * 100 + c, Hq, Hkv, L, D)
        q0, k0, v = map(torch.from_numpy, (q0, k0, v))
        if mrope:
            pos = torch.from_numpy(synth.mrope_position_ids(5 + c * gpc, gpc, gh, gw, hw0=5))
        else:
            pos = (torch.arange(L, dtype=torch.int64) + 5 + c * L)[None]
        # --- what the attention patch does before calling update (qwen2_vl.py:68-79 / llava_onevision.py:78-100)
        if reforge:
            prev = cache.get_prev_temporal_idx(layer)
            cur = pos[0, 0, 0] if mrope else pos[0, 0]
            if prev + 1 != cur:
                pos = pos.clone()
                if mrope:
                    pos[0, 0, :] += prev + 1 - cur
                else:
                    pos[0, :] += prev + 1 - cur
        q = synth.rope_forward(q0, pos, rotary, mrope)
        k = synth.rope_forward(k0, pos, rotary, mrope)
        mask = torch.from_numpy(rng.uniform(size=L) < mrate) if mrate > 0 else None
        cache.keypatches_mask_chunk = mask
        cache.kvcache_compression = True
        kw = {"sin": None, "cos": None, "cache_position": None, "query_states": q, "position_ids": pos.clone(),
              "rotary_emb": rotary}
        if mrope:
            kw["mrope_section"] = list(mrope)
        prev_len = 0 if len(cache.key_cache) <= layer or len(cache.key_cache[layer]) == 0 else cache.key_cache[layer].shape[2]
        torch.Tensor.topk = spy_topk
        try:
            kout, vout = cache.update(k, v, layer, kw)
        finally:
            torch.Tensor.topk = orig_topk
        assert set(kw.keys()) == {"sin", "cos", "cache_position"}, kw.keys()
        keep = max(1, int(ratio * L))
        assert kout.shape == (1, Hkv, prev_len + L, D)
        assert torch.equal(kout[:, :, prev_len:], k) and torch.equal(vout[:, :, prev_len:], v)
        kc = cache.key_cache[layer]
        vc_ = cache.value_cache[layer]
        assert kc.shape[2] == prev_len + keep
        kept_k = kc[:, :, prev_len:].numpy()
        kept_v = vc_[:, :, prev_len:].numpy()
        # recover keep indices from V rows (exact copies)
        vn = v.numpy()
        idx = np.empty(keep, dtype=np.int64)
        for r in range(keep):
            m = np.nonzero((vn[0, 0, :, :4] == kept_v[0, 0, r, :4]).all(axis=1))[0]
            m = [i for i in m if np.array_equal(vn[0, :, i], kept_v[0, :, r])]
            assert len(m) == 1
            idx[r] = m[0]
        assert (np.diff(idx) > 0).all()
        score32 = captured["score"].numpy()
        # fp64 decision variable: un-rotated (reforge) or rotated (no reforge) q,k
        if reforge:
            s64 = score_fp64(q0, k0, Hkv)
        else:
            s64 = score_fp64(q, k, Hkv)
        if mask is not None:
            s64 = s64.copy()
            s64[mask.numpy()] = 1.0
        srt = -np.sort(-s64)
        gap = srt[keep - 1] - srt[keep] if keep < L else np.inf
        if gap < FRAGILE and not (allow_tie and gap == 0.0):
            return None
        if allow_tie and gap != 0.0:
            return None  # the tie case must actually have its boundary inside the ties
        assert np.abs(score32 - s64).max() < 5e-5, np.abs(score32 - s64).max()
        min_gap = min(min_gap, gap)
        if mask is not None:
            masked_kept += int(mask.numpy()[idx].sum())
        pre = f"c{c}_"
        rec[pre + "pos"] = pos.numpy()
        rec[pre + "mask"] = mask.numpy() if mask is not None else np.zeros(0, dtype=bool)
        rec[pre + "score32"] = score32
        rec[pre + "score64"] = s64
        rec[pre + "keep_idx"] = idx
        rec[pre + "num_evicted"] = cache.num_evicted_tokens[layer]
        if reforge:
            pc = cache.position_cache[layer]
            rec[pre + "position_cache"] = pc.numpy().copy()
        if raw:
            rec[pre + "q"] = q.numpy()
            rec[pre + "k"] = k.numpy()
            rec[pre + "v"] = vn
            rec[pre + "kept_k"] = kept_k
            rec[pre + "kept_v"] = kept_v
        else:
            rec[pre + "q_crc"] = synth.checksum(q.numpy())
            rec[pre + "k_crc"] = synth.checksum(k.numpy())
            rec[pre + "v_crc"] = synth.checksum(vn)
            rec[pre + "kept_k"] = kept_k  # [1,Hkv,keep,D] fp32 — still small (<=1.2 MB)
            rec[pre + "kept_v_crc"] = synth.checksum(kept_v)
    rec["keep"] = keep
    rec["tie_case"] = allow_tie
    rec["min_topk_gap"] = min_gap
    rec["masked_kept"] = masked_kept
    rec["position_cache_len"] = len(cache.position_cache)
    rec["num_evicted_list"] = np.array(cache.num_evicted_tokens, dtype=np.int64)
    return rec


# --------------------------------------------------------------------------------------
# glue (G2-G5): pure index bookkeeping functions of retake/qwen2_vl.py
# --------------------------------------------------------------------------------------
def gen_glue(outdir):
    qv = import_reference_glue()
    VID, TXT = 151656, 7
    rec = {}
    # toy prompt: 5 text, 16 grids x 4 tokens video, 7 text
    grid_t, gh, gw = 16, 4, 4  # thw before merge (merge 2 -> 4 tokens per grid)
    n_vid = grid_t * gh * gw // 4
    ids = torch.tensor([[TXT] * 5 + [VID] * n_vid + [TXT] * 7])
    S = ids.shape[1]
    cfg = Cfg(video_token_id=VID,
              vision_config=Cfg(spatial_merge_size=2, temporal_patch_size=1),
              longvideo_kwargs={"chunked_prefill_frames": 8, "visual_compression": True,
                                "visual_compression_kwargs": {"compression_ratio": 0.5, "compression_method": "Keyframe",
                                                              "patch_sync": False, "return_keyframe_mask": True},
                                "kvcache_compression": True,
                                "kvcache_compression_kwargs": {"compression_ratio": 0.5, "compression_method": "pivotkv"}})
    me = Cfg(config=cfg)
    seg = qv.retake_Qwen2VLForConditionalGeneration_segment_input_ids(me, ids)
    rec["ids"] = ids.numpy()
    rec["seg_s"] = np.array([int(s) for s, e, t in seg])
    rec["seg_e"] = np.array([int(e) for s, e, t in seg])
    rec["seg_t"] = np.array([t for s, e, t in seg])
    thw = torch.tensor([[grid_t, gh, gw]])
    rec["thw"] = thw.numpy()
    rec["chunk_size"] = qv.retake_Qwen2VLForConditionalGeneration_get_chunk_size(me, cfg, thw)
    # compress_video_tokens
    emb = torch.from_numpy(synth.frames_video(77, grid_t, gh * gw // 4, 32))[0].reshape(-1, 32)
    pos = torch.arange(S)[None, None].repeat(3, 1, 1)
    am = torch.ones(1, S, dtype=torch.long)
    cp = torch.arange(S)
    out = qv.retake_Qwen2VLForConditionalGeneration_compress_video_tokens(
        me, input_ids=ids.clone(), attention_mask=am.clone(), video_embeds=emb.clone(), cache_position=cp.clone(),
        position_ids=pos.clone(), labels=None, video_grid_thw=thw)
    names = ["ids", "am", "emb", "cp", "pos", "labels", "mask"]
    rec["cvt_in_emb"] = emb.numpy()
    for n, o in zip(names, out):
        if o is not None:
            rec["cvt_" + n] = o.numpy()
    # forge_input_chunks
    ie = torch.arange(S * 2, dtype=torch.float32).reshape(1, S, 2)
    o = qv.retake_Qwen2VLForConditionalGeneration_forge_input_chunks(me, 9, 21, seg, cp, pos, am, None, ie)
    for n, t in zip(["cp", "pos", "am", "ie"], o[:4]):
        rec["fic_" + n] = t.numpy()
    rec["fic_prompt_length_is_none"] = o[4] is None
    # ids starting / ending with video, multiple segments
    ids2 = torch.tensor([[VID] * 3 + [TXT] * 2 + [VID] * 4])
    seg2 = qv.retake_Qwen2VLForConditionalGeneration_segment_input_ids(me, ids2)
    rec["ids2"] = ids2.numpy()
    rec["seg2_s"] = np.array([int(s) for s, e, t in seg2])
    rec["seg2_e"] = np.array([int(e) for s, e, t in seg2])
    rec["seg2_t"] = np.array([t for s, e, t in seg2])
    np.savez_compressed(os.path.join(outdir, "glue_qwen2vl.npz"), **rec)
    print("glue_qwen2vl: segments", seg, "chunk", rec["chunk_size"], "ids", ids.shape, "->", out[0].shape)



# --------------------------------------------------------------------------------------
# MA-LLM / MA-LLM-hard merges (visual_compression.py:5-83, driven as qwen2_vl.py:402-410 does)
# --------------------------------------------------------------------------------------
def gen_mallm(vc, outdir):
    import torch.nn.functional as F

    # (name, seed, T, N, C, tgt, sync, hard, dtype)
    cases = [
        ("mallm_soft_async_16x6x32_t9", 11, 16, 6, 32, 9, False, False, "fp32"),
        ("mallm_soft_sync_16x6x32_t9", 12, 16, 6, 32, 9, True, False, "fp32"),
        ("mallm_hard_async_16x6x32_t9", 13, 16, 6, 32, 9, False, True, "fp32"),
        ("mallm_hard_sync_16x6x32_t9", 14, 16, 6, 32, 9, True, True, "fp32"),
        ("mallm_soft_async_40x9x72_t22", 15, 40, 9, 72, 22, False, False, "fp32"),
        ("mallm_soft_async_12x5x64_t1", 16, 12, 5, 64, 1, False, False, "fp32"),
        ("mallm_soft_async_16x6x32_t11_bf16", 17, 16, 6, 32, 11, False, False, "bf16"),
        ("mallm_hard_async_16x6x32_t11_bf16", 18, 16, 6, 32, 11, False, True, "bf16"),
    ]
    for name, seed, T, N, C, tgt, sync, hard, dtype in cases:
        x = synth.frames_video(seed, T, N, C)
        xt = torch.from_numpy(x)
        if dtype == "bf16":
            xt = xt.bfloat16()
        bank = xt.clone()
        size = torch.ones_like(bank[:, :, :, 0])
        steps_idx, margins = [], []
        while bank.shape[1] > tgt:
            sim = F.cosine_similarity(bank[:, :-1, :], bank[:, 1:, :], dim=-1)        # what the reference computes
            sim64 = F.cosine_similarity(bank.double()[:, :-1, :], bank.double()[:, 1:, :], dim=-1)
            if sync:
                sim = sim.mean(-1, keepdim=True).expand(-1, -1, N)
                sim64 = sim64.mean(-1, keepdim=True).expand(-1, -1, N)
            steps_idx.append(torch.max(sim, dim=1).indices[0].numpy().copy())
            if sim64.shape[1] > 1:
                top2 = torch.topk(sim64[0], 2, dim=0).values
                margins.append(float((top2[0] - top2[1]).min()))
            if hard:
                bank = vc.memory_bank_compress_MALLM_hard(bank, sync=sync)
            else:
                bank, size = vc.memory_bank_compress_MALLM(bank, size, sync=sync)
        margin = min(margins) if margins else np.inf
        if dtype == "fp32":
            assert margin > FRAGILE, (name, margin)
        out = bank.float().numpy() if dtype == "fp32" else bank.view(torch.int16).numpy().view(np.uint16)
        rec = dict(x=x if dtype == "fp32" else xt.view(torch.int16).numpy().view(np.uint16), dtype=dtype, T=T, N=N, C=C,
                   tgt=tgt, sync=sync, hard=hard, out=out, steps_idx=np.stack(steps_idx).astype(np.int64), margin=margin,
                   size=(size.float().numpy() if not hard else np.zeros((0,), np.float32)))
        np.savez_compressed(os.path.join(outdir, name + ".npz"), **rec)
        print(name, "steps", len(steps_idx), "margin %.3g" % margin, "out", out.shape)


# --------------------------------------------------------------------------------------
# PivotKV in the production dtype: the reference run in bf16 (longvideo_cache.py:260-270 rounds the logits, the
# probabilities, the per-head column sums and both means to bf16)
# --------------------------------------------------------------------------------------
def bf16_bits(t: torch.Tensor) -> np.ndarray:
    return t.contiguous().view(torch.int16).numpy().view(np.uint16)


def gen_pivotkv_bf16(lc, outdir):
    S = synth.YARN_FACTOR4_ATTENTION_SCALING
    # name, gh, gw, grids per chunk, chunks, ratio, mask rate, seed, raw
    cases = [("bf16_qwen_L256", 8, 8, 4, 2, 0.25, 0.3, 211, True),
             ("bf16_qwen_L1568", 14, 14, 8, 1, 0.25, 0.3, 212, False),
             ("bf16_qwen_L6272", 14, 14, 32, 1, 0.25, 0.3, 213, False),
             ("bf16_qwen_L576_ratio1", 12, 12, 4, 2, 1, 0.3, 214, True)]   # dynamic ratio of a short prompt: keep all
    Hq, Hkv, D, mrope = 28, 4, 128, [16, 24, 24]
    for (name, gh, gw, gpc, nch, ratio, mrate, seed, raw) in cases:
        L = gpc * gh * gw
        inv_f = synth.inv_freq(D, 1e6)
        rotary = synth.RotaryStub(inv_f, S)
        cache = lc.PivotKVCache(make_config(Hq, Hkv, D, 1, ratio, True))
        rec = dict(Hq=Hq, Hkv=Hkv, D=D, L=L, gh=gh, gw=gw, grids_per_chunk=gpc, n_chunks=nch, ratio=ratio, reforge=True,
                   mrope_section=np.array(mrope, dtype=np.int64), attention_scaling=S, inv_freq=inv_f, seed=seed, layer=0,
                   raw=raw, theta=1e6, dtype="bf16")
        rng = np.random.default_rng(seed + 7)
        captured = {}
        orig_topk = torch.Tensor.topk

        def spy_topk(self, *args, **kw):
            captured["score"] = self.detach().clone()
            return orig_topk(self, *args, **kw)

        for c in range(nch):
            q0, k0, v = map(torch.from_numpy, synth.qkv_chunk(seed * 100 + c, Hq, Hkv, L, D))
            pos = torch.from_numpy(synth.mrope_position_ids(5 + c * gpc, gpc, gh, gw, hw0=5))
            prev = cache.get_prev_temporal_idx(0)
            if prev + 1 != pos[0, 0, 0]:
                pos = pos.clone()
                pos[0, 0, :] += prev + 1 - pos[0, 0, 0]
            q = synth.rope_forward(q0, pos, rotary, mrope).bfloat16()   # what a bf16 model hands to update
            k = synth.rope_forward(k0, pos, rotary, mrope).bfloat16()
            v = v.bfloat16()
            mask = torch.from_numpy(rng.uniform(size=L) < mrate)
            cache.keypatches_mask_chunk = mask
            cache.kvcache_compression = True
            kw = {"sin": None, "cos": None, "cache_position": None, "query_states": q, "position_ids": pos.clone(),
                  "rotary_emb": rotary, "mrope_section": list(mrope)}
            prev_len = 0 if not len(cache.key_cache) or len(cache.key_cache[0]) == 0 else cache.key_cache[0].shape[2]
            torch.Tensor.topk = spy_topk
            try:
                cache.update(k, v, 0, kw)
            finally:
                torch.Tensor.topk = orig_topk
            keep = max(1, int(ratio * L))
            kept_k = cache.key_cache[0][:, :, prev_len:]
            kept_v = cache.value_cache[0][:, :, prev_len:]
            vb, kvb = bf16_bits(v), bf16_bits(kept_v)
            idx = np.empty(keep, dtype=np.int64)        # kept rows are exact copies of V rows
            look = {}
            for i in range(L):
                look.setdefault(vb[0, 0, i, :8].tobytes(), []).append(i)
            for r in range(keep):
                m = [i for i in look[kvb[0, 0, r, :8].tobytes()] if np.array_equal(vb[0, :, i], kvb[0, :, r])]
                assert len(m) == 1
                idx[r] = m[0]
            assert (np.diff(idx) > 0).all()
            score_ref = captured["score"]                  # bf16 [L], after masked_fill_
            assert score_ref.dtype == torch.bfloat16
            # the reference's own un-rotated operands (its helper, its dtype) and the exact score they define
            cos, sin = rotary(v, pos)
            qt, kt = lc.apply_multimodal_rotary_pos_emb(q, k, cos, sin, mrope, reverse=True, attention_scaling=S)
            s64 = score_fp64(qt, kt, Hkv)
            pre = f"c{c}_"
            rec[pre + "pos"] = pos.numpy()
            rec[pre + "mask"] = mask.numpy()
            rec[pre + "score_bf16"] = bf16_bits(score_ref)
            rec[pre + "score64"] = s64                      # before the mask override
            rec[pre + "keep_idx"] = idx
            rec[pre + "kept_k_bits"] = bf16_bits(kept_k)
            rec[pre + "position_cache"] = cache.position_cache[0].numpy().copy()
            if raw:
                rec[pre + "q_bits"], rec[pre + "k_bits"], rec[pre + "v_bits"] = bf16_bits(q), bf16_bits(k), vb
            else:
                rec[pre + "q_crc"], rec[pre + "k_crc"] = synth.checksum(bf16_bits(q)), synth.checksum(bf16_bits(k))
                rec[pre + "v_crc"] = synth.checksum(vb)
            sr = score_ref.float().numpy()
            s64m = s64.copy()
            s64m[mask.numpy()] = 1.0
            exact = np.sort(np.lexsort((np.arange(L), -s64m))[:keep])
            thr = np.sort(sr)[::-1][keep - 1]
            print(f"pivotkv_{name} c{c}: L={L} keep={keep} distinct bf16 scores {len(np.unique(sr))}, kept-set overlap "
                  f"with exact scoring {np.intersect1d(idx, exact).size}/{keep}, ties at the threshold "
                  f"{int((sr == thr).sum())}, max |score_bf16 - exact| {np.abs(sr - s64m).max():.4f}")
        rec["keep"] = keep
        np.savez_compressed(os.path.join(outdir, f"pivotkv_{name}.npz"), **rec)


# --------------------------------------------------------------------------------------
# round 5: the reference's whole attention-side chain on a bf16 MODEL, from the bf16 PRE-RoPE projections: the rotary
# module's tables rounded to bf16 (`.to(x.dtype)`), the reference's own apply_multimodal_rotary_pos_emb /
# apply_rotary_pos_emb on bf16 tensors (longvideo_cache.py:35-116; what qwen2_vl.py:75-79 / llava_onevision.py:90-100
# call), then PivotKVCache.update (:217-323).  Pins PivotKVCache.update_pre_rope - the route the build's own attention
# patch takes - which is handed the same q0 / k0 and never sees the rotated tensors.
# --------------------------------------------------------------------------------------
def gen_pivotkv_prerope_bf16(lc, outdir, only=None):
    S = synth.YARN_FACTOR4_ATTENTION_SCALING
    M = [16, 24, 24]
    # name, gh, gw, grids per chunk, chunks, ratio, mask rate, seed, raw, mrope
    cases = [("prerope_bf16_qwen_L256", 8, 8, 4, 2, 0.25, 0.3, 231, True, M),
             ("prerope_bf16_qwen_L1568", 14, 14, 8, 2, 0.25, 0.3, 232, False, M),
             ("prerope_bf16_qwen_L6272", 14, 14, 32, 1, 0.25, 0.3, 233, False, M),
             # the REAL Qwen2-VL chunk (448 px, 16:9: 9 x 16 merged tokens per temporal grid, 16 grids = 2304 tokens,
             # cal_flops.py:8,47), two chunks: the geometry bench.py's `real_geometry` companion times, in the production dtype
             ("prerope_bf16_qwen_L2304", 9, 16, 16, 2, 0.25, 0.3, 237, False, M),
             ("prerope_bf16_llava_L1568", 14, 14, 8, 2, 0.25, 0.3, 234, False, None),
             # BASELINE configs[4]'s own setting: LLaVA-Video chunk (32 frames x 196 pooled tokens), plain RoPE, the dynamic
             # ratio of a 2048-frame prompt (max_input_length 40000 / 401409 tokens: keep 624 of 6272)
             ("prerope_bf16_llava_L6272_dyn", 14, 14, 32, 1, 40000 / (2048 * 196 + 1), 0.3, 235, False, None),
             # the same chain on a float16 model (the reference guards its eager attention for fp16, qwen2_vl.py:98-101)
             ("prerope_fp16_qwen_L1568", 14, 14, 8, 2, 0.25, 0.3, 236, False, M)]
    if only:
        cases = [c for c in cases if c[0] in only]
    Hq, Hkv, D = 28, 4, 128
    for (name, gh, gw, gpc, nch, ratio, mrate, seed, raw, mrope) in cases:
        tdt = torch.float16 if "_fp16_" in name else torch.bfloat16
        L = gpc * gh * gw
        inv_f = synth.inv_freq(D, 1e6)
        rotary = synth.RotaryStub(inv_f, S)
        cache = lc.PivotKVCache(make_config(Hq, Hkv, D, 1, ratio, True, llava=(mrope is None)))
        rec = dict(Hq=Hq, Hkv=Hkv, D=D, L=L, gh=gh, gw=gw, grids_per_chunk=gpc, n_chunks=nch, ratio=ratio, reforge=True,
                   mrope_section=np.array(mrope if mrope else [], dtype=np.int64), attention_scaling=S, inv_freq=inv_f,
                   seed=seed, layer=0, raw=raw, theta=1e6, dtype="fp16" if tdt is torch.float16 else "bf16")
        rng = np.random.default_rng(seed + 7)
        captured = {}
        orig_topk = torch.Tensor.topk

        def spy_topk(self, *args, **kw):
            captured["score"] = self.detach().clone()
            return orig_topk(self, *args, **kw)

        for c in range(nch):
            q0, k0, v = (torch.from_numpy(a).to(tdt) for a in synth.qkv_chunk(seed * 100 + c, Hq, Hkv, L, D))
            if mrope:
                pos_in = torch.from_numpy(synth.mrope_position_ids(5 + c * gpc, gpc, gh, gw, hw0=5))
            else:
                pos_in = (torch.arange(L, dtype=torch.int64) + 5 + c * L)[None]
            # the attention patch's continuity shift (qwen2_vl.py:68-73 / llava_onevision.py:78-88)
            pos = pos_in.clone()
            prev = cache.get_prev_temporal_idx(0)
            cur = pos[0, 0, 0] if mrope else pos[0, 0]
            if prev + 1 != cur:
                if mrope:
                    pos[0, 0, :] += prev + 1 - cur
                else:
                    pos[0, :] += prev + 1 - cur
            cos, sin = rotary(v, pos)                      # bf16 tables, like HF's rotary module on a bf16 model
            assert cos.dtype == tdt
            if mrope:
                q, k = lc.apply_multimodal_rotary_pos_emb(q0, k0, cos, sin, list(mrope))
            else:
                q, k = lc.apply_rotary_pos_emb(q0, k0, cos, sin)
            assert q.dtype == tdt
            mask = torch.from_numpy(rng.uniform(size=L) < mrate)
            cache.keypatches_mask_chunk = mask
            cache.kvcache_compression = True
            kw = {"sin": sin, "cos": cos, "cache_position": None, "query_states": q, "position_ids": pos.clone(),
                  "rotary_emb": rotary}
            if mrope:
                kw["mrope_section"] = list(mrope)
            prev_len = 0 if not len(cache.key_cache) or len(cache.key_cache[0]) == 0 else cache.key_cache[0].shape[2]
            torch.Tensor.topk = spy_topk
            try:
                cache.update(k, v, 0, kw)
            finally:
                torch.Tensor.topk = orig_topk
            keep = max(1, int(ratio * L))
            kept_k = cache.key_cache[0][:, :, prev_len:]
            kept_v = cache.value_cache[0][:, :, prev_len:]
            vb, kvb = bf16_bits(v), bf16_bits(kept_v)
            idx = np.empty(keep, dtype=np.int64)        # kept rows are exact copies of V rows
            look = {}
            for i in range(L):
                look.setdefault(vb[0, 0, i, :8].tobytes(), []).append(i)
            for r in range(keep):
                m = [i for i in look[kvb[0, 0, r, :8].tobytes()] if np.array_equal(vb[0, :, i], kvb[0, :, r])]
                assert len(m) == 1
                idx[r] = m[0]
            assert (np.diff(idx) > 0).all()
            score_ref = captured["score"]                  # bf16 [L], after masked_fill_
            assert score_ref.dtype == tdt
            s64_pre = score_fp64(q0, k0, Hkv)              # the exact score of the PRE-RoPE operands (what the prologue scores)
            # ... and of the reference's own round-tripped operands (its helper, its dtype)
            if mrope:
                qt, kt = lc.apply_multimodal_rotary_pos_emb(q, k, cos, sin, list(mrope), reverse=True, attention_scaling=S)
            else:
                qt, kt = lc.apply_rotary_pos_emb(q, k, cos, sin, reverse=True, attention_scaling=S)
            s64_rt = score_fp64(qt, kt, Hkv)
            pre = f"c{c}_"
            rec[pre + "pos_in"] = pos_in.numpy()            # what the caller hands over ...
            rec[pre + "pos"] = pos.numpy()                  # ... and what the reference rotates with
            rec[pre + "mask"] = mask.numpy()
            rec[pre + "score_bf16"] = bf16_bits(score_ref)
            rec[pre + "score64_pre"] = s64_pre              # both before the mask override
            rec[pre + "score64"] = s64_rt
            rec[pre + "keep_idx"] = idx
            rec[pre + "kept_k_bits"] = bf16_bits(kept_k)
            rec[pre + "position_cache"] = cache.position_cache[0].numpy().copy()
            rec[pre + "q_rot_crc"], rec[pre + "k_rot_crc"] = synth.checksum(bf16_bits(q)), synth.checksum(bf16_bits(k))
            if raw:
                rec[pre + "q0_bits"], rec[pre + "k0_bits"], rec[pre + "v_bits"] = bf16_bits(q0), bf16_bits(k0), vb
            else:
                rec[pre + "q0_crc"], rec[pre + "k0_crc"] = synth.checksum(bf16_bits(q0)), synth.checksum(bf16_bits(k0))
                rec[pre + "v_crc"] = synth.checksum(vb)
            sr = score_ref.float().numpy()
            m_ = mask.numpy()
            s64m = s64_pre.copy()
            s64m[m_] = 1.0
            exact = np.sort(np.lexsort((np.arange(L), -s64m))[:keep])
            thr = np.sort(sr)[::-1][keep - 1]
            print(f"pivotkv_{name} c{c}: L={L} keep={keep} kept-set overlap with exact pre-RoPE scoring "
                  f"{np.intersect1d(idx, exact).size}/{keep}, ties at the threshold {int((sr == thr).sum())}, "
                  f"max |score_bf16 - exact pre| {np.abs(sr - s64m).max():.4f}, "
                  f"max |exact round-trip - exact pre| {np.abs(s64_rt - s64_pre).max():.2e}")
        rec["keep"] = keep
        np.savez_compressed(os.path.join(outdir, f"pivotkv_{name}.npz"), **rec)


# --------------------------------------------------------------------------------------
# fp16 (round 3): the reference run on float16 tensors - DPSelect and PivotKV.  Same recording as the bf16 fixtures;
# 16-bit payloads are stored as their bit patterns.
# --------------------------------------------------------------------------------------
def gen_fp16(vc, lc, outdir):
    # ---- DPSelect
    for (name, seed, T, N, C, tgt, sync, raw) in [("video64x16x64_async_r50_fp16", 51, 64, 16, 64, 32, False, True),
                                                  ("video64x16x64_sync_r50_fp16", 52, 64, 16, 64, 32, True, True),
                                                  ("video64x196x1280_async_r100_fp16", 53, 64, 196, 1280, 64, False, False),
                                                  ("video64x196x1280_async_r50_fp16", 54, 64, 196, 1280, 32, False, False),
                                                  ("video32x144x3584_async_r100_fp16", 55, 32, 144, 3584, 32, False, False)]:
        x = torch.from_numpy(synth.make_frames("video", seed, T, N, C)).half()
        d32, d64 = dis_matrices(x)
        out, mask = vc.memory_bank_compress_keyframe(x.clone(), tgt, 3, sync=sync)
        xin, outn = x.float().numpy(), out.float().numpy()
        if sync:
            idx = recover_frame_idx(xin[:, :, :1], outn[:, :, :1])[:, 0]
            rows64 = d64.mean(1, keepdims=True).T
        else:
            idx = recover_frame_idx(xin, outn)
            rows64 = d64.T
        pm = peak_margins(rows64)
        pk = (rows64 > np.concatenate([np.full((rows64.shape[0], 1), -np.inf), rows64[:, :-1]], 1)) & \
             (rows64 >= np.concatenate([rows64[:, 1:], np.full((rows64.shape[0], 1), -np.inf)], 1))
        tm = topk_margin(rows64 + 2.0 * pk, tgt)
        rec = dict(kind="video", seed=seed, T=T, N=N, C=C, tgt=tgt, sync=sync, window=3, dtype="fp16",
                   x_crc=synth.checksum(x.view(torch.int16).numpy()), idx=idx, mask=mask.numpy(), dis32=d32, dis64=d64,
                   out_crc=synth.checksum(out.view(torch.int16).numpy()), min_peak_gap=pm, min_topk_gap=tm)
        if raw:
            rec["x"] = x.view(torch.int16).numpy()
        np.savez_compressed(os.path.join(outdir, f"dpselect_{name}.npz"), **rec)
        print(f"dpselect_{name}: t={tgt} sync={sync} mask_rate={mask.float().mean():.3f}")

    # ---- PivotKV
    S = synth.YARN_FACTOR4_ATTENTION_SCALING
    Hq, Hkv, D, mrope = 28, 4, 128, [16, 24, 24]
    for (name, gh, gw, gpc, ratio, mrate, seed, raw) in [("fp16_qwen_L256", 8, 8, 4, 0.25, 0.3, 221, True),
                                                         ("fp16_qwen_L1568", 14, 14, 8, 0.25, 0.3, 222, False)]:
        L = gpc * gh * gw
        inv_f = synth.inv_freq(D, 1e6)
        rotary = synth.RotaryStub(inv_f, S)
        cache = lc.PivotKVCache(make_config(Hq, Hkv, D, 1, ratio, True))
        rec = dict(Hq=Hq, Hkv=Hkv, D=D, L=L, gh=gh, gw=gw, grids_per_chunk=gpc, n_chunks=1, ratio=ratio, reforge=True,
                   mrope_section=np.array(mrope, dtype=np.int64), attention_scaling=S, inv_freq=inv_f, seed=seed, layer=0,
                   raw=raw, theta=1e6, dtype="fp16")
        rng = np.random.default_rng(seed + 7)
        captured = {}
        orig_topk = torch.Tensor.topk

        def spy_topk(self, *args, **kw):
            captured["score"] = self.detach().clone()
            return orig_topk(self, *args, **kw)

        q0, k0, v = map(torch.from_numpy, synth.qkv_chunk(seed * 100, Hq, Hkv, L, D))
        pos = torch.from_numpy(synth.mrope_position_ids(5, gpc, gh, gw, hw0=5))
        q = synth.rope_forward(q0, pos, rotary, mrope).half()   # what a float16 model hands to update
        k = synth.rope_forward(k0, pos, rotary, mrope).half()
        v = v.half()
        mask = torch.from_numpy(rng.uniform(size=L) < mrate)
        cache.keypatches_mask_chunk = mask
        cache.kvcache_compression = True
        kw = {"sin": None, "cos": None, "cache_position": None, "query_states": q, "position_ids": pos.clone(),
              "rotary_emb": rotary, "mrope_section": list(mrope)}
        torch.Tensor.topk = spy_topk
        try:
            cache.update(k, v, 0, kw)
        finally:
            torch.Tensor.topk = orig_topk
        keep = max(1, int(ratio * L))
        kept_k, kept_v = cache.key_cache[0], cache.value_cache[0]
        vb, kvb = bf16_bits(v), bf16_bits(kept_v)           # (bit patterns of any 16-bit dtype)
        idx = np.empty(keep, dtype=np.int64)
        look = {}
        for i in range(L):
            look.setdefault(vb[0, 0, i, :8].tobytes(), []).append(i)
        for r in range(keep):
            m = [i for i in look[kvb[0, 0, r, :8].tobytes()] if np.array_equal(vb[0, :, i], kvb[0, :, r])]
            assert len(m) == 1
            idx[r] = m[0]
        assert (np.diff(idx) > 0).all()
        score_ref = captured["score"]
        assert score_ref.dtype == torch.float16
        cos, sin = rotary(v, pos)
        qt, kt = lc.apply_multimodal_rotary_pos_emb(q, k, cos, sin, mrope, reverse=True, attention_scaling=S)
        s64 = score_fp64(qt, kt, Hkv)
        rec["c0_pos"], rec["c0_mask"] = pos.numpy(), mask.numpy()
        rec["c0_score_bits"] = bf16_bits(score_ref)
        rec["c0_score64"] = s64
        rec["c0_keep_idx"] = idx
        rec["c0_kept_k_bits"] = bf16_bits(kept_k)
        rec["c0_k_unrot_bits"] = bf16_bits(kt)              # the reference's own un-rotated keys (its helper, fp16)
        rec["c0_position_cache"] = cache.position_cache[0].numpy().copy()
        if raw:
            rec["c0_q_bits"], rec["c0_k_bits"], rec["c0_v_bits"] = bf16_bits(q), bf16_bits(k), vb
        else:
            rec["c0_q_crc"], rec["c0_k_crc"] = synth.checksum(bf16_bits(q)), synth.checksum(bf16_bits(k))
            rec["c0_v_crc"] = synth.checksum(vb)
        rec["keep"] = keep
        sr = score_ref.float().numpy()
        s64m = s64.copy()
        s64m[mask.numpy()] = 1.0
        exact = np.sort(np.lexsort((np.arange(L), -s64m))[:keep])
        print(f"pivotkv_{name}: L={L} keep={keep} distinct fp16 scores {len(np.unique(sr))}, kept-set overlap with exact "
              f"scoring {np.intersect1d(idx, exact).size}/{keep}, max |score_fp16 - exact| {np.abs(sr - s64m).max():.5f}")
        np.savez_compressed(os.path.join(outdir, f"pivotkv_{name}.npz"), **rec)

    # ---- MA-LLM / MA-LLM-hard on a float16 bank (the merge loop of qwen2_vl.py:402-410)
    for (name, seed, T, N, C, tgt, sync, hard) in [("fp16mallm_soft_async_16x6x32_t11", 27, 16, 6, 32, 11, False, False),
                                                   ("fp16mallm_hard_sync_16x6x32_t11", 28, 16, 6, 32, 11, True, True)]:
        xt = torch.from_numpy(synth.frames_video(seed, T, N, C)).half()
        bank = xt.clone()
        size = torch.ones_like(bank[:, :, :, 0])
        while bank.shape[1] > tgt:
            if hard:
                bank = vc.memory_bank_compress_MALLM_hard(bank, sync=sync)
            else:
                bank, size = vc.memory_bank_compress_MALLM(bank, size, sync=sync)
        np.savez_compressed(os.path.join(outdir, name + ".npz"), x=xt.view(torch.int16).numpy(), dtype="fp16", T=T, N=N, C=C,
                            tgt=tgt, sync=sync, hard=hard, out=bank.view(torch.int16).numpy(),
                            size=(size.float().numpy() if not hard else np.zeros((0,), np.float32)))
        print(name, "out", tuple(bank.shape))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    args = ap.parse_args()
    vc, lc = import_reference()
    if args.only in (None, "dpselect"):
        gen_dpselect(vc, HERE)
    if args.only in (None, "dpselect_degenerate"):
        gen_dpselect_degenerate(vc, HERE)
    if args.only in (None, "pivotkv"):
        gen_pivotkv(lc, HERE)
    if args.only in (None, "pivotkv_bf16"):
        gen_pivotkv_bf16(lc, HERE)
    if args.only == "pivotkv_r5":     # only the fp32 cases round 5 added (the others are unchanged)
        gen_pivotkv(lc, HERE, only=("qwen_L1568", "llava_L1568", "qwen_L6272"))
    if args.only in (None, "pivotkv_prerope_bf16"):
        gen_pivotkv_prerope_bf16(lc, HERE)
    if args.only == "pivotkv_prerope_llava_dyn":
        gen_pivotkv_prerope_bf16(lc, HERE, only=("prerope_bf16_llava_L6272_dyn",))
    if args.only == "pivotkv_prerope_qwen_L2304":
        gen_pivotkv_prerope_bf16(lc, HERE, only=("prerope_bf16_qwen_L2304",))
    if args.only == "pivotkv_prerope_fp16":
        gen_pivotkv_prerope_bf16(lc, HERE, only=("prerope_fp16_qwen_L1568",))
    if args.only in (None, "glue"):
        gen_glue(HERE)
    if args.only in (None, "mallm"):
        gen_mallm(vc, HERE)
    if args.only in (None, "fp16"):
        gen_fp16(vc, lc, HERE)


if __name__ == "__main__":
    main()
