"""numpy/ctypes front-end of the CPU oracle (oracle/retake_oracle.c).

TEST INFRASTRUCTURE ONLY — see the header of retake_oracle.c.  Imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg; never by the product package.

The functions mirror the reference entry points in numpy terms:
  dpselect(...)            <-> retake/visual_compression.py:86-177  memory_bank_compress_keyframe
  OraclePivotKV.update(...) <-> retake/longvideo_cache.py:217-323   PivotKVCache.update
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libretake_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "retake_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        for name in ("orc_dpselect_dis_f32", "orc_dpselect_dis_bf16", "orc_topk_sorted", "orc_dpselect_select",
                     "orc_gather_frames", "orc_mrope_merge", "orc_rope_apply", "orc_pivotkv_score",
                     "orc_pivotkv_select", "orc_gather_rows", "orc_pivotkv_positions", "orc_num_threads", "orc_mallm_step",
                     "orc_rope_apply_bf16", "orc_pivotkv_score_bf16", "orc_pivotkv_score_fp16", "orc_round_fp16"):
            getattr(_lib, name).restype = C.c_int
        _lib.orc_rope_apply.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                        C.c_double, C.c_void_p]
        _lib.orc_rope_apply_bf16.argtypes = _lib.orc_rope_apply.argtypes
    return _lib


def num_threads() -> int:
    return lib().orc_num_threads()


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def _chk(rc: int, what: str):
    if rc != 0:
        raise RuntimeError(f"oracle {what} failed rc={rc}")


# ---------------------------------------------------------------------------------------------
# MA-LLM merges
# ---------------------------------------------------------------------------------------------
def mallm_step(x: np.ndarray, sizes, sync: bool, hard: bool):
    """One merge step (visual_compression.py:5-83).  x [T,N,C] float32 or uint16 (bf16 bits); sizes [T,N] same
    dtype (ignored for hard).  Returns (out [T-1,N,C], sizes_out [T-1,N] or None, idx [N])."""
    x = np.ascontiguousarray(x)
    T, N, Cc = x.shape
    is_bf16 = x.dtype == np.uint16
    out = np.empty((T - 1, N, Cc), dtype=x.dtype)
    so = None if hard else np.empty((T - 1, N), dtype=x.dtype)
    idx = np.empty((N,), dtype=np.int64)
    sz = None if hard else np.ascontiguousarray(sizes, dtype=x.dtype)
    _chk(lib().orc_mallm_step(_p(x), None if hard else _p(sz), int(is_bf16), T, N, Cc, int(bool(sync)), int(bool(hard)),
                              _p(out), None if hard else _p(so), _p(idx)), "mallm_step")
    return out, so, idx


def mallm_compress(x: np.ndarray, tgt: int, sync: bool, hard: bool):
    """The loop of qwen2_vl.py:402-410: merge until tgt frames are left.  Returns (bank, sizes, idx per step)."""
    x = np.ascontiguousarray(x)
    one = np.uint16(0x3F80) if x.dtype == np.uint16 else np.float32(1.0)
    sizes = np.full(x.shape[:2], one, dtype=x.dtype)
    steps = []
    while x.shape[0] > tgt:
        x, sizes_new, idx = mallm_step(x, sizes, sync, hard)
        sizes = sizes_new if sizes_new is not None else sizes[:-1]
        steps.append(idx)
    return x, (None if hard else sizes), np.stack(steps) if steps else np.zeros((0, x.shape[1]), np.int64)


# ---------------------------------------------------------------------------------------------
# DPSelect
# ---------------------------------------------------------------------------------------------
def dpselect_dis_f16(x16: np.ndarray) -> np.ndarray:
    """x [T,N,C] numpy float16 -> dis [T,N] float32: visual_compression.py:100-106 on a float16 tensor.  ATen's
    cosine_similarity is a composite of tensor ops, each rounding to the tensor dtype with fp32 arithmetic inside
    (verified against the imported reference: equal except for the last bit of ~0.1-0.4 % of the sums, the fp32
    summation order):  norm = fp16(sqrt(sum x^2)), clamp_min(fp16(1e-8) = 0), xn = fp16(x / norm),
    cos = fp16(sum fp16(xn_t * xn_t+1)),  dis = 1 - float(cos), row 0 = 1.  numpy's float16 arithmetic rounds every op
    like torch's CPU half kernels (fp32 operation, then one rounding to half)."""
    xf = np.ascontiguousarray(x16, dtype=np.float16).astype(np.float32)
    nrm = np.sqrt((xf * xf).sum(-1, dtype=np.float32)).astype(np.float16)
    nrm = np.maximum(nrm, np.float16(1e-8))
    with np.errstate(divide="ignore", invalid="ignore"):
        xn = (xf / nrm.astype(np.float32)[..., None]).astype(np.float16)
    prod = (xn[:-1].astype(np.float32) * xn[1:].astype(np.float32)).astype(np.float16)
    cos = prod.astype(np.float32).sum(-1, dtype=np.float32).astype(np.float16)
    dis = np.empty(xf.shape[:2], dtype=np.float32)
    dis[0] = 1.0
    dis[1:] = np.float32(1.0) - cos.astype(np.float32)
    return dis


def dpselect_dis(x: np.ndarray) -> np.ndarray:
    """x [T,N,C] float32, uint16 holding bf16 bits, or numpy float16 -> dis [T,N] float32."""
    x = np.ascontiguousarray(x)
    if x.dtype == np.float16:
        return dpselect_dis_f16(x)
    T, N, Cc = x.shape
    dis = np.empty((T, N), dtype=np.float32)
    if x.dtype == np.float32:
        _chk(lib().orc_dpselect_dis_f32(_p(x), T, N, Cc, _p(dis)), "dis_f32")
    elif x.dtype == np.uint16:
        _chk(lib().orc_dpselect_dis_bf16(_p(x), T, N, Cc, _p(dis)), "dis_bf16")
    else:
        raise TypeError(x.dtype)
    return dis


def dpselect_select(dis: np.ndarray, tgt: int, window: int = 3, sync: bool = True):
    dis = np.ascontiguousarray(dis, dtype=np.float32)
    T, N = dis.shape
    idx = np.empty((tgt,) if sync else (tgt, N), dtype=np.int64)
    mask = np.empty((tgt, N), dtype=np.uint8)
    keys = np.empty((T,) if sync else (N, T), dtype=np.float32)
    _chk(lib().orc_dpselect_select(_p(dis), T, N, tgt, window, int(sync), _p(idx), _p(mask), _p(keys)), "select")
    return idx, mask.astype(bool), keys


def gather_frames(x: np.ndarray, idx: np.ndarray, sync: bool) -> np.ndarray:
    x = np.ascontiguousarray(x)
    T, N, Cc = x.shape
    t = idx.shape[0]
    out = np.empty((t, N, Cc), dtype=x.dtype)
    idx = np.ascontiguousarray(idx, dtype=np.int64)
    _chk(lib().orc_gather_frames(_p(x), x.dtype.itemsize, T, N, Cc, _p(idx), t, int(sync), _p(out)), "gather")
    return out


def dpselect(memory_bank: np.ndarray, tgt_mem_len: int, window_size: int = 3, sync: bool = True):
    """memory_bank [1,T,N,C] -> (compressed [1,t,N,C], mask_flat [t*N] bool, idx, dis [T,N])."""
    assert memory_bank.shape[0] == 1
    if not sync and memory_bank.shape[2] == 1:
        # visual_compression.py:153 `.squeeze()` drops the patch axis -> IndexError at :156
        raise IndexError("DPSelect async mode with N == 1 is a reference crash")
    x = memory_bank[0]
    dis = dpselect_dis(x)
    idx, mask, _ = dpselect_select(dis, tgt_mem_len, window_size, sync)
    out = gather_frames(x, idx, sync)
    return out[None], mask.reshape(-1), idx, dis


# ---------------------------------------------------------------------------------------------
# PivotKV
# ---------------------------------------------------------------------------------------------
def mrope_merge(cs3: np.ndarray, sections) -> np.ndarray:
    """cos or sin [3,L,D] -> merged [L,D] (longvideo_cache.py:68-74)."""
    cs3 = np.ascontiguousarray(cs3, dtype=np.float32)
    _, L, D = cs3.shape
    sec = np.asarray(list(sections), dtype=np.int32)
    out = np.empty((L, D), dtype=np.float32)
    _chk(lib().orc_mrope_merge(_p(cs3), L, D, _p(sec), len(sec), _p(out)), "mrope_merge")
    return out


def rope_apply(x: np.ndarray, cos: np.ndarray, sin: np.ndarray, reverse: bool, attention_scaling: float = 1.0,
               bf16: bool = False):
    """x [H,L,D], cos/sin [L,D] -> rotated / un-rotated copy (longvideo_cache.py:76-81).  bf16: the arrays hold bf16
    values and every torch op of the formula rounds to bf16 (the reference on a bf16 model)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    H, L, D = x.shape
    cos = np.ascontiguousarray(cos, dtype=np.float32)
    sin = np.ascontiguousarray(sin, dtype=np.float32)
    out = np.empty_like(x)
    fn = lib().orc_rope_apply_bf16 if bf16 else lib().orc_rope_apply
    _chk(fn(_p(x), H, L, D, _p(cos), _p(sin), int(reverse), float(attention_scaling), _p(out)), "rope_apply")
    return out


def rope_apply_f16(x: np.ndarray, cos: np.ndarray, sin: np.ndarray, reverse: bool, attention_scaling: float = 1.0):
    """rope_apply on a float16 model: x [H,L,D], cos/sin [L,D] fp32 arrays holding fp16 values; every torch op of
    longvideo_cache.py:76-81 rounds to fp16 (numpy float16 arithmetic = fp32 operation + one rounding, like torch's CPU
    half kernels); the division by attention_scaling**2 is tensor / python scalar: fp32 quotient, one rounding."""
    x16, c16, s16 = (np.ascontiguousarray(a, dtype=np.float32).astype(np.float16) for a in (x, cos, sin))
    h2 = x16.shape[-1] // 2
    rot = np.concatenate([-x16[..., h2:], x16[..., :h2]], axis=-1)
    if reverse:
        t = (x16 * c16[None]) - (rot * s16[None])
        a2 = np.float32(float(attention_scaling) ** 2)
        out = (t.astype(np.float32) / a2).astype(np.float16) if a2 != 1.0 else t
    else:
        out = (x16 * c16[None]) + (rot * s16[None])
    return out.astype(np.float32)


def bf16_round(a: np.ndarray) -> np.ndarray:
    """fp32 -> nearest bf16 value (ties to even), returned as fp32."""
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def bf16_bits_to_f32(b: np.ndarray) -> np.ndarray:
    return (np.ascontiguousarray(b).astype(np.uint32) << 16).view(np.float32)


def pivotkv_score_bf16(q: np.ndarray, k: np.ndarray) -> np.ndarray:
    """The score in the reference's bf16 semantics (longvideo_cache.py:264-270 on bf16 tensors): q [Hq,L,D], k
    [Hkv,L,D] fp32 arrays of bf16 values -> score [L] fp32 array of bf16 values."""
    q = np.ascontiguousarray(q, dtype=np.float32)
    k = np.ascontiguousarray(k, dtype=np.float32)
    Hq, L, D = q.shape
    score = np.empty(L, dtype=np.float32)
    _chk(lib().orc_pivotkv_score_bf16(_p(q), _p(k), Hq, k.shape[0], L, D, _p(score)), "score_bf16")
    return score


def pivotkv_score_fp16(q: np.ndarray, k: np.ndarray) -> np.ndarray:
    """The same chain on a float16 model (every rounding to fp16): fp32 arrays of fp16 values in, fp16 values out."""
    q = np.ascontiguousarray(q, dtype=np.float32)
    k = np.ascontiguousarray(k, dtype=np.float32)
    Hq, L, D = q.shape
    score = np.empty(L, dtype=np.float32)
    _chk(lib().orc_pivotkv_score_fp16(_p(q), _p(k), Hq, k.shape[0], L, D, _p(score)), "score_fp16")
    return score


def round_fp16(x: np.ndarray) -> np.ndarray:
    """The C oracle's fp16 rounding (gcc 11 has no _Float16 on x86-64): checked against numpy.float16 by the tests."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty_like(x)
    _chk(lib().orc_round_fp16(_p(x), C.c_long(x.size), _p(out)), "round_fp16")
    return out


def pivotkv_score(q: np.ndarray, k: np.ndarray) -> np.ndarray:
    """q [Hq,L,D], k [Hkv,L,D] fp32 -> score [L] (longvideo_cache.py:260-270)."""
    q = np.ascontiguousarray(q, dtype=np.float32)
    k = np.ascontiguousarray(k, dtype=np.float32)
    Hq, L, D = q.shape
    Hkv = k.shape[0]
    score = np.empty(L, dtype=np.float32)
    _chk(lib().orc_pivotkv_score(_p(q), _p(k), Hq, Hkv, L, D, _p(score)), "score")
    return score


def pivotkv_select(score: np.ndarray, mask, keep: int) -> np.ndarray:
    """In-place mask override + top-k -> ascending int64 indices (longvideo_cache.py:272-277)."""
    assert score.dtype == np.float32 and score.flags.c_contiguous
    L = score.shape[0]
    idx = np.empty(keep, dtype=np.int64)
    m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
    _chk(lib().orc_pivotkv_select(_p(score), _p(m) if m is not None else None, L, keep, _p(idx)), "pk_select")
    return idx


def gather_rows(x: np.ndarray, idx: np.ndarray) -> np.ndarray:
    x = np.ascontiguousarray(x)
    H, L, D = x.shape
    out = np.empty((H, len(idx), D), dtype=x.dtype)
    idx = np.ascontiguousarray(idx, dtype=np.int64)
    _chk(lib().orc_gather_rows(_p(x), x.dtype.itemsize, H, L, D, _p(idx), len(idx), _p(out)), "gather_rows")
    return out


def pivotkv_positions(pos: np.ndarray, idx: np.ndarray, reforge: bool) -> np.ndarray:
    """pos [P,L] int64 -> kept (and temporally rescaled) ids [P,keep] (longvideo_cache.py:283-295)."""
    pos = np.ascontiguousarray(pos, dtype=np.int64)
    P, L = pos.shape
    idx = np.ascontiguousarray(idx, dtype=np.int64)
    out = np.empty((P, len(idx)), dtype=np.int64)
    _chk(lib().orc_pivotkv_positions(_p(pos), P, L, _p(idx), len(idx), int(reforge), _p(out)), "positions")
    return out


class OraclePivotKV:
    """numpy restatement of PivotKVCache (longvideo_cache.py:119-323) for bsz == 1.

    `rotary` is the same kind of callable the reference receives in cache_kwargs['rotary_emb']:
    (x_torch, position_ids_torch) -> (cos, sin), with `.attention_scaling`.
    """

    def __init__(self, num_heads: int, num_kv_heads: int, head_dim: int, compression_ratio: float,
                 pos_embed_reforge: bool = False, bf16: bool = False, score_rounding: str = "reference",
                 fp16: bool = False):
        """bf16: inputs are fp32 arrays of bf16 values and the RoPE steps round per torch op like the reference on a
        bf16 model; score_rounding 'reference' = the reference's bf16 score chain, 'fp32' = exact products with fp32
        accumulation and fp32 softmax / sums (what the HIP default computes for bf16 inputs)."""
        self.bf16, self.score_rounding = bf16, score_rounding
        self.fp16 = fp16   # inputs are fp32 arrays of fp16 values, RoPE steps round to fp16; the score is exact unless
                           # score_rounding == "reference16" (the reference's fp16 chain; the default "reference" names the bf16 one)
        assert not (bf16 and fp16)
        self.Hq, self.Hkv, self.D = num_heads, num_kv_heads, head_dim
        self.compression_ratio = compression_ratio
        self.pos_embed_reforge = pos_embed_reforge
        self.kvcache_compression = True
        self.keypatches_mask_chunk = None
        self.key_cache, self.value_cache = [], []
        self.position_cache, self.num_evicted_tokens = [], []
        self.last = {}

    # longvideo_cache.py:152-209
    def _upd_evicted(self, n, layer):
        if len(self.num_evicted_tokens) <= layer:
            self.num_evicted_tokens += [0] * (layer - len(self.num_evicted_tokens)) + [n]
        else:
            self.num_evicted_tokens[layer] += n

    def _upd_pos(self, pos, layer):
        if len(self.position_cache) <= layer:
            self.position_cache += [[] for _ in range(layer - len(self.position_cache))] + [pos]
        elif len(self.position_cache[layer]) == 0:
            self.position_cache[layer] = pos
        else:
            self.position_cache[layer] = np.concatenate([self.position_cache[layer], pos], axis=-1)

    def get_prev_temporal_idx(self, layer):  # :211-215
        if len(self.position_cache) <= layer:
            return -1
        pc = self.position_cache[layer]
        return int(pc[0, 0, -1] if pc.ndim == 3 else pc[0, -1])

    def _tables(self, rotary, x, pos, mrope_section):
        import torch

        xt = torch.from_numpy(x)
        xt = xt.bfloat16() if self.bf16 else (xt.half() if self.fp16 else xt)
        cos, sin = rotary(xt, torch.from_numpy(pos))   # tables in the model dtype
        cos, sin = cos.float().numpy(), sin.float().numpy()
        if mrope_section:
            return mrope_merge(cos[:, 0], mrope_section), mrope_merge(sin[:, 0], mrope_section)
        return np.ascontiguousarray(cos[0]), np.ascontiguousarray(sin[0])

    def update(self, k, v, layer, q=None, position_ids=None, rotary=None, mrope_section=None):
        """k,v [1,Hkv,L,D], q [1,Hq,L,D], position_ids [3,1,L] or [1,L] -> returned (K,V) uncompressed."""
        # 1) base append (:238)
        if len(self.key_cache) <= layer:
            for _ in range(len(self.key_cache), layer):
                self.key_cache.append([])
                self.value_cache.append([])
            self.key_cache.append(k)
            self.value_cache.append(v)
        elif len(self.key_cache[layer]) == 0:
            self.key_cache[layer], self.value_cache[layer] = k, v
        else:
            self.key_cache[layer] = np.concatenate([self.key_cache[layer], k], axis=2)
            self.value_cache[layer] = np.concatenate([self.value_cache[layer], v], axis=2)
        k_out, v_out = self.key_cache[layer], self.value_cache[layer]
        if not self.kvcache_compression:  # :319-321
            if self.pos_embed_reforge:
                self._upd_pos(position_ids, layer)
            return k_out, v_out
        assert q.shape[0] == 1
        L = q.shape[2]
        k_len = k.shape[2]
        qs, ks = q[0], k[0]
        if self.pos_embed_reforge:  # :248-259
            cos, sin = self._tables(rotary, v, position_ids, mrope_section)
            a = rotary.attention_scaling
            if self.fp16:
                qs, ks = rope_apply_f16(qs, cos, sin, True, a), rope_apply_f16(ks, cos, sin, True, a)
            else:
                qs = rope_apply(qs, cos, sin, True, a, bf16=self.bf16)
                ks = rope_apply(ks, cos, sin, True, a, bf16=self.bf16)
        keep = max(1, int(self.compression_ratio * L))  # :263
        if self.bf16 and self.score_rounding == "reference":
            score = pivotkv_score_bf16(qs, ks)
        elif self.fp16 and self.score_rounding == "reference16":
            score = pivotkv_score_fp16(qs, ks)
        else:
            score = pivotkv_score(qs, ks)  # :264-270
        idx = pivotkv_select(score, self.keypatches_mask_chunk, keep)  # :272-277
        kk = gather_rows(ks, idx)  # :278-280
        vv = gather_rows(v[0], idx)
        P = position_ids.shape[0] if position_ids.ndim == 3 else 1
        pos2 = position_ids.reshape(P, -1)
        newpos = pivotkv_positions(pos2, idx, self.pos_embed_reforge)  # :283-295
        newpos_t = newpos.reshape((3, 1, keep) if position_ids.ndim == 3 else (1, keep))
        if self.pos_embed_reforge:  # :297-306
            cos, sin = self._tables(rotary, vv[None], newpos_t, mrope_section)
            kk = rope_apply_f16(kk, cos, sin, False) if self.fp16 else rope_apply(kk, cos, sin, False, bf16=self.bf16)
            self._upd_pos(newpos_t, layer)  # :308-309
        self._upd_evicted(k_len - keep, layer)  # :310
        self.key_cache[layer] = np.concatenate([k_out[:, :, :-L], kk[None]], axis=2)  # :313-318
        self.value_cache[layer] = np.concatenate([v_out[:, :, :-L], vv[None]], axis=2)
        self.last = dict(score=score, keep_idx=idx, kept_k=kk[None], kept_v=vv[None], pos=newpos_t, k_unrot=ks)
        return k_out, v_out
