"""PivotKV cache on MI355X.  Same surface as the reference's retake/longvideo_cache.py.

`PivotKVCache.update` (reference :217-323) keeps its signature, its `cache_kwargs` protocol (pops
`position_ids`, `query_states`, `rotary_emb`, `mrope_section`) and its return value (the UNCOMPRESSED
keys/values of the layer); scoring, selection and the eviction scan run as HIP kernels:
    rtk_rope_merge / rtk_rope_table   M-RoPE section merge, cos/sin tables        (:68-74, :249, :298)
    rtk_pivotkv_score                 un-rotate + softmax column mass per key     (:248-270)
    rtk_pivotkv_select                mask override, top-k, id gather + rescale    (:272-295)
    rtk_pivotkv_evict                 append + gather + re-rotate + compaction     (:238, :278-318)

Memory layout (differs from the reference on purpose): each layer owns ONE pre-allocated
[1, Hkv, capacity, D] key and value buffer.  A chunk is appended at the tail (that view is what
`update` returns to the layer's attention), the kept rows are staged and committed over the tail's
head once the layer's attention has consumed the view (next `update` of the layer, `after_forward`,
or any access to `key_cache` / `value_cache`).  This removes the reference's two O(cache) torch.cat
rebuilds per (layer, chunk).
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Any, Dict, List, Optional, Tuple

import torch

from . import _native as nv

try:  # the HF classes are third-party; they only matter for isinstance checks inside `generate`
    from transformers.cache_utils import DynamicCache as _HFDynamicCache
except Exception:  # noqa: BLE001
    _HFDynamicCache = None

__all__ = ["repeat_kv", "rotate_half", "apply_multimodal_rotary_pos_emb", "apply_rotary_pos_emb", "PivotKVCache",
           "build_kvcache", "DynamicCache"]


# ---------------------------------------------------------------------------------------------------
# small torch helpers of the reference surface (longvideo_cache.py:16-116); used by the attention
# patch on the current chunk, not by the cache's hot path
# ---------------------------------------------------------------------------------------------------
def repeat_kv(hidden_states: torch.Tensor, n_rep: int) -> torch.Tensor:
    """[B, Hkv, L, D] -> [B, Hkv*n_rep, L, D]  (longvideo_cache.py:16-25)."""
    b, h, s, d = hidden_states.shape
    if n_rep == 1:
        return hidden_states
    return hidden_states[:, :, None, :, :].expand(b, h, n_rep, s, d).reshape(b, h * n_rep, s, d)


def rotate_half(x):
    """cat(-x[D/2:], x[:D/2])  (longvideo_cache.py:28-32)."""
    half = x.shape[-1] // 2
    return torch.cat((-x[..., half:], x[..., :half]), dim=-1)


def _rotate(q, k, cos, sin, reverse, attention_scaling):
    if reverse:  # rotate towards the opposite direction (longvideo_cache.py:76-78)
        q_embed = ((q * cos) - (rotate_half(q) * sin)) / attention_scaling ** 2
        k_embed = ((k * cos) - (rotate_half(k) * sin)) / attention_scaling ** 2
    else:
        q_embed = (q * cos) + (rotate_half(q) * sin) if q is not None else None
        k_embed = (k * cos) + (rotate_half(k) * sin) if k is not None else None
    return q_embed, k_embed


def apply_multimodal_rotary_pos_emb(q, k, cos, sin, mrope_section, unsqueeze_dim=1, reverse=False,
                                    attention_scaling=1):
    """M-RoPE with the reference's extra `reverse` / `attention_scaling` arguments (longvideo_cache.py:35-83)."""
    sections = mrope_section * 2
    cos = torch.cat([m[i % 3] for i, m in enumerate(cos.split(sections, dim=-1))], dim=-1).unsqueeze(unsqueeze_dim)
    sin = torch.cat([m[i % 3] for i, m in enumerate(sin.split(sections, dim=-1))], dim=-1).unsqueeze(unsqueeze_dim)
    return _rotate(q, k, cos, sin, reverse, attention_scaling)


def apply_rotary_pos_emb(q, k, cos, sin, position_ids=None, unsqueeze_dim=1, reverse=False, attention_scaling=1):
    """1-D RoPE with `reverse` / `attention_scaling` (longvideo_cache.py:86-116)."""
    return _rotate(q, k, cos.unsqueeze(unsqueeze_dim), sin.unsqueeze(unsqueeze_dim), reverse, attention_scaling)


# ---------------------------------------------------------------------------------------------------
# cache base with transformers==4.48 DynamicCache public behaviour (key_cache / value_cache lists)
# ---------------------------------------------------------------------------------------------------
def _hf_dynamic_cache_is_legacy() -> bool:
    if _HFDynamicCache is None:
        return False
    try:
        return hasattr(_HFDynamicCache(), "key_cache")
    except Exception:  # noqa: BLE001
        return False


class _ListDynamicCache:
    """Minimal stand-in used when the installed transformers no longer has the 4.48 list-based
    DynamicCache the reference subclasses (third-party API, restated from its documentation)."""

    def __init__(self, *args, **kwargs) -> None:
        self.key_cache: List[torch.Tensor] = []
        self.value_cache: List[torch.Tensor] = []
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

    def get_seq_length(self, layer_idx: int = 0) -> int:
        if len(self.key_cache) <= layer_idx or len(self.key_cache[layer_idx]) == 0:
            return 0
        return self.key_cache[layer_idx].shape[-2]

    def get_max_cache_shape(self):
        return None

    def get_max_length(self):
        return None

    def __len__(self):
        return len(self.key_cache)

    def __getitem__(self, layer_idx):
        return self.key_cache[layer_idx], self.value_cache[layer_idx]

    def __iter__(self):
        for i in range(len(self)):
            yield self.key_cache[i], self.value_cache[i]


DynamicCache = _HFDynamicCache if _hf_dynamic_cache_is_legacy() else _ListDynamicCache


class _LayerStore:
    """One layer's pre-allocated K/V buffers [1, Hkv, cap, D] plus the staged kept rows."""

    __slots__ = ("k", "v", "length", "pending", "k_stage", "v_stage", "pending_keep", "pending_event", "pending_pos")

    def __init__(self):
        self.k = self.v = None
        self.length = 0          # committed tokens
        self.pending = 0         # uncompressed chunk tokens sitting at [length, length+pending)
        self.k_stage = self.v_stage = None
        self.pending_keep = 0
        self.pending_event = None  # side-stream completion of the staged rows (overlap_streams > 0)
        self.pending_pos = None    # compressed position ids not yet appended to position_cache


class _Side:
    """A worker stream with its own scratch buffers (overlap_streams > 0)."""

    def __init__(self, device):
        self.stream = torch.cuda.Stream(device=device)
        self.ws: Dict[str, torch.Tensor] = {}


class _CacheView:
    """List-like view handed out as `key_cache` / `value_cache`: indexing commits pending compaction
    first, so readers always see the compacted cache exactly like the reference's lists."""

    def __init__(self, owner: "PivotKVCache", which: str):
        self._o, self._w = owner, which

    def __len__(self):
        return len(self._o._layers)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        if i < 0:
            i += len(self)
        st = self._o._layers[i]
        if st.k is None:
            return []
        self._o._commit(i)
        buf = st.k if self._w == "k" else st.v
        return buf[:, :, :st.length]

    def __setitem__(self, i, value):
        # external writers (e.g. HF crop / reorder utilities) replace a layer wholesale
        self._o._adopt(i, self._w, value)

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]

    def append(self, value):
        self._o._layers.append(_LayerStore())
        if not (isinstance(value, list) and len(value) == 0):
            self._o._adopt(len(self._o._layers) - 1, self._w, value)


class PivotKVCache(DynamicCache):
    """Drop-in for the reference's PivotKVCache (longvideo_cache.py:119-323)."""

    def __init__(self, config) -> None:
        self._layers: List[_LayerStore] = []
        self._kview = _CacheView(self, "k")
        self._vview = _CacheView(self, "v")
        super().__init__()
        self._layers = []  # drop whatever the base class appended through the views
        self.config = config
        llm_config = config.text_config if hasattr(config, "text_config") else config  # LLaVA-OneVision / Qwen2-VL
        self.hidden_size = llm_config.hidden_size
        self.num_hidden_layers = llm_config.num_hidden_layers
        self.num_heads = llm_config.num_attention_heads
        self.head_dim = self.hidden_size // self.num_heads
        self.num_key_value_heads = llm_config.num_key_value_heads
        self.num_key_value_groups = self.num_heads // self.num_key_value_heads

        kv_compression_kwargs = config.longvideo_kwargs["kvcache_compression_kwargs"]
        self.kvcache_compression = True
        self.kv_compression_kwargs = kv_compression_kwargs
        self.compression_ratio = kv_compression_kwargs["compression_ratio"]  # captured at construction (P0b)
        self.compression_method = kv_compression_kwargs["compression_method"]
        self.pos_embed_reforge = kv_compression_kwargs.get("pos_embed_reforge", False)
        # MI355X build option: compute cos/sin tables in a HIP kernel from rotary_emb.inv_freq instead of
        # calling the rotary module (valid for the default / YaRN inv_freq*position rotary modules)
        self.native_rope = bool(kv_compression_kwargs.get("native_rope", False))
        # MI355X build option: run scoring / selection / eviction of each update on one of N worker HIP
        # streams.  Only the tail append stays on the caller's stream (it is all the layer's attention
        # needs); the staged rows are committed after waiting for the worker's event.  Independent
        # updates then overlap on the GPU (the one-workgroup select kernel hides under MFMA kernels).
        self.overlap_streams = int(kv_compression_kwargs.get("overlap_streams", 0))
        self._sides: List[_Side] = []
        self._side_rr = 0
        self._position_cache: List[torch.Tensor] = []
        self.num_evicted_tokens: List[int] = []
        self.keypatches_mask_chunk = None
        self._ws: Dict[str, torch.Tensor] = {}
        self._warned = False

    # ---- list views --------------------------------------------------------------------------
    @property
    def key_cache(self):
        return self._kview

    @key_cache.setter
    def key_cache(self, value):  # the base class assigns [] in __init__
        self._layers = []
        for v in value:
            self._kview.append(v)

    @property
    def value_cache(self):
        return self._vview

    @value_cache.setter
    def value_cache(self, value):
        for i, v in enumerate(value):
            if i >= len(self._layers):
                self._layers.append(_LayerStore())
            if not (isinstance(v, list) and len(v) == 0):
                self._adopt(i, "v", v)

    def _adopt(self, i: int, which: str, value):
        st = self._layers[i]
        self._commit(i)
        if isinstance(value, list) and len(value) == 0:
            st.k = st.v = None
            st.length = 0
            return
        if which == "k":
            st.k = value
            st.length = value.shape[2]
        else:
            st.v = value

    def get_seq_length(self, layer_idx: int = 0) -> int:
        if len(self._layers) <= layer_idx or self._layers[layer_idx].k is None:
            return 0
        st = self._layers[layer_idx]
        return st.length + (st.pending_keep if st.pending else 0)

    def __len__(self):
        return len(self._layers)

    def __getitem__(self, layer_idx):
        return self.key_cache[layer_idx], self.value_cache[layer_idx]

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]

    @property
    def position_cache(self):
        """Per-layer position ids of the cached tokens (reference :143).  Reading it flushes deferred work."""
        for i in range(len(self._layers)):
            if self._layers[i].pending_pos is not None:
                self._commit(i)
        return self._position_cache

    @position_cache.setter
    def position_cache(self, value):
        self._position_cache = value

    # ---- hooks (reference :146-150): after_forward is where deferred compaction is flushed -------
    def before_forward(self, **kwargs):
        pass

    def after_forward(self, **kwargs):
        for i in range(len(self._layers)):
            self._commit(i)

    # ---- bookkeeping lists (reference :152-215) ------------------------------------------------
    def update_num_evicted_tokens(self, num_tokens: int, layer_idx: int):
        """num_evicted_tokens[layer] += num_tokens, padding skipped layers with 0 (longvideo_cache.py:152-177)."""
        if len(self.num_evicted_tokens) <= layer_idx:
            self.num_evicted_tokens.extend([0] * (layer_idx - len(self.num_evicted_tokens)))
            self.num_evicted_tokens.append(num_tokens)
        else:
            self.num_evicted_tokens[layer_idx] += num_tokens
        return self.num_evicted_tokens[layer_idx]

    def update_position_ids(self, position_ids: torch.Tensor, layer_idx: int):
        """position_cache[layer] = cat(prev, position_ids, dim=-1), padding skipped layers with []
        (longvideo_cache.py:179-209)."""
        pc = self._position_cache
        if len(pc) <= layer_idx:
            pc.extend([[] for _ in range(layer_idx - len(pc))])
            pc.append(position_ids)
        elif len(pc[layer_idx]) == 0:
            pc[layer_idx] = position_ids
        else:
            pc[layer_idx] = torch.cat([pc[layer_idx], position_ids], dim=-1)
        return pc[layer_idx]

    def get_prev_temporal_idx(self, layer_idx: int):
        """Last temporal id stored for the layer, -1 if none (longvideo_cache.py:211-215)."""
        if len(self._layers) > layer_idx and self._layers[layer_idx].pending_pos is not None:
            self._commit(layer_idx)  # ids of the previous chunk are still on a worker stream
        if len(self._position_cache) <= layer_idx:
            return -1
        cache_layer = self._position_cache[layer_idx]
        return cache_layer[0, 0, -1] if cache_layer.ndim == 3 else cache_layer[0, -1]

    # ---- storage -------------------------------------------------------------------------------
    def _store(self, layer_idx: int) -> _LayerStore:
        while len(self._layers) <= layer_idx:
            self._layers.append(_LayerStore())
        return self._layers[layer_idx]

    def reserve(self, layer_idx: int, tokens: int, like: torch.Tensor):
        """Make room for `tokens` more rows after the committed length of the layer."""
        st = self._store(layer_idx)
        need = st.length + tokens
        if st.k is not None and st.k.shape[2] >= need and st.k.is_contiguous():
            return st
        cap = max(need, 2 * (st.k.shape[2] if st.k is not None else 0), 1024)
        shape = (1, like.shape[1], cap, like.shape[3])
        nk = torch.empty(shape, dtype=like.dtype, device=like.device)
        nvv = torch.empty(shape, dtype=like.dtype, device=like.device)
        if st.k is not None and st.length:
            nk[:, :, :st.length].copy_(st.k[:, :, :st.length])
            nvv[:, :, :st.length].copy_(st.v[:, :, :st.length])
        st.k, st.v = nk, nvv
        return st

    def _commit(self, layer_idx: int):
        """Move the staged kept rows over the head of the uncompressed tail (reference :313-318)."""
        st = self._layers[layer_idx]
        if not st.pending:
            return
        keep, H, D = st.pending_keep, st.k.shape[1], st.k.shape[3]
        cap = st.k.shape[2]
        off = st.length * D * st.k.element_size()
        with torch.cuda.device(st.k.device):
            if st.pending_event is not None:  # staged rows were produced on a worker stream
                torch.cuda.current_stream().wait_event(st.pending_event)
                st.pending_event = None
            if st.pending_pos is not None:
                st.pending_pos.record_stream(torch.cuda.current_stream())
                self.update_position_ids(st.pending_pos, layer_idx)
                st.pending_pos = None
            s = nv.stream()
            dt = nv.dtype_code(st.k)
            sst = st.k_stage.shape[2] * D  # staging head stride (its capacity may exceed this chunk's keep)
            nv.check(nv.lib.rtk_pivotkv_commit(nv.ptr(st.k_stage), nv.ptr(st.v_stage), sst,
                                               C.c_void_p(st.k.data_ptr() + off), C.c_void_p(st.v.data_ptr() + off),
                                               cap * D, H, keep, D, dt, s), "rtk_pivotkv_commit")
        st.length += keep
        st.pending = 0
        st.pending_keep = 0

    def _next_side(self, device) -> Optional[_Side]:
        if self.overlap_streams <= 0:
            return None
        while len(self._sides) < self.overlap_streams:
            self._sides.append(_Side(device))
        side = self._sides[self._side_rr % len(self._sides)]
        self._side_rr += 1
        return side

    def _buf(self, name: str, shape, dtype, device, ws: Optional[dict] = None) -> torch.Tensor:
        ws = self._ws if ws is None else ws
        t = ws.get(name)
        n = 1
        for s in shape:
            n *= s
        if t is None or t.numel() < n or t.dtype != dtype or t.device != device:
            t = torch.empty(max(n, 1), dtype=dtype, device=device)
            ws[name] = t
        return t[:n].view(*shape)

    def _rope_tables(self, name, rotary_emb_fn, x_like, position_ids, mrope_section, n, D, ws=None):
        """fp32 [n, D] cos/sin tables of `position_ids`, section-merged (reference :249 + :68-74)."""
        dev = x_like.device
        cos_t = self._buf(name + "_cos", (n, D), torch.float32, dev, ws)
        sin_t = self._buf(name + "_sin", (n, D), torch.float32, dev, ws)
        P = 3 if position_ids.ndim == 3 else 1
        sec = (C.c_int * len(mrope_section))(*mrope_section) if mrope_section else None
        nsec = len(mrope_section) if mrope_section else 0
        s = nv.stream()
        if self.native_rope and hasattr(rotary_emb_fn, "inv_freq"):
            pos = position_ids.reshape(P, n)
            if not pos.is_contiguous():
                pos = pos.contiguous()
            inv = rotary_emb_fn.inv_freq
            if inv.device != dev or inv.dtype != torch.float32 or not inv.is_contiguous():
                inv = inv.to(device=dev, dtype=torch.float32).contiguous()
            nv.check(nv.lib.rtk_rope_table(nv.ptr(pos), P, n, nv.ptr(inv), D, float(rotary_emb_fn.attention_scaling),
                                           sec, nsec, int(x_like.dtype == torch.bfloat16), nv.ptr(cos_t), nv.ptr(sin_t),
                                           s), "rtk_rope_table")
            return cos_t, sin_t
        cos, sin = rotary_emb_fn(x_like, position_ids)  # third-party module, exactly as the reference calls it
        cos = cos.reshape(P, n, D)
        sin = sin.reshape(P, n, D)
        if not cos.is_contiguous():
            cos = cos.contiguous()
        if not sin.is_contiguous():
            sin = sin.contiguous()
        nv.check(nv.lib.rtk_rope_merge(nv.ptr(cos), nv.ptr(sin), P, n, D, nv.dtype_code(cos), sec, nsec, nv.ptr(cos_t),
                                       nv.ptr(sin_t), s), "rtk_rope_merge")
        return cos_t, sin_t

    # ---- the hot path ---------------------------------------------------------------------------
    def update(
        self,
        key_states: torch.Tensor,
        value_states: torch.Tensor,
        layer_idx: int,
        cache_kwargs: Optional[Dict[str, Any]] = None,
    ) -> Tuple[torch.Tensor, torch.Tensor]:
        """
        Input
            query_states: [bsz, num_heads, q_len, d]      (cache_kwargs['query_states'], post-RoPE)
            key_states:   [bsz, num_key_value_heads, q_len, d]
            position_ids: [3, bsz, q_len] / [bsz, q_len]  (cache_kwargs['position_ids'])
        Output
            key_states_output, value_states_output: the layer's UNCOMPRESSED keys/values
            ([prefix | whole current chunk]) for this layer's self attention (reference :217-323).
        """
        if not self._warned:
            self._warned = True
            print("Enable PivotKVCache compression: length after compression %.2f" % (self.compression_ratio))
        cache_kwargs = cache_kwargs if cache_kwargs is not None else {}
        position_ids = cache_kwargs.pop("position_ids", None)
        nv.require_device(key_states, value_states)
        assert key_states.shape[0] == 1, "PivotKVCache supports bsz == 1 only"

        # 1) append: the next layer's hidden states see the uncompressed chunk (reference :238)
        n_new = key_states.shape[2]
        if len(self._layers) > layer_idx and self._layers[layer_idx].k is not None:
            self._commit(layer_idx)
        st = self.reserve(layer_idx, n_new, key_states)
        P0 = st.length

        if not self.kvcache_compression:  # text prefill / decode (reference :319-321)
            st.k[:, :, P0:P0 + n_new].copy_(key_states)
            st.v[:, :, P0:P0 + n_new].copy_(value_states)
            st.length += n_new
            if self.pos_embed_reforge:
                self.update_position_ids(position_ids, layer_idx)
            return st.k[:, :, :st.length], st.v[:, :, :st.length]

        query_states = cache_kwargs.pop("query_states")
        rotary_emb_fn = cache_kwargs.pop("rotary_emb")
        mrope_section = cache_kwargs.pop("mrope_section", None)  # M-RoPE only
        bsz, num_heads, q_len, head_dim = query_states.shape
        num_key_value_heads, k_len = key_states.shape[1:3]
        assert bsz == 1
        nv.require_device(query_states)
        dev, dt = key_states.device, nv.dtype_code(key_states)
        D, L, Hkv, Hq = head_dim, q_len, num_key_value_heads, num_heads
        for t in (query_states, key_states, value_states):
            if t.stride(-1) != 1:
                raise ValueError("q/k/v must be contiguous along head_dim")
        keep_len = max(1, int(self.compression_ratio * q_len))  # evict new tokens only (reference :263)

        mask = getattr(self, "keypatches_mask_chunk", None)
        if mask is not None:
            nv.require_device(mask)
            if mask.dtype != torch.bool or not mask.is_contiguous():
                mask = mask.to(torch.bool).contiguous()
            assert mask.numel() == L, "keypatches_mask_chunk must have one entry per chunk token"
        if st.k_stage is None or st.k_stage.shape[2] < keep_len or st.k_stage.dtype != key_states.dtype:
            st.k_stage = torch.empty((1, Hkv, keep_len, D), dtype=key_states.dtype, device=dev)
            st.v_stage = torch.empty((1, Hkv, keep_len, D), dtype=key_states.dtype, device=dev)

        def compress(ws, append_tail: bool):
            """score -> select -> eviction scan on the CURRENT stream, scratch from `ws`."""
            s = nv.stream()
            cos_t = sin_t = None
            if self.pos_embed_reforge:
                cos_t, sin_t = self._rope_tables("old", rotary_emb_fn, value_states, position_ids, mrope_section, L, D,
                                                 ws)
            # 2) score (reference :248-270)
            ws_bytes = nv.lib.rtk_pivotkv_score_workspace_bytes(Hq, Hkv, L, D, dt)
            wsb = self._buf("score_ws", (ws_bytes + 256,), torch.uint8, dev, ws)
            ws_ptr = (wsb.data_ptr() + 255) & ~255
            score = self._buf("score", (L,), torch.float32, dev, ws)
            k_unrot = self._buf("k_unrot", (Hkv, L, D), key_states.dtype, dev, ws)
            nv.check(nv.lib.rtk_pivotkv_score(
                nv.ptr(query_states), query_states.stride(1), query_states.stride(2),
                nv.ptr(key_states), key_states.stride(1), key_states.stride(2),
                Hq, Hkv, L, D, dt, nv.ptr(cos_t), nv.ptr(sin_t),
                float(getattr(rotary_emb_fn, "attention_scaling", 1.0)) if self.pos_embed_reforge else 1.0,
                nv.ptr(score), nv.ptr(k_unrot), C.c_void_p(ws_ptr), ws_bytes, s), "rtk_pivotkv_score")
            # 3) mask override + top-k + position ids (reference :272-295)
            keep_idx = self._buf("keep_idx", (keep_len,), torch.int64, dev, ws)
            rank = self._buf("rank", (L,), torch.int32, dev, ws)
            pos_in = pos_out = None
            Pn = 0
            if position_ids is not None:
                Pn = 3 if position_ids.ndim == 3 else 1
                pos_in = position_ids.reshape(Pn, L)
                if not pos_in.is_contiguous():
                    pos_in = pos_in.contiguous()
                pos_out = torch.empty((Pn, keep_len), dtype=torch.int64, device=dev)  # becomes part of position_cache
            nv.check(nv.lib.rtk_pivotkv_select(nv.ptr(score), nv.ptr(mask), L, keep_len, nv.ptr(pos_in), Pn,
                                               int(bool(self.pos_embed_reforge)), nv.ptr(keep_idx), nv.ptr(rank),
                                               nv.ptr(pos_out), s), "rtk_pivotkv_select")
            cpos = None
            if pos_out is not None:
                cpos = pos_out.view(3, 1, keep_len) if Pn == 3 else pos_out.view(1, keep_len)
            # 4) eviction scan: (append +) gather (+ re-rotate at the new ids) (reference :238, :278-306)
            cos_n = sin_n = None
            if self.pos_embed_reforge:
                cos_n, sin_n = self._rope_tables("new", rotary_emb_fn, value_states[:, :, :1], cpos, mrope_section,
                                                 keep_len, D, ws)
            cap = st.k.shape[2]
            esz = st.k.element_size()
            k_tail = C.c_void_p(st.k.data_ptr() + P0 * D * esz) if append_tail else None
            v_tail = C.c_void_p(st.v.data_ptr() + P0 * D * esz) if append_tail else None
            nv.check(nv.lib.rtk_pivotkv_evict(
                nv.ptr(key_states), key_states.stride(1), key_states.stride(2),
                nv.ptr(value_states), value_states.stride(1), value_states.stride(2),
                nv.ptr(k_unrot), Hkv, L, D, dt, nv.ptr(keep_idx), keep_len, nv.ptr(cos_n), nv.ptr(sin_n),
                k_tail, v_tail, cap * D, nv.ptr(st.k_stage), nv.ptr(st.v_stage), st.k_stage.shape[2] * D, s),
                "rtk_pivotkv_evict")
            self.last_keep_indices = keep_idx  # scratch views, valid until the worker's next update (diagnostics)
            self.last_scores = score
            return cpos

        with torch.cuda.device(dev):
            side = self._next_side(dev)
            if side is None:
                compressed_position_ids = compress(self._ws, append_tail=True)
                if self.pos_embed_reforge:  # bookkeeping (reference :308-309)
                    self.update_position_ids(compressed_position_ids, layer_idx)
            else:
                main = torch.cuda.current_stream()
                st.k[:, :, P0:P0 + n_new].copy_(key_states)      # the append is all this layer's attention needs
                st.v[:, :, P0:P0 + n_new].copy_(value_states)
                ready = torch.cuda.Event()
                ready.record(main)
                for t in (query_states, key_states, value_states, position_ids, mask):
                    if t is not None:
                        t.record_stream(side.stream)
                with torch.cuda.stream(side.stream):
                    side.stream.wait_event(ready)
                    compressed_position_ids = compress(side.ws, append_tail=False)
                    done = torch.cuda.Event()
                    done.record(side.stream)
                st.pending_event = done
                st.pending_pos = compressed_position_ids if self.pos_embed_reforge else None
        self.update_num_evicted_tokens(k_len - keep_len, layer_idx)  # reference :310
        st.pending = n_new
        st.pending_keep = keep_len
        return st.k[:, :, :P0 + n_new], st.v[:, :, :P0 + n_new]


def build_kvcache(config):
    """DynamicCache unless longvideo_kwargs enables 'pivotkv' compression (reference :326-334)."""
    if getattr(config, "longvideo_kwargs", None) is None or not config.longvideo_kwargs.get("kvcache_compression", False):
        return DynamicCache()
    compression_method = config.longvideo_kwargs["kvcache_compression_kwargs"]["compression_method"]
    if compression_method.lower() == "pivotkv":
        return PivotKVCache(config)
    raise NotImplementedError
