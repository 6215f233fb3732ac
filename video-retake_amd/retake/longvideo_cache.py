"""PivotKV cache on MI355X.  Same surface as the reference's retake/longvideo_cache.py.

`PivotKVCache.update` (reference :217-323) keeps its signature, its `cache_kwargs` protocol (pops
`position_ids`, `query_states`, `rotary_emb`, `mrope_section`) and its return value (the UNCOMPRESSED
keys/values of the layer); everything it computes runs as HIP kernels behind the C ABI (include/retake_hip.h):
    rtk_pivotkv_update    one call per update: fused prepare (rotary tables + un-rotate q / k + append k / v, :238,
                          :248-259) - or, from the pre-RoPE projections, the whole attention prologue
                          (`update_pre_rope`: continuity shift + tables + RoPE + append + scoring operands)
    rtk_pivotkv_flush     one call per chunk, all pending layers: the two score passes (:260-268), finalize +
                          mask override + top-k + id gather / rescale (:269-295), and ONE in-place compaction launch
                          (kept K re-rotated at the new ids, kept V compacted inside the tail, ids; :278-318)
    rtk_pivotkv_append_rope   text prefill / decode segments (`append_pre_rope`; the else-branch :319-321)
The stage-by-stage entry points (rtk_pivotkv_prepare / _score_stages / _select / _evict_batched ...) serve the shapes
the one-call path does not: small chunks, fp32 passes, rotary modules that must be called, worker streams.

Memory layout (differs from the reference on purpose): each layer owns ONE pre-allocated
[1, Hkv, capacity, D] key and value buffer and one [P, capacity] position-id buffer.  `update` appends
the chunk at the tail (that view is what it returns to the layer's attention), prepares the scoring operands in the
layer's slot of a per-chunk batch and returns.  The eviction itself - scoring, selection, gather of the kept
rows, re-rotation of the kept keys at their new ids, compaction over the head of the tail, position
bookkeeping - is deferred until the view has been consumed and then flushed for ALL pending layers at
once (`after_forward`, which the reference calls after every video chunk; or the next `update` of a
pending layer; or any access to `key_cache` / `value_cache` / `position_cache`; or, with `flush_every_layers: N`, every N
layers).  This removes the reference's two O(cache) torch.cat rebuilds per (layer, chunk) and turns 28 launch-bound
evictions into one bandwidth-bound launch.  `memory_footprint()` says what all of it costs in bytes.
"""
from __future__ import annotations

import ctypes as C
import math
import sys
import weakref
from typing import Any, Dict, List, Optional, Tuple

import torch

from . import _native as nv

try:  # the HF classes are third-party; they only matter for isinstance checks inside `generate`
    from transformers.cache_utils import DynamicCache as _HFDynamicCache
except Exception:  # noqa: BLE001
    _HFDynamicCache = None

_WARNED = set()


def _warn_once(msg: str):
    if msg not in _WARNED:
        _WARNED.add(msg)
        print(msg, file=sys.stderr)


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
    """One layer's pre-allocated K/V buffers [1, Hkv, cap, D] and position ids [P, cap].  The numbers live in a
    rtk_layer_state block (`c`) the library reads and advances itself (rtk_pivotkv_update / rtk_pivotkv_flush)."""

    __slots__ = ("c", "cref", "_k", "_v", "_pos", "pending_event", "pos_ndim")

    def __init__(self):
        self.c = nv.LayerState()
        self.cref = C.addressof(self.c)
        self._k = self._v = self._pos = None
        self.pending_event = None  # worker-stream completion of this layer's scoring (overlap_streams > 0)
        self.pos_ndim = 0          # 3: ids are [3, 1, n] (M-RoPE), 2: [1, n]

    def _sync(self):
        k, v, c = self._k, self._v, self.c
        c.k = k.data_ptr() if k is not None else None
        c.v = v.data_ptr() if v is not None else None
        # the library may only use the buffers when both are dense [1, Hkv, cap, D] blocks of one capacity
        ok = (k is not None and v is not None and k.ndim == 4 and k.shape == v.shape and k.is_contiguous()
              and v.is_contiguous())
        c.cap = k.shape[2] if ok else 0

    @property
    def k(self):
        return self._k

    @k.setter
    def k(self, t):
        self._k = t
        self._sync()

    @property
    def v(self):
        return self._v

    @v.setter
    def v(self, t):
        self._v = t
        self._sync()

    @property
    def pos(self):           # int64 [P, cap]: position ids of the cached tokens (pos_embed_reforge)
        return self._pos

    @pos.setter
    def pos(self, t):
        self._pos = t
        self.c.pos = t.data_ptr() if t is not None else None
        self.c.pos_cap = t.shape[1] if t is not None else 0

    @property
    def length(self):        # committed tokens
        return self.c.length

    @length.setter
    def length(self, n):
        self.c.length = n

    @property
    def pending(self):       # uncompressed chunk tokens sitting at [length, length + pending)
        return self.c.pending

    @pending.setter
    def pending(self, n):
        self.c.pending = n

    @property
    def pending_keep(self):
        return self.c.pending_keep

    @pending_keep.setter
    def pending_keep(self, n):
        self.c.pending_keep = n

    @property
    def pos_len(self):
        return self.c.pos_len

    @pos_len.setter
    def pos_len(self, n):
        self.c.pos_len = n


class _Side:
    """A worker stream with its own scratch buffers (overlap_streams > 0)."""

    def __init__(self, device):
        self.stream = torch.cuda.Stream(device=device)
        self.ws: Dict[str, torch.Tensor] = {}


class _Rotary:
    """What the kernels need of an inv_freq * position rotary module (native RoPE): its inv_freq on the device and
    attention_scaling.  HF builds one rotary module per attention layer; modules with equal contents share an entry."""

    __slots__ = ("inv", "scaling", "device")

    def __init__(self, inv, scaling, device):
        self.inv, self.scaling, self.device = inv, scaling, device


def _version_of(t: torch.Tensor):
    """The tensor's version counter, None for tensors that do not keep one (created under torch.inference_mode()): for
    those a shift the previous layer's launch has already applied is simply applied again (it is idempotent)."""
    return None if t.is_inference() else t._version


def _inv_stamp(rotary_emb_fn):
    """Identity + write counter of a rotary module's inv_freq: the native-RoPE snapshot of the module (_Rotary) is only
    valid while this is unchanged (a module whose inv_freq is re-assigned or modified in place is snapshotted again)."""
    inv = getattr(rotary_emb_fn, "inv_freq", None)
    return (inv.data_ptr(), _version_of(inv)) if isinstance(inv, torch.Tensor) else None


class _Batch:
    """Per-chunk batch of pending evictions: slot = layer index.  All units share the chunk geometry."""

    def __init__(self, key, slots, Hq, Hkv, L, D, keep, P, reforge, dtype, device, refround=False, fast=False,
                 keep_all=False, skip_masked=True, in_place_compaction=True):
        self.key, self.slots, self.keep, self.P, self.reforge = key, slots, keep, P, reforge
        self.wrap = False          # flush_every_layers: slot = layer % slots (else slot == layer)
        self.keep_all = keep_all   # keep == L and no scoring asked for: the selection is the identity
        # dtype code of the scoring entry points: bf16 payloads with the reference's bf16 rounding chain, or through the
        # fp16 matrix instruction with pre-scaled queries (score_rounding="fast"), on request
        self.score_dt = ((nv.RTK_BF16_REFROUND if refround else (nv.RTK_BF16_FAST if fast else nv.RTK_BF16))
                         if dtype == torch.bfloat16 else
                         ((nv.RTK_F16_REFROUND if refround else nv.RTK_F16) if dtype == torch.float16 else nv.RTK_F32))
        self.fast = self.score_dt == nv.RTK_BF16_FAST
        self.batched_passes = dtype in (torch.bfloat16, torch.float16) and D == 128 and L >= 512
        if self.batched_passes:   # all layers of a chunk per launch: splits chosen for the stream length (same flag everywhere)
            self.score_dt |= nv.RTK_SCORE_MANY_UNITS
        # what rtk_pivotkv_prepare is told: the payload dtype (the reference-rounding mode prepares like plain bf16)
        self.prep_dt = ((nv.RTK_BF16 if dtype == torch.bfloat16 else nv.RTK_F16) if refround else self.score_dt & 0xFF) \
            | (self.score_dt & ~0xFF)
        self.Hkv, self.L, self.D, self.dtype, self.device = Hkv, L, D, dtype, device
        self.Hq = Hq
        self.keep_idx = torch.arange(keep, dtype=torch.int64, device=device).repeat(slots, 1) if keep_all \
            else torch.empty((slots, keep), dtype=torch.int64, device=device)
        self.pos_new = torch.empty((P, slots, keep), dtype=torch.int64, device=device) if P else None
        # deferred selection (flushed for all layers at once): per-slot column partials of the scoring passes, the
        # final score, a private copy of the chunk's position ids (the caller shifts its tensor in place for the next
        # layer), the key-patch mask of the update and the selection scratch
        self.rs_n = C.c_int(0)
        self.part_floats = nv.lib.rtk_pivotkv_score_partials(Hq, Hkv, L, D, self.score_dt, C.byref(self.rs_n))
        self.pos_old = torch.empty((slots, P, L), dtype=torch.int64, device=device) if P else None
        self.sel_bytes = nv.lib.rtk_pivotkv_select_workspace_bytes(L)
        self.masks: Dict[int, Optional[torch.Tensor]] = {}
        self.selected = set()      # layers whose selection already ran inside update (small chunks)
        self.scored = set()        # layers whose matrix passes already ran inside update
        # one score workspace per slot (q~, lse partials): the matrix passes of all layers run in one launch each
        self.ws_bytes = nv.lib.rtk_pivotkv_score_workspace_bytes(Hq, Hkv, L, D, self.score_dt)
        self.ws_stride = (self.ws_bytes + 255) & ~255
        self.partials = self.score = self.sel_ws = self.score_ws = self.key_index = None
        self.score_ws_base = 0
        self.v_stage = None
        if reforge:  # kept K is re-rotated from the un-rotated copy straight into the cache: no K staging
            self.k_unrot = torch.empty((slots, Hkv, L, D), dtype=dtype, device=device)
            self.k_stage = None
        else:
            self.k_unrot = None
            self.k_stage = None if (keep_all or in_place_compaction) else torch.empty((slots, Hkv, keep, D), dtype=dtype, device=device)
        self.cos_new = self.sin_new = None   # tables of a third-party rotary module, allocated when one is used
        self.pending: List[int] = []
        self.c_pending = 0         # how many of them were appended by rtk_pivotkv_update (the one-call path)
        self.rotary_emb_fn = None
        self.rot: Optional[_Rotary] = None
        self.mrope_section = None
        self.x_like = None
        self.mask_obj = None       # the last key-patch mask tensor that passed validation, and its address
        self.mask_ptr = None
        self.shift_ids = None      # pre-RoPE units: the caller's ids tensor, shifted in place by the flush
        self.defer = False         # deferred re-rotation (PivotKVCache.defer_rerotation)
        self.qshape, self.kshape = torch.Size((1, Hq, L, D)), torch.Size((1, Hkv, L, D))
        self.dev_index = device.index if device.index is not None else torch.cuda.current_device()
        # the one-call path (rtk_pivotkv_update / rtk_pivotkv_flush): argument blocks bound once per batch
        self.c = nv.PivotKVBatch()
        self.cref = C.addressof(self.c)
        self.io = nv.UpdateIO()
        self.ioref = C.addressof(self.io)
        # RTK_UPDATE_SHIFT_NEXT's words (launch count + arrival counters of the prepare launch), zeroed once
        self.shift_ticket = torch.zeros(max(1, nv.lib.rtk_pivotkv_shift_ticket_ints(L, D)), dtype=torch.int32, device=device)
        self.io.ticket, self.io.ticket_ints = self.shift_ticket.data_ptr(), self.shift_ticket.numel()
        # ... and a word of PINNED HOST memory the watching workgroup increments if its bounded wait ever runs out (it then
        # shifts nothing): the host reads it without a device synchronisation (PivotKVCache._shift_latch_check raises)
        self.shift_status = torch.zeros(1, dtype=torch.int32, pin_memory=True) if device.type == "cuda" else None
        self.shift_latch = self.shift_status.numpy() if self.shift_status is not None else None
        self.io.status = self.shift_status.data_ptr() if self.shift_status is not None else None
        self.shift_stream = None   # the stream of the batch's last RTK_UPDATE_SHIFT_NEXT launch (the words serve one stream at a time)
        # prologue route: queries that are scored where they lie (no packed copy) - the pointers the library reads at
        # the flush, and the tensors themselves, kept alive until then
        self.q_units = (C.c_void_p * slots)()
        self.q_keep: List[Optional[torch.Tensor]] = [None] * slots
        c = self.c
        c.Hq, c.Hkv, c.L, c.D, c.keep, c.P, c.slots = Hq, Hkv, L, D, keep, P, slots
        c.dtype = nv.RTK_BF16 if dtype == torch.bfloat16 else (nv.RTK_F16 if dtype == torch.float16 else nv.RTK_F32)
        c.score_dtype, c.prep_dtype = self.score_dt, self.prep_dt
        c.reforge, c.keep_all, c.round_mode = int(reforge), int(keep_all), nv.round_mode(dtype)
        c.rs_n, c.skip_masked, c.batched_passes = self.rs_n.value, int(skip_masked), int(self.batched_passes)
        c.partial_floats = self.part_floats
        c.keep_idx = self.keep_idx.data_ptr()
        c.pos_new = self.pos_new.data_ptr() if P else None
        c.pos_old = self.pos_old.data_ptr() if P else None
        c.k_unrot = self.k_unrot.data_ptr() if reforge else None
        c.k_stage = self.k_stage.data_ptr() if self.k_stage is not None else None
        c.q_units = C.addressof(self.q_units)
        # the one-call path serves the deferred chip-wide selection (L >= 512) of reforging caches with position ids
        self.c_capable = bool(reforge and P and L >= 512)
        # rtk_pivotkv_flush compacts the tails in place in one launch (rtk_pivotkv_compact_batched): tickets and flags
        # of its workgroups live here, zeroed once; the staging rows are then only allocated by the stage-by-stage route
        self.compact_sync = None
        self.sync_stream = None    # the stream the batch's last in-place compaction was launched on
        if in_place_compaction and not keep_all:
            n_ints = nv.lib.rtk_pivotkv_compact_sync_ints(slots, Hkv, keep, D, c.dtype)
            if n_ints:
                self.compact_sync = torch.zeros(n_ints, dtype=torch.int32, device=device)
                c.compact_sync, c.compact_sync_ints = self.compact_sync.data_ptr(), n_ints
        if not keep_all:
            self.ensure_scoring()
            if self.compact_sync is None:
                self.ensure_staging()
        else:  # nothing is scored or staged: the scratch is allocated only if a route that needs it comes along
            self._dummy = torch.empty(512, dtype=torch.uint8, device=device)
            c.score_ws = (self._dummy.data_ptr() + 255) & ~255
            c.score_ws_stride, c.score_ws_bytes = 0, 0

    def slot(self, layer_idx: int) -> int:
        return layer_idx % self.slots if self.wrap else layer_idx

    def ensure_scoring(self):
        """Scoring scratch of every slot (q~ / lse workspace, column partials, scores, selection scratch, live-key
        lists); keep-all batches get it only on the routes that still un-rotate the queries."""
        if self.partials is not None:
            return
        slots, L, device, c = self.slots, self.L, self.device, self.c
        self.partials = torch.empty((slots, self.part_floats), dtype=torch.float32, device=device)
        self.score = torch.empty((slots, L), dtype=torch.float32, device=device)
        self.sel_ws = torch.empty((slots, self.sel_bytes), dtype=torch.uint8, device=device)
        self.score_ws = torch.empty(slots * self.ws_stride + 256, dtype=torch.uint8, device=device)
        self.score_ws_base = (self.score_ws.data_ptr() + 255) & ~255
        # pass 2's live-key lists (the unmasked tokens of every slot + their count): the columns the mask override
        # discards (reference :272-274) are not computed
        self.key_index = torch.empty((slots, L + 1), dtype=torch.int32, device=device)
        c.partials, c.score, c.sel_ws = self.partials.data_ptr(), self.score.data_ptr(), self.sel_ws.data_ptr()
        c.sel_ws_stride = self.sel_bytes
        c.score_ws, c.score_ws_stride, c.score_ws_bytes = self.score_ws_base, self.ws_stride, self.ws_bytes
        c.key_index = self.key_index.data_ptr()

    def ensure_staging(self):
        if self.v_stage is None:
            self.v_stage = torch.empty((self.slots, self.Hkv, self.keep, self.D), dtype=self.dtype, device=self.device)
            self.c.v_stage = self.v_stage.data_ptr()
            if not self.reforge and self.k_stage is None:
                self.k_stage = torch.empty_like(self.v_stage)
                self.c.k_stage = self.k_stage.data_ptr()

    def ensure_tables(self):
        if self.cos_new is None:
            self.cos_new = torch.empty((self.slots * self.keep, self.D), dtype=torch.float32, device=self.device)
            self.sin_new = torch.empty((self.slots * self.keep, self.D), dtype=torch.float32, device=self.device)


class _CacheView:
    """List-like view handed out as `key_cache` / `value_cache`: indexing flushes pending compaction
    first, so readers always see the compacted cache exactly like the reference's lists."""

    def __init__(self, owner: "PivotKVCache", which: str):
        # a weak reference: the cache owns its views, not the other way round - with a strong one the pair is a reference
        # cycle and a dropped cache (gigabytes of device memory) lives on until the garbage collector happens to run
        self._ref, self._w = weakref.ref(owner), which

    @property
    def _o(self) -> "PivotKVCache":
        o = self._ref()
        if o is None:
            raise ReferenceError("the PivotKVCache this view belongs to has been released")
        return o

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
        if st.pending:
            self._o._flush()
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

    def __init__(self, config, reserve_tokens: Optional[int] = None) -> None:
        """reserve_tokens (not in the reference): tokens per layer this cache is expected to hold at most - compressed
        prompt + one uncompressed chunk + generation.  The first allocation of a layer takes that size, which saves the
        geometric regrowth copies (~0.4 % of a 2048-frame prefill) and half the memory; without it buffers double."""
        self.reserve_tokens = int(reserve_tokens) if reserve_tokens else 0
        self._layers: List[_LayerStore] = []
        self._batch: Optional[_Batch] = None
        self._last_slot = None
        self._pos_layers = 0
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
        # cos/sin of the position ids are computed inside the HIP kernels from rotary_emb.inv_freq / attention_scaling
        # whenever the rotary module is of the static inv_freq * position kind (HF's default / linear / YaRN / llama3
        # rope types, which is every shipped config: configs/*.yaml patch YaRN, monkeypatch.py:24-48) - correctly
        # rounded sin / cos, within one fp32 ulp of the module's, kept keys inside the 1e-5 bar.  Any other module
        # (dynamic / longrope types, no inv_freq) is CALLED, exactly as the reference does (:249, :298).
        # native_rope: False forces the call for every module (the bit-faithful opt-out).
        self.native_rope = bool(kv_compression_kwargs.get("native_rope", True))
        self._rotaries: Dict[int, Tuple[Any, Optional[_Rotary], Any]] = {}
        self._aio = nv.UpdateIO()   # argument block of append_pre_rope
        # MI355X build option for 16-bit models: "fp32" (default) scores with exact bf16 / fp16 products, fp32 accumulation,
        # softmax and sums; "reference" reproduces the reference's own bf16 (fp16 on a float16 model) roundings of the
        # logits, probabilities, per-head sums and means (longvideo_cache.py:264-270) - coarser, but what the reference
        # computes; "fast" (bf16 only) is the two-instruction softmax on the fp16 matrix path
        self.score_rounding = str(kv_compression_kwargs.get("score_rounding", "fp32"))
        if self.score_rounding not in ("fp32", "reference", "fast"):
            raise ValueError(f"score_rounding must be 'fp32', 'reference' or 'fast', got {self.score_rounding!r}")
        # MI355X build option: run scoring / selection of each update on one of N worker HIP streams.  Only
        # the tail append stays on the caller's stream (it is all the layer's attention needs); the flush
        # waits for the workers' events.  Independent updates then overlap on the GPU.
        self.overlap_streams = int(kv_compression_kwargs.get("overlap_streams", 0))
        # the reference overwrites the score of every key-patch token with 1.0 (:272-274): the batched pass 2 does not
        # compute those columns.  Same kept set, same scores after the override; False restores the full pass (tests / A/B)
        self.skip_masked_columns = bool(kv_compression_kwargs.get("skip_masked_columns", True))
        # compression_ratio 1 (what `dynamic_compression_ratio` sets for every prompt within max_input_length,
        # qwen2_vl.py:553-554): keep == chunk length, so topk(k = L) followed by the ascending sort (:276-277) is the
        # identity whatever the scores are, and the id rescale (:288-292) multiplies by 1.0.  The scoring passes and the
        # selection are not run then - the K round trip through the un-rotated frame and the bookkeeping still are.
        # True restores them (tests, `last_scores`)
        self.score_when_keeping_all = bool(kv_compression_kwargs.get("score_when_keeping_all", False))
        # MI355X build option (retake/sharded.py sets it): the kept keys stay UN-rotated in the cache and their ids
        # provisional; the owner rotates them once, at their final ids (rtk_rope_rotate_rows), when the temporal offset
        # of its block is known.  key_cache then holds un-rotated rows until that call.
        self.defer_rerotation = bool(kv_compression_kwargs.get("defer_rerotation", False))
        # MI355X build option: what the attention prologue (update_pre_rope) hands to the score passes and the re-rotation.
        #   "reference" (default)  q~ / k~ = the reference's own operands: the un-rotation of the rotated rows, with the
        #                          model dtype's rounding per torch op (longvideo_cache.py:76-78, :248-259).  On a bf16 model
        #                          that round trip moves the exact score of a token by up to ~2e-2 (6 bf16 ulps; measured,
        #                          tests/golden/gen_golden.py --only pivotkv_prerope_bf16) - it is part of what the reference
        #                          computes, so scores, kept sets and kept keys follow it bit for bit;
        #   "pre_rope"             q~ := q0, k~ := k0 (equal up to that rounding; more accurate, not the
        #                          reference's bits) - no copy of the queries (they are scored where they lie).
        self.prologue_operands = str(kv_compression_kwargs.get("prologue_operands", "reference"))
        if self.prologue_operands not in ("reference", "pre_rope"):
            raise ValueError(f"prologue_operands must be 'reference' or 'pre_rope', got {self.prologue_operands!r}")
        # MI355X build option (tests / A-B): False makes the prologue route pack a copy of the queries for the score passes
        # even when they could be read where they lie (prologue_operands="pre_rope" only)
        self.score_queries_in_place = bool(kv_compression_kwargs.get("score_queries_in_place", True))
        # MI355X build option (tests / A-B): False makes rtk_pivotkv_flush stage the rows whose source lies inside the
        # destination range and place them with a second launch, instead of the one in-place compaction launch
        self.in_place_compaction = bool(kv_compression_kwargs.get("in_place_compaction", True))
        # MI355X build option (tests / A-B): False sends every update and flush through the stage-by-stage route
        # instead of the one-call entry points rtk_pivotkv_update / rtk_pivotkv_flush - same kernels, same results
        self.one_call_update = bool(kv_compression_kwargs.get("one_call_update", True))
        # MI355X build option: bound the scratch of the deferred eviction.  0 (default): one flush per chunk for ALL layers -
        # every layer's q~ / k~ / partials wait in their own slot until `after_forward` (scratch = layers x ~(Hq + Hkv) L D
        # elements: 1.4 GB at L = 6272, 28 layers, bf16).  N > 0: the batch has N slots (slot = layer mod N) and is flushed
        # whenever the next layer's slot is taken, i.e. every N layers - scratch / launches-per-chunk trade N/layers : layers/N
        self.flush_every_layers = int(kv_compression_kwargs.get("flush_every_layers", 0))
        # MI355X build option: on the reference's protocol (`update` on rotated tensors) the launch that un-rotates and
        # appends layer l's chunk can also apply layer l + 1's continuity shift (qwen2_vl.py:68-73) to the caller's ids, so
        # that `shift_temporal_ids_` of the attention patch launches once per chunk instead of once per layer.  It WRITES
        # the caller's `position_ids` - which the reference's `update` never does - so it is OPT-IN per call: only a caller
        # that passes cache_kwargs["shift_next_position_ids"] = True gets it (the Qwen2-VL patch does: it shares one ids
        # tensor between the layers and shifts it in place itself; the LLaVA patch, which shifts a private clone per layer,
        # does not).  This key is the kill switch: False = every layer's shift is its own launch, whatever the caller says.
        self.shift_next_in_update = bool(kv_compression_kwargs.get("shift_next_in_update", True))
        self._preshifted = None       # (ids tensor, its version, layer, stream) an update launch has already shifted for
        if self.flush_every_layers < 0:
            raise ValueError("flush_every_layers must be >= 0")
        self._sides: List[_Side] = []
        self._side_rr = 0
        self._pos_layers = 0          # len(position_cache) of the reference (skipped layers are padded with [])
        self.num_evicted_tokens: List[int] = []
        self.keypatches_mask_chunk = None
        self._ws: Dict[str, torch.Tensor] = {}
        self._batch: Optional[_Batch] = None
        self._warned = False

    # ---- diagnostics of the most recent compressed update (the selection may still be deferred: flush first) ----
    @property
    def last_scores(self):
        """Scores of the most recent compressed update; None if it kept its whole chunk without scoring it."""
        b, l = self._last_slot
        self._flush()
        return None if b.keep_all else b.score[l]

    @property
    def last_keep_indices(self):
        b, l = self._last_slot
        self._flush()
        return b.keep_idx[l]

    def memory_footprint(self) -> Dict[str, int]:
        """Device bytes this cache holds right now, by what they are for (not in the reference; bench.py's `memory` block and
        tests/test_memory_gpu.py read it).  The reference's cache is the compressed rows only (longvideo_cache.py:313-318) -
        its transients (the [Hq, L, L] fp32 softmax and its casts, two torch.cat copies of the layer) come and go inside
        every update; here the transients are pre-allocated once per chunk geometry and reused:
          cache_rows        K / V / id rows of the committed tokens (what the reference's lists hold)
          cache_headroom    the rest of the pre-allocated K / V / id buffers: room for the in-flight chunk (the uncompressed
                            tail update() returns to the layer's attention) and for generation, or unused growth
          k_unrotated       per-slot copy of the chunk's un-rotated keys (re-rotated into the cache by the flush)
          score_operands    per-slot q~ (and the fast mode's fp16 k~) + row statistics of the deferred score passes
          score_partials    per-slot column partials, final scores, live-key lists
          selection         kept indices, ids of the pending chunk (old / new), selection scratch, compaction tickets
          staging           kept-row staging of the two-launch eviction (in_place_compaction=False / no reforge)
          deferred_queries  pre-RoPE queries kept alive for a flush that scores them where they lie (prologue_operands="pre_rope")
          worker_scratch    per-update scratch of the stage-by-stage route / worker streams."""
        def nbytes(t):
            return 0 if t is None else t.numel() * t.element_size()

        out = dict.fromkeys(("cache_rows", "cache_headroom", "k_unrotated", "score_operands", "score_partials", "selection",
                             "staging", "deferred_queries", "worker_scratch"), 0)
        for st in self._layers:
            if st.k is not None:
                row = 2 * st.k.shape[1] * st.k.shape[3] * st.k.element_size()
                out["cache_rows"] += row * st.length
                out["cache_headroom"] += nbytes(st.k) + nbytes(st.v) - row * st.length
            if st.pos is not None:
                out["cache_rows"] += 8 * st.pos.shape[0] * st.pos_len
                out["cache_headroom"] += nbytes(st.pos) - 8 * st.pos.shape[0] * st.pos_len
        b = self._batch
        if b is not None:
            out["k_unrotated"] = nbytes(b.k_unrot)
            out["score_operands"] = nbytes(b.score_ws)
            out["score_partials"] = nbytes(b.partials) + nbytes(b.score) + nbytes(b.key_index)
            out["selection"] = (nbytes(b.keep_idx) + nbytes(b.pos_new) + nbytes(b.pos_old) + nbytes(b.sel_ws)
                                + nbytes(b.compact_sync) + nbytes(b.cos_new) + nbytes(b.sin_new))
            out["staging"] = nbytes(b.v_stage) + nbytes(b.k_stage)
            seen = set()
            for t in b.q_keep:
                if t is not None and t.data_ptr() not in seen:
                    seen.add(t.data_ptr())
                    out["deferred_queries"] += nbytes(t)
        out["worker_scratch"] = sum(nbytes(t) for t in self._ws.values()) + sum(nbytes(t) for sd in self._sides for t in sd.ws.values())
        out["total"] = sum(out.values())
        return out

    # ---- list views --------------------------------------------------------------------------
    @property
    def key_cache(self):
        return self._kview

    @key_cache.setter
    def key_cache(self, value):  # the base class assigns [] in __init__
        self._flush()
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
        if st.pending:
            self._flush()
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

    # ---- position ids of the cached tokens (reference :143, :179-215) -----------------------------
    def _pos_view(self, st: _LayerStore):
        v = st.pos[:, :st.pos_len]
        return v.unsqueeze(1) if st.pos_ndim == 3 else v

    @property
    def position_cache(self):
        """Per-layer position ids of the cached tokens (reference :143): a list of [3, 1, n] / [1, n] views
        of the layers' id buffers, [] for skipped layers.  Reading it flushes deferred work."""
        self._flush()
        out = []
        for i in range(self._pos_layers):
            st = self._layers[i] if i < len(self._layers) else None
            out.append([] if st is None or st.pos is None else self._pos_view(st))
        return out

    @position_cache.setter
    def position_cache(self, value):
        self._flush()
        self._pos_layers = len(value)
        for i, t in enumerate(value):
            st = self._store(i)
            if isinstance(t, list) and len(t) == 0:
                st.pos, st.pos_len, st.pos_ndim = None, 0, 0
            else:
                st.pos_ndim = t.ndim
                st.pos = t.reshape(t.shape[0], t.shape[-1]).contiguous()
                st.pos_len = t.shape[-1]

    def _pos_reserve(self, st: _LayerStore, P: int, ndim: int, more: int, device):
        need = st.pos_len + more
        if st.pos is not None and st.pos.shape[1] >= need:
            return
        cap = max(need, 2 * (st.pos.shape[1] if st.pos is not None else 0), 4096, self.reserve_tokens)
        buf = torch.empty((P, cap), dtype=torch.int64, device=device)
        if st.pos is not None and st.pos_len:
            buf[:, :st.pos_len].copy_(st.pos[:, :st.pos_len])
        st.pos, st.pos_ndim = buf, ndim

    # ---- hooks (reference :146-150): after_forward is where deferred compaction is flushed -------
    def before_forward(self, **kwargs):
        pass

    def after_forward(self, **kwargs):
        self._flush()

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
        (longvideo_cache.py:179-209).  The ids are appended to the layer's pre-allocated id buffer."""
        st = self._store(layer_idx)
        if st.pending:
            self._flush()
        n = position_ids.shape[-1]
        P = position_ids.shape[0]
        self._pos_reserve(st, P, position_ids.ndim, n, position_ids.device)
        st.pos[:, st.pos_len:st.pos_len + n].copy_(position_ids.reshape(P, n))
        st.pos_len += n
        self._pos_layers = max(self._pos_layers, layer_idx + 1)
        return self._pos_view(st)

    def get_prev_temporal_idx(self, layer_idx: int):
        """Last temporal id stored for the layer, -1 if none (longvideo_cache.py:211-215)."""
        if layer_idx >= self._pos_layers and not (len(self._layers) > layer_idx and self._layers[layer_idx].pending):
            return -1
        st = self._layers[layer_idx]
        if st.pending:
            self._flush()  # the ids of the previous chunk are still in the batch
        if layer_idx >= self._pos_layers or st.pos is None or st.pos_len == 0:
            return -1
        return st.pos[0, st.pos_len - 1]

    def _shift_latch_check(self):
        """The in-launch id shift (RTK_UPDATE_SHIFT_NEXT) bounds its wait; a wait that ran out shifted NOTHING and latched a
        host-visible word.  Every entry point looks at it (one read of pinned host memory, no device sync) before it
        trusts ids a launch was supposed to have shifted - same policy as p2p.check(): a bounded wait that gives up is an
        error, never a silent continuation."""
        b = self._batch
        if b is not None and b.shift_latch is not None and b.shift_latch[0]:
            self._shift_failed(b)

    def _shift_failed(self, b: "_Batch"):
        n = int(b.shift_latch[0])
        self._preshifted = None
        self.shift_next_in_update = False     # this cache goes back to one shift launch per layer
        try:   # put the words back (the failed launches left residue in the counters, the latch stops every later watcher)
            torch.cuda.synchronize(b.device)
            b.shift_ticket.zero_()
            torch.cuda.synchronize(b.device)
        except Exception:  # noqa: BLE001  (the error below is the one to report)
            pass
        b.shift_latch[0] = 0
        b.shift_stream = None
        raise RuntimeError(
            f"PivotKVCache: the in-launch position-id shift of {n} update launch(es) ran out of its bounded wait and shifted "
            "nothing (the arrival counters were not zero at launch: were update launches of one cache issued on two streams "
            "at once?).  The layers after the first such launch ran on UNSHIFTED temporal ids: the cache contents of the "
            "current video are invalid - rebuild the cache.  The feature is now off for this cache (one rtk_position_shift "
            "launch per layer).")

    def check(self):
        """Synchronise the cache's device and raise if a bounded device-side wait of this cache has run out (the in-launch
        id shift).  The entry points look at the latch on every call without synchronising; call this after the last
        update of a run when nothing else of the cache is called before its contents are used."""
        b = self._batch
        if b is not None and b.device.type == "cuda":
            torch.cuda.synchronize(b.device)
        self._shift_latch_check()

    def shift_temporal_ids_(self, position_ids: torch.Tensor, layer_idx: int):
        """The attention patch's continuity fix (qwen2_vl.py:68-73, llava_onevision.py:68-72) on the device:
        position_ids[0, 0, :] (or [0, :]) += prev + 1 - its first element, in place, where prev is the last
        temporal id cached for the layer.  Same result as the reference's compare-then-shift, without the
        host round trip of the comparison."""
        if not self.pos_embed_reforge:
            return position_ids
        return self._shift_row(position_ids, layer_idx)

    def _shift_row(self, position_ids: torch.Tensor, layer_idx: int):
        if not position_ids.is_cuda:
            nv.require_device(position_ids)
        if position_ids.dtype is not torch.int64 or position_ids.stride(-1) != 1:
            row = position_ids[0, 0] if position_ids.ndim == 3 else position_ids[0]
            prev = self.get_prev_temporal_idx(layer_idx)
            row += prev + 1 - row[0].clone()
            return position_ids
        self._shift_latch_check()   # (before the memo below is trusted)
        done, self._preshifted = self._preshifted, None
        idx = position_ids.get_device()
        if done is not None and done[0]() is position_ids and done[1] is not None and done[1] == _version_of(position_ids) \
                and done[2] == layer_idx and nv.current_device() == idx and done[3] == nv.raw_stream(idx):
            return position_ids   # the previous layer's update launch has shifted this very tensor for this layer
        prev_ptr = None
        if len(self._layers) > layer_idx:
            c = self._layers[layer_idx].c
            if c.pending:
                self._flush()
            if layer_idx < self._pos_layers and c.pos and c.pos_len:
                prev_ptr = c.pos + 8 * (c.pos_len - 1)
        # the temporal row ([0, 0, :] of [3, 1, n] ids, [0, :] of [1, n] ids) starts at the tensor's first element
        if nv.current_device() == idx:
            rc = nv.lib.rtk_position_shift(position_ids.data_ptr(), position_ids.shape[-1], prev_ptr, nv.raw_stream(idx))
        else:
            with torch.cuda.device(idx):
                rc = nv.lib.rtk_position_shift(position_ids.data_ptr(), position_ids.shape[-1], prev_ptr,
                                               nv.raw_stream(idx))
        nv.check(rc, "rtk_position_shift")
        return position_ids

    # ---- storage -------------------------------------------------------------------------------
    def _store(self, layer_idx: int) -> _LayerStore:
        while len(self._layers) <= layer_idx:
            self._layers.append(_LayerStore())
        return self._layers[layer_idx]

    def reserve(self, layer_idx: int, tokens: int, like: torch.Tensor):
        """Make room for `tokens` more rows after the committed length of the layer."""
        st = self._store(layer_idx)
        need = st.length + tokens
        if st.k is not None and st.k.shape[2] >= need and st.k.is_contiguous() and st.v.is_contiguous() \
                and st.v.shape[2] == st.k.shape[2]:
            return st
        cap = max(need, 2 * (st.k.shape[2] if st.k is not None else 0), 1024, self.reserve_tokens)
        shape = (1, like.shape[1], cap, like.shape[3])
        nk = torch.empty(shape, dtype=like.dtype, device=like.device)
        nvv = torch.empty(shape, dtype=like.dtype, device=like.device)
        if st.k is not None and st.length:
            nk[:, :, :st.length].copy_(st.k[:, :, :st.length])
            nvv[:, :, :st.length].copy_(st.v[:, :, :st.length])
        st.k, st.v = nk, nvv
        return st

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

    _STATIC_ROPE_TYPES = ("default", "linear", "yarn", "llama3", "mrope")

    def _rotary(self, rotary_emb_fn, device) -> Optional[_Rotary]:
        """The native-RoPE view of a rotary module, or None when the module has to be called (reference :249, :298).
        Memoised per module object; modules with equal inv_freq / attention_scaling (HF: one per layer) share one
        entry, so a chunk's layers stay in one batch."""
        hit = self._rotaries.get(id(rotary_emb_fn))
        if hit is not None and hit[0] is rotary_emb_fn and (hit[1] is None or hit[1].device == device) \
                and hit[2] == _inv_stamp(rotary_emb_fn):
            return hit[1]
        entry = None
        inv = getattr(rotary_emb_fn, "inv_freq", None)
        if (self.native_rope and isinstance(inv, torch.Tensor) and hasattr(rotary_emb_fn, "attention_scaling")
                and getattr(rotary_emb_fn, "rope_type", "default") in self._STATIC_ROPE_TYPES):
            inv = inv.detach().to(device=device, dtype=torch.float32).contiguous()
            scaling = float(rotary_emb_fn.attention_scaling)
            for _, other, _ in self._rotaries.values():
                if (other is not None and other.device == device and other.scaling == scaling
                        and other.inv.shape == inv.shape and torch.equal(other.inv, inv)):
                    entry = other
                    break
            if entry is None:
                entry = _Rotary(inv, scaling, device)
        # (the strong reference pins the id; the stamp notices a module whose inv_freq was replaced or written in place)
        self._rotaries[id(rotary_emb_fn)] = (rotary_emb_fn, entry, _inv_stamp(rotary_emb_fn))
        return entry

    def _rope_tables(self, cos_t, sin_t, rotary_emb_fn, x_like, pos2d, pos_ld, ndim, mrope_section, n, D):
        """fp32 [n, D] cos/sin tables of the ids pos2d [P, n] (row stride pos_ld), section-merged
        (reference :249 / :298 + :68-74), written into cos_t / sin_t."""
        dev = x_like.device
        P = pos2d.shape[0]
        sec = (C.c_int * len(mrope_section))(*mrope_section) if mrope_section else None
        nsec = len(mrope_section) if mrope_section else 0
        s = nv.stream()
        rot = self._rotary(rotary_emb_fn, dev)
        if rot is not None:
            nv.check(nv.lib.rtk_rope_table(nv.ptr(pos2d), pos_ld, P, n, nv.ptr(rot.inv), D, rot.scaling, sec, nsec,
                                           nv.round_mode(x_like.dtype), nv.ptr(cos_t), nv.ptr(sin_t), s),
                     "rtk_rope_table")
            return
        ids = pos2d.unsqueeze(1) if ndim == 3 else pos2d
        cos, sin = rotary_emb_fn(x_like, ids)  # third-party module, exactly as the reference calls it
        cos = cos.reshape(P, n, D)
        sin = sin.reshape(P, n, D)
        if not cos.is_contiguous():
            cos = cos.contiguous()
        if not sin.is_contiguous():
            sin = sin.contiguous()
        nv.check(nv.lib.rtk_rope_merge(nv.ptr(cos), nv.ptr(sin), P, n, D, nv.dtype_code(cos), sec, nsec, nv.ptr(cos_t),
                                       nv.ptr(sin_t), s), "rtk_rope_merge")

    # ---- deferred eviction -----------------------------------------------------------------------
    def _get_batch(self, layer_idx, Hq, Hkv, L, D, keep, P, dtype, device) -> _Batch:
        refround = self.score_rounding == "reference" and dtype in (torch.bfloat16, torch.float16)
        if refround and D != 128:
            raise NotImplementedError("score_rounding='reference' needs head_dim 128")
        # "fast" (opt in): bf16 chunks of head_dim 128 on the fp16 matrix instruction; every other shape scores as usual
        fast = self.score_rounding == "fast" and dtype == torch.bfloat16 and D == 128
        keep_all = keep == L and not self.score_when_keeping_all
        defer = bool(self.defer_rerotation and self.pos_embed_reforge)
        key = (Hq, Hkv, L, D, keep, P, bool(self.pos_embed_reforge), dtype, device, refround, fast, keep_all, defer)
        b = self._batch
        wrap = self.flush_every_layers > 0
        if b is not None and b.key == key and (layer_idx < b.slots or b.wrap) and b.wrap == wrap:
            return b
        self._flush()
        if wrap:
            slots = min(self.flush_every_layers, max(int(self.num_hidden_layers), 1))
        else:
            slots = max(int(self.num_hidden_layers), layer_idx + 1, b.slots if b is not None and b.key == key else 0)
        self._batch = None  # release the old buffers before allocating the new ones
        self._batch = _Batch(key, slots, Hq, Hkv, L, D, keep, P, bool(self.pos_embed_reforge), dtype, device, refround, fast,
                             keep_all, self.skip_masked_columns, self.in_place_compaction)
        self._batch.wrap = wrap
        self._batch.defer = defer
        self._batch.c.defer_rot = int(defer)
        return self._batch

    def _claim_slot(self, b: _Batch, layer_idx: int) -> int:
        """The layer's slot in the batch; with flush_every_layers the pending units are flushed first when that slot (or
        a later one: the slots of a flush ascend with its layers) is still taken."""
        sl = b.slot(layer_idx)
        if b.wrap and b.pending and (sl <= b.slot(b.pending[-1]) or layer_idx <= b.pending[-1]):
            self._flush()
        return sl

    def _flush(self):
        """Evict every pending (layer, chunk) unit (reference :260-318 for all layers of the chunk): the score passes,
        the selection, one batched gather / re-rotate launch and one batched placement launch."""
        b = self._batch
        if b is not None and b.shift_latch is not None and b.shift_latch[0]:
            self._shift_failed(b)
        if b is None or not b.pending:
            return
        if b.c_pending == len(b.pending) and self._flush_c(b):
            return
        self._flush_general(b)

    def _flush_c(self, b: _Batch) -> bool:
        """rtk_pivotkv_flush: the whole chain in one call (units appended by rtk_pivotkv_update only)."""
        layers = b.pending
        if any(layers[i] >= layers[i + 1] for i in range(len(layers) - 1)):
            layers = sorted(set(layers))
        L_ = self._layers
        n = len(layers)
        if b.reforge and b.P:
            for l in layers:
                st = L_[l]
                if st.c.pos_len + b.keep > st.c.pos_cap:
                    self._pos_reserve(st, b.P, 3 if b.P == 3 else 2, b.keep, b.device)
                elif st.pos_ndim == 0:
                    st.pos_ndim = 3 if b.P == 3 else 2
        states = (C.c_void_p * n)(*[L_[l].cref for l in layers])
        slots = (C.c_int32 * n)(*[b.slot(l) for l in layers])
        idx = b.dev_index
        self._order_compaction(b)
        if nv.current_device() == idx:
            rc = nv.lib.rtk_pivotkv_flush(b.cref, states, slots, n, nv.raw_stream(idx))
        else:
            with torch.cuda.device(b.device):
                rc = nv.lib.rtk_pivotkv_flush(b.cref, states, slots, n, nv.raw_stream(idx))
        if rc == nv.RTK_EUNSUPPORTED:
            return False
        if rc:
            self._reset_compaction(b)
        nv.check(rc, "rtk_pivotkv_flush")
        b.pending = []
        b.c_pending = 0
        b.masks.clear()
        b.scored.clear()
        b.selected.clear()
        b.shift_ids = None
        for l in layers:
            b.q_keep[b.slot(l)] = None
        if b.reforge and b.P:
            self._pos_layers = max(self._pos_layers, layers[-1] + 1)
        return True

    def _flush_general(self, b: _Batch):
        """The same chain launched stage by stage: worker streams, small chunks (selection inside update), rotary
        modules that have to be called for the tables of the new ids."""
        layers, b.pending = b.pending, []
        b.c_pending = 0
        # slot of a layer: the layer itself, or (flush_every_layers) layer mod slots - within one flush slots ascend with the
        # layers, so the offset is one number
        off = b.slot(layers[0]) - layers[0]
        assert all(b.slot(l) == l + off for l in layers)
        if b.shift_ids is not None:   # pre-RoPE units: the caller's ids tensor takes the last layer's shift now
            ids, b.shift_ids = b.shift_ids, None
            b.c.shift_row = None
            self._shift_row(ids, layers[-1])
        if not b.keep_all:
            b.ensure_scoring()
        keep, D, Hkv, P = b.keep, b.D, b.Hkv, b.P
        if b.keep_all:   # units of the one-call path carry their ids in pos_old only: ids x 1.0 = the ids (:288-292)
            for l in layers:
                if l not in b.selected:
                    if P:
                        b.pos_new[:, l + off].copy_(b.pos_old[l + off])
                    b.selected.add(l)
                    b.scored.add(l)
        es = 4 if b.dtype == torch.float32 else 2
        dt = nv.RTK_BF16 if b.dtype == torch.bfloat16 else (nv.RTK_F16 if b.dtype == torch.float16 else nv.RTK_F32)
        with torch.cuda.device(b.device):
            main = torch.cuda.current_stream()
            for l in layers:
                st = self._layers[l]
                if st.pending_event is not None:  # scored on a worker stream
                    main.wait_event(st.pending_event)
                    st.pending_event = None
            unscored = sorted(l for l in layers if l not in b.scored)
            i = 0
            def qkey(l):   # queries scored where they lie (prologue route) carry their strides; packed ones None
                t = b.q_keep[l + off]
                return None if t is None else t.stride()

            while i < len(unscored):  # the matrix passes of every run of consecutive slots whose queries live alike
                j = i                   # in one launch per kernel (:260-268)
                while j + 1 < len(unscored) and unscored[j + 1] == unscored[j] + 1 and qkey(unscored[j + 1]) == qkey(unscored[i]):
                    j += 1
                l0, n = unscored[i], j - i + 1
                mptr = [b.masks.get(l) for l in range(l0, l0 + n)]
                l0 += off     # from here on: the run's first SLOT
                km = (C.c_void_p * n)(*[m.data_ptr() if m is not None else None for m in mptr]) \
                    if self.skip_masked_columns and any(m is not None for m in mptr) else None
                qk = [b.q_keep[l] for l in range(l0, l0 + n)]
                qu, qsh, qsl = None, 0, 0
                if qk[0] is not None:   # units of the prologue route whose queries are scored in place
                    qu = (C.c_void_p * n)(*[t.data_ptr() for t in qk])
                    qsh, qsl = qk[0].stride(1), qk[0].stride(2)
                nv.check(nv.lib.rtk_pivotkv_score_passes_batched_q(
                    C.c_void_p(b.score_ws_base + l0 * b.ws_stride), b.ws_stride,
                    nv.ptr(b.k_unrot[l0]) if b.reforge else None, b.L * D * Hkv * es,
                    nv.ptr(b.partials[l0]), b.part_floats, n, b.Hq, Hkv, b.L, D, b.score_dt,
                    km, nv.ptr(b.key_index[l0]) if km is not None else None, qu, qsh, qsl, nv.stream()),
                    "rtk_pivotkv_score_passes_batched")
                i = j + 1
            b.scored.clear()
            todo = [l for l in layers if l not in b.selected]
            if todo:  # mask override + top-k + id gather / rescale of every layer of the chunk (reference :269-295)
                su = (nv.SelectUnit * len(todo))()
                for i, l in enumerate(todo):
                    u = su[i]
                    sl = l + off
                    u.partial = b.partials[sl].data_ptr()
                    u.score = b.score[sl].data_ptr()
                    m = b.masks.get(l)
                    u.mask = m.data_ptr() if m is not None else None
                    u.pos = b.pos_old[sl].data_ptr() if P else None
                    u.keep_idx = b.keep_idx[sl].data_ptr()
                    u.rank = None
                    u.pos_out = (b.pos_new.data_ptr() + sl * keep * 8) if P else None
                    u.workspace = b.sel_ws[sl].data_ptr()
                nv.check(nv.lib.rtk_pivotkv_select_batched(su, len(todo), Hkv, b.rs_n.value, b.Hq // Hkv, b.L, keep, P,
                                                           int(b.reforge), b.slots * keep, b.score_dt, nv.stream()),
                         "rtk_pivotkv_select_batched")
            b.selected.clear()
            b.masks.clear()
            lo, hi = min(layers) + off, max(layers) + off      # slots
            # reforge: K is re-rotated at the NEW ids (reference :297-306).  With the native RoPE the eviction kernel
            # computes their cos/sin itself; a third-party rotary module is called once for every pending slot and its
            # section-merged fp32 tables are handed over.
            defer = bool(getattr(b, "defer", False))
            rot = self._rotary(b.rotary_emb_fn, b.device) if (b.reforge and P and not defer) else None
            rope_in_kernel = rot is not None
            if b.reforge and not rope_in_kernel and not defer:
                b.ensure_tables()
                n = (hi - lo + 1) * keep
                pos2d, ld = b.pos_new[:, lo:hi + 1].reshape(P, n), n  # a copy when the slot range is partial
                self._rope_tables(b.cos_new[lo * keep:], b.sin_new[lo * keep:], b.rotary_emb_fn, b.x_like, pos2d, ld,
                                  3 if P == 3 else 2, b.mrope_section, n, D)
            if b.compact_sync is not None and not b.keep_all and (rope_in_kernel or defer or not b.reforge):
                self._compact(b, layers, rot, defer, dt, es)
                layers_done, layers = layers, []
            else:
                layers_done = layers
                if not b.keep_all:
                    b.ensure_staging()
            units = (nv.EvictUnit * max(1, len(layers)))()
            places = (nv.PlaceUnit * max(1, len(layers) * (1 if b.reforge else 2)))()
            nc = 0
            for i, l in enumerate(layers):
                st = self._layers[l]
                sl = l + off
                cap = st.k.shape[2]
                tail = st.length * D * es
                u = units[i]
                if b.reforge:
                    u.k_src, u.k_src_stride_h = b.k_unrot[sl].data_ptr(), b.L * D
                    if rope_in_kernel or defer:
                        u.cos_new = u.sin_new = None
                    else:
                        u.cos_new = b.cos_new.data_ptr() + sl * keep * D * 4
                        u.sin_new = b.sin_new.data_ptr() + sl * keep * D * 4
                    u.k_dst, u.k_dst_stride_h = st.k.data_ptr() + tail, cap * D  # straight into the cache
                elif b.keep_all:   # every row already sits where the append put it
                    u.k_src, u.k_src_stride_h = st.k.data_ptr() + tail, cap * D
                    u.cos_new = u.sin_new = u.k_dst = None
                else:
                    u.k_src, u.k_src_stride_h = st.k.data_ptr() + tail, cap * D
                    u.cos_new = u.sin_new = None
                    u.k_dst, u.k_dst_stride_h = b.k_stage[sl].data_ptr(), keep * D
                    places[nc].stage, places[nc].stage_stride_h_bytes = b.k_stage[sl].data_ptr(), keep * D * es
                    places[nc].tail, places[nc].tail_stride_h_bytes = st.k.data_ptr() + tail, cap * D * es
                    places[nc].keep_idx = b.keep_idx[sl].data_ptr()
                    nc += 1
                u.v_src, u.v_src_stride_h = st.v.data_ptr() + tail, cap * D
                if b.keep_all:
                    u.v_dst = None
                else:
                    u.v_dst, u.v_dst_stride_h = b.v_stage[sl].data_ptr(), keep * D
                    places[nc].stage, places[nc].stage_stride_h_bytes = b.v_stage[sl].data_ptr(), keep * D * es
                    places[nc].tail, places[nc].tail_stride_h_bytes = st.v.data_ptr() + tail, cap * D * es
                    places[nc].keep_idx = b.keep_idx[sl].data_ptr()
                    nc += 1
                u.keep_idx = b.keep_idx[sl].data_ptr()
                if b.reforge and P:  # bookkeeping (reference :308-309)
                    self._pos_reserve(st, P, 3 if P == 3 else 2, keep, b.device)
                    u.pos_src, u.pos_src_stride = b.pos_new.data_ptr() + sl * keep * 8, b.slots * keep
                    u.pos_dst, u.pos_dst_stride = st.pos.data_ptr() + st.pos_len * 8, st.pos.shape[1]
                else:
                    u.pos_src = u.pos_dst = None
            s = nv.stream()
            if not layers:
                pass
            elif rope_in_kernel:
                sec = (C.c_int * len(b.mrope_section))(*b.mrope_section) if b.mrope_section else None
                nv.check(nv.lib.rtk_pivotkv_evict_batched_rope(units, len(layers), Hkv, D, keep, P, dt, nv.ptr(rot.inv),
                                                               rot.scaling, sec,
                                                               len(b.mrope_section) if b.mrope_section else 0,
                                                               nv.round_mode(b.x_like.dtype), 1, s),
                         "rtk_pivotkv_evict_batched_rope")
            elif b.reforge or not b.keep_all:   # (deferred re-rotation: every kept un-rotated K row is copied, bit 1)
                nv.check(nv.lib.rtk_pivotkv_evict_batched(units, len(layers), Hkv, D, keep, P if b.reforge else 0, dt,
                                                          3 if defer else 1, s), "rtk_pivotkv_evict_batched")
            # kept rows -> head of the tail: in place, except the ~ratio of them whose source lies inside the destination
            # range (parked in the staging rows by the launch above) - reference :313-318 without a full second copy
            if nc:
                nv.check(nv.lib.rtk_pivotkv_place_batched(places, nc, Hkv, keep, D, dt, s), "rtk_pivotkv_place_batched")
            layers = layers_done
        for l in layers:
            st = self._layers[l]
            b.q_keep[l + off] = None
            b.q_units[l + off] = None
            st.length += keep
            st.pending = 0
            st.pending_keep = 0
            st.c.mask = None
            if b.reforge and P:
                st.pos_len += keep
                self._pos_layers = max(self._pos_layers, l + 1)

    def _order_compaction(self, b: _Batch):
        """The in-place compaction's tickets and flags (batch.compact_sync) serve ONE launch at a time: flushes of a batch
        are ordered on one stream.  A flush that arrives on another stream than the batch's previous one first waits for
        the device (rare: the caller changed its current stream between two chunks)."""
        if b.compact_sync is None:
            return
        cur = nv.raw_stream(b.dev_index)
        if b.sync_stream is not None and b.sync_stream != cur:
            torch.cuda.synchronize(b.device)
        b.sync_stream = cur

    def _reset_compaction(self, b: _Batch):
        """After a failed flush: tickets / flags back to zero (a launch that completes leaves them zeroed itself)."""
        if b.compact_sync is not None:
            try:
                b.compact_sync.zero_()
            except Exception:  # noqa: BLE001  (a device-side abort leaves the context unusable; the first error is what is raised)
                pass

    def _compact(self, b: _Batch, layers, rot, defer, dt, es):
        """The eviction scan of the pending layers as one in-place launch (rtk_pivotkv_compact_batched; reference
        :278-318): kept K re-rotated at the new ids (or copied un-rotated when the rotation is deferred; compacted in
        place without reforge), V compacted inside the tail, ids to the position cache."""
        keep, D, Hkv, P = b.keep, b.D, b.Hkv, b.P
        units = (nv.CompactUnit * len(layers))()
        for i, l in enumerate(layers):
            st = self._layers[l]
            sl = b.slot(l)
            cap = st.k.shape[2]
            tail = st.length * D * es
            u = units[i]
            if b.reforge:
                u.k_src, u.k_src_stride_h = b.k_unrot[sl].data_ptr(), b.L * D
            else:
                u.k_src, u.k_src_stride_h = None, 0
            u.k_tail, u.k_tail_stride_h = st.k.data_ptr() + tail, cap * D
            u.v_tail, u.v_tail_stride_h = st.v.data_ptr() + tail, cap * D
            u.keep_idx = b.keep_idx[sl].data_ptr()
            if b.reforge and P:  # bookkeeping (reference :308-309)
                self._pos_reserve(st, P, 3 if P == 3 else 2, keep, b.device)
                u.pos_src, u.pos_src_stride = b.pos_new.data_ptr() + sl * keep * 8, b.slots * keep
                u.pos_dst, u.pos_dst_stride = st.pos.data_ptr() + st.pos_len * 8, st.pos.shape[1]
            else:
                u.pos_src = u.pos_dst = None
        mode = nv.COMPACT_K_INPLACE if not b.reforge else (nv.COMPACT_K_COPY if defer else nv.COMPACT_K_ROTATE)
        sec = (C.c_int * len(b.mrope_section))(*b.mrope_section) if (b.mrope_section and mode == nv.COMPACT_K_ROTATE) else None
        self._order_compaction(b)
        nv.check(nv.lib.rtk_pivotkv_compact_batched(
            units, len(layers), Hkv, D, keep, P if b.reforge else 0, dt, mode,
            nv.ptr(rot.inv) if mode == nv.COMPACT_K_ROTATE else None, rot.scaling if mode == nv.COMPACT_K_ROTATE else 1.0,
            sec, len(sec) if sec is not None else 0, nv.round_mode(b.x_like.dtype) if b.x_like is not None else nv.round_mode(b.dtype),
            b.c.compact_sync, b.c.compact_sync_ints, nv.stream()), "rtk_pivotkv_compact_batched")

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

        A chunk whose geometry matches the current batch goes through ONE library call (rtk_pivotkv_update: argument
        blocks bound once per batch and per layer); everything else - the first update of a geometry, text / decode
        appends, worker streams, small chunks, rotary modules that must be called - through `_update_general`.

        Like the reference, `update` pops position_ids / query_states / rotary_emb / mrope_section from `cache_kwargs` and
        leaves every tensor it is handed untouched - with ONE opt-in exception the reference does not have:
        cache_kwargs["shift_next_position_ids"] = True lets the launch apply the NEXT layer's continuity shift
        (qwen2_vl.py:68-73) to `position_ids` in place, i.e. after update(l) the caller's ids tensor may already carry layer
        l + 1's temporal offset.  Only a caller that shares one ids tensor between its layers and shifts it in place itself
        (the Qwen2-VL attention patch: `shift_temporal_ids_` then finds the shift done) should set it.
        """
        b = self._batch
        if b is not None and b.shift_latch is not None and b.shift_latch[0]:
            self._shift_failed(b)
        if b is not None and b.c_capable and cache_kwargs is not None and self.kvcache_compression \
                and self.overlap_streams <= 0 and self.one_call_update:
            out = self._update_c(b, key_states, value_states, layer_idx, cache_kwargs, None)
            if out is not None:
                return out
        return self._update_general(key_states, value_states, layer_idx, cache_kwargs)

    def _update_c(self, b: _Batch, key_states, value_states, layer_idx, ck, q0, shift_ids_in_place=True, q_out=None):
        """One rtk_pivotkv_update call.  q0 None: `ck["query_states"]` / key_states are rotated (the reference's
        protocol); else q0 / key_states are the pre-RoPE projections and the rotated queries are written over q0.
        Returns None - having changed nothing - when the call does not fit the batch."""
        q = ck.get("query_states") if q0 is None else q0
        pos = ck.get("position_ids")
        if q is None or pos is None or (layer_idx >= b.slots and not b.wrap) or layer_idx >= len(self._layers):
            return None
        rot_fn = ck.get("rotary_emb")
        hit = self._rotaries.get(id(rot_fn))
        if hit is None or hit[0] is not rot_fn or hit[1] is None or hit[1] is not b.rot \
                or ck.get("mrope_section") != b.mrope_section or hit[2] != _inv_stamp(rot_fn):
            return None
        dt = b.dtype
        if q.shape != b.qshape or key_states.shape != b.kshape or value_states.shape != b.kshape \
                or q.dtype is not dt or key_states.dtype is not dt or value_states.dtype is not dt:
            return None
        idx = b.dev_index
        if not (q.is_cuda and q.get_device() == idx and key_states.get_device() == idx
                and value_states.get_device() == idx and pos.is_cuda and pos.get_device() == idx
                and nv.current_device() == idx):
            return None
        L, P = b.L, b.P
        if pos.dtype is not torch.int64 or pos.shape[-1] != L or pos.shape[0] != P or pos.ndim != (3 if P == 3 else 2) \
                or pos.stride(-1) != 1 or (pos.ndim == 3 and (pos.shape[1] != 1 or pos.stride(0) < L)):
            return None   # (ids whose rows alias - `.expand(3, ..)` - take the stage-by-stage route)
        mask = self.keypatches_mask_chunk
        if mask is None:
            mptr = None
        elif mask is b.mask_obj:
            mptr = b.mask_ptr
        else:
            if not (mask.is_cuda and mask.dtype is torch.bool and mask.numel() == L and mask.is_contiguous()
                    and mask.get_device() == idx):
                return None
            b.mask_obj, b.mask_ptr = mask, mask.data_ptr()
            mptr = b.mask_ptr
        st = self._layers[layer_idx]
        c = st.c
        if c.pending:
            self._flush()
        P0 = c.length
        if P0 + L > c.cap:   # (cap is 0 for buffers the library may not use)
            return None
        pre = q0 is not None
        roundtrip = pre and self.prologue_operands == "reference"
        mode = (2 if roundtrip else 1) if pre else 0
        if b.pending and b.c.pre_rope != mode:
            self._flush()
        slot = self._claim_slot(b, layer_idx)
        b.c.pre_rope = mode   # what the units of this batch's next flush were appended from
        qs, ks, vs = q.stride(), key_states.stride(), value_states.stride()
        if qs[3] != 1 or ks[3] != 1 or vs[3] != 1:
            return None
        io = b.io
        stream = nv.raw_stream(idx)
        io.q, io.q_stride_h, io.q_stride_l = q.data_ptr(), qs[1], qs[2]
        io.k, io.k_stride_h, io.k_stride_l = key_states.data_ptr(), ks[1], ks[2]
        io.v, io.v_stride_h, io.v_stride_l = value_states.data_ptr(), vs[1], vs[2]
        io.pos, io.pos_stride = pos.data_ptr(), pos.stride(0)
        q_in_place = False
        if pre:
            rt = nv.RTK_UPDATE_ROUNDTRIP if roundtrip else 0
            if q_out is None or q_out is q:
                io.q_rot, io.qr_stride_h, io.qr_stride_l, io.flags = io.q, qs[1], qs[2], nv.RTK_UPDATE_PRE_ROPE | rt
            else:
                io.q_rot, io.qr_stride_h, io.qr_stride_l = q_out.data_ptr(), q_out.stride(1), q_out.stride(2)
                # the rotated queries go elsewhere, so q0 survives: the batched passes score it where it lies
                q_in_place = self._scores_q0_in_place(b)
                io.flags = nv.RTK_UPDATE_PRE_ROPE | rt | (nv.RTK_UPDATE_Q_IN_PLACE if q_in_place else 0)
        else:
            io.q_rot, io.flags = None, 0
            # the NEXT layer's continuity shift rides in this launch (its `shift_temporal_ids_` then finds it done) - for a
            # caller that asked for it (it writes the caller's ids: the attention patch's own in-place shift, one layer
            # early), and only where the library cannot decline AFTER the launch (batched passes / keep-all batches)
            nxt = False
            if self.shift_next_in_update and ck.get("shift_next_position_ids") and (b.batched_passes or b.keep_all) \
                    and b.shift_latch is not None:
                nxt = self._next_layer_prev(layer_idx, idx)
            if nxt is not False:
                io.flags, io.next_prev = nv.RTK_UPDATE_SHIFT_NEXT, nxt
                # the arrival counters serve ONE launch at a time: launches that share them are ordered on one stream.  A
                # caller that switched its current stream between two layers first waits for the device (rare)
                if b.shift_stream is not None and b.shift_stream != stream:
                    torch.cuda.synchronize(b.device)
                b.shift_stream = stream
        self._preshifted = None
        c.mask = mptr
        rc = nv.lib.rtk_pivotkv_update(b.cref, st.cref, slot, b.ioref, stream)
        if rc:
            c.mask = None
            if rc == nv.RTK_EUNSUPPORTED:
                # declined BEFORE anything was launched - also with the in-launch shift: the library only accepts that flag
                # where nothing can decline after the prepare launch (include/retake_hip.h, RTK_UPDATE_SHIFT_NEXT)
                return None
            nv.check(rc, "rtk_pivotkv_update")
        if io.flags & nv.RTK_UPDATE_SHIFT_NEXT:
            self._preshifted = (weakref.ref(pos), _version_of(pos), layer_idx + 1, stream)
        if not self._warned:  # the reference's logger.warning_once (:232)
            self._warned = True
            _warn_once("Enable PivotKVCache compression: length after compression %.2f" % (self.compression_ratio))
        if not pre:  # the reference's cache_kwargs protocol (:235, :241-243)
            ck.pop("position_ids", None)
            ck.pop("query_states", None)
            ck.pop("rotary_emb", None)
            ck.pop("mrope_section", None)
            ck.pop("shift_next_position_ids", None)   # (the build's own key)
        else:
            if shift_ids_in_place:     # the flush shifts the caller's ids in place (qwen2_vl.py:73)
                b.shift_ids = pos
                b.c.shift_row = io.pos
        nev = self.num_evicted_tokens
        if len(nev) > layer_idx:
            nev[layer_idx] += L - b.keep   # reference :310
        else:
            self.update_num_evicted_tokens(L - b.keep, layer_idx)
        if mask is not None:
            b.masks[layer_idx] = mask
        b.q_keep[slot] = q if q_in_place else None
        if b.keep_all:
            b.scored.add(layer_idx)
        elif not b.batched_passes:
            b.scored.add(layer_idx)
        b.pending.append(layer_idx)
        b.c_pending += 1
        self._last_slot = (b, slot)
        n = P0 + L
        return st._k.narrow(2, 0, n), st._v.narrow(2, 0, n)

    def _next_layer_prev(self, layer_idx: int, dev_index: int):
        """Address of the last temporal id cached for layer_idx + 1 (None: nothing cached, the rule's prev = -1), or
        False when this launch must not shift for it: there is no such layer, its previous chunk is still pending, or its
        ids live on another device (a model spread over GPUs by device_map: that layer gets its own launch there)."""
        nxt = layer_idx + 1
        if nxt >= int(self.num_hidden_layers):
            return False
        if len(self._layers) > nxt:
            st = self._layers[nxt]
            c = st.c
            if c.pending:
                return False
            if nxt < self._pos_layers and c.pos and c.pos_len:
                if st.pos is None or st.pos.get_device() != dev_index:
                    return False
                return c.pos + 8 * (c.pos_len - 1)
        return None

    def _scores_q0_in_place(self, b: _Batch) -> bool:
        """Can the chunk-batched passes read the pre-RoPE queries where they lie?  Only when q~ IS q0."""
        return bool(b.batched_passes and not b.fast and not b.keep_all and self.score_queries_in_place
                    and self.prologue_operands == "pre_rope")

    def update_pre_rope(self, query_states, key_states, value_states, layer_idx, position_ids, rotary_emb,
                        mrope_section=None, shift_ids_in_place=True, query_out=None):
        """The attention patch's whole prologue as ONE kernel (not in the reference: there it is the continuity shift,
        the rotary module, apply_multimodal_rotary_pos_emb and PivotKVCache.update, qwen2_vl.py:68-86 + :217-259).
        query_states [1, Hq, L, D], key_states / value_states [1, Hkv, L, D] are the PRE-RoPE projections of a video
        chunk.  Returns (rotated queries - written over `query_states` unless `query_out` says otherwise -, keys, values)
        with keys / values as `update` returns them, or None - nothing touched - when this call has to take the eager
        route (text segments, chunks below 512 tokens, rotary modules that must be called, worker streams,
        score_rounding="reference" combined with prologue_operands="pre_rope").
        What the deferred score passes and the re-rotation are handed is `prologue_operands`: "reference" (default) - the
        un-rotation of the rotated rows in the model dtype, the reference's own operands (longvideo_cache.py:76-78,
        :248-259), so scores / kept sets / kept keys follow the reference's bf16 run bit for bit; "pre_rope" - the
        projections themselves.
        shift_ids_in_place: the Qwen2-VL patch shifts the ids tensor it was handed (qwen2_vl.py:73; done here by the
        chunk's flush, with the last layer's rule - what the reference's layer loop leaves behind); the LLaVA patch
        shifts a private clone (llava_onevision.py:76-88), i.e. leaves the caller's tensor alone.
        query_out: where the rotated queries go (same shape and dtype, head_dim contiguous); it is the first element of
        the returned tuple.  None: the library's pick - with prologue_operands="pre_rope" a fresh tensor when the
        chunk-batched score passes can then read `query_states` where it lies (no copy of the queries is made;
        `query_states` must stay unmodified until the chunk's flush, and is kept alive by the cache), else over
        `query_states`; pass `query_states` itself to force the in-place rotation."""
        self._shift_latch_check()
        if not (self.kvcache_compression and self.pos_embed_reforge and self.one_call_update) or self.overlap_streams > 0 \
                or position_ids is None or not key_states.is_cuda or key_states.shape[0] != 1 \
                or (torch.is_grad_enabled() and query_states.requires_grad):
            return None
        L = key_states.shape[2]
        b = self._batch
        hit = self._rotaries.get(id(rotary_emb))
        if b is None or b.L != L or not b.c_capable or hit is None or hit[0] is not rotary_emb or hit[1] is None \
                or hit[1] is not b.rot or hit[2] != _inv_stamp(rotary_emb) \
                or (layer_idx >= b.slots and not b.wrap) or layer_idx >= len(self._layers) \
                or self._layers[layer_idx].c.length + L > self._layers[layer_idx].c.cap:
            # not the steady state: find / build the batch of this geometry, bind the rotary, make room
            dev = key_states.device
            rot = self._rotary(rotary_emb, dev)
            keep_len = max(1, int(self.compression_ratio * L))
            if rot is None or L < 512 or keep_len > L or (self.score_rounding == "reference"
                                                         and self.prologue_operands != "reference"
                                                         and key_states.dtype in (torch.bfloat16, torch.float16)):
                return None   # (the reference's rounding chain scores the reference's operands)
            Hq, D = query_states.shape[1], query_states.shape[3]
            b = self._get_batch(layer_idx, Hq, key_states.shape[1], L, D, keep_len, 3 if position_ids.ndim == 3 else 1,
                                key_states.dtype, dev)
            if not b.c_capable:
                return None
            self._bind_rotary(b, rotary_emb, mrope_section, rot)
            b.x_like = value_states[:, :, :1]
            self.reserve(layer_idx, L, key_states)
        ck = {"position_ids": position_ids, "rotary_emb": rotary_emb, "mrope_section": mrope_section}
        if query_out is None:
            # the library's pick: a fresh tensor whenever the queries can then be scored where they lie (the chunk-batched
            # passes read q0 itself - no packed copy, a quarter of the kernel's traffic), else over `query_states`
            if self._scores_q0_in_place(b):
                query_out = torch.empty_like(query_states)
        elif query_out is not query_states and (query_out.shape != query_states.shape
                                                or query_out.dtype is not query_states.dtype
                                                or query_out.device != query_states.device or query_out.stride(-1) != 1):
            raise ValueError("query_out must match query_states in shape, dtype and device, with a contiguous head_dim")
        out = self._update_c(b, key_states, value_states, layer_idx, ck, query_states, shift_ids_in_place, query_out)
        if out is None:
            return None
        return (query_states if query_out is None else query_out), out[0], out[1]

    def append_pre_rope(self, query_states, key_states, value_states, layer_idx, position_ids, rotary_emb,
                        mrope_section=None, shift_ids_in_place=True):
        """The attention patch's prologue for a segment that is NOT compressed - text prefill, decode (qwen2_vl.py:68-86
        + the cache's else-branch :319-321) - as one kernel (rtk_pivotkv_append_rope): continuity shift, rotary tables,
        RoPE of q (in place) and k, append of the rotated k and of v, the shifted ids appended to the layer's position
        cache; the caller's ids are shifted in place afterwards when `shift_ids_in_place` (Qwen2-VL; LLaVA shifts a
        clone).  Replaces ~25 eager launches per layer and token.  Returns (rotated q, keys, values) like `update`, or
        None - nothing touched - when the op-by-op route has to run (compression on: `update_pre_rope`; no reforging;
        rotary modules that must be called; CPU tensors)."""
        self._shift_latch_check()
        if self.kvcache_compression or not (self.pos_embed_reforge and self.one_call_update) or position_ids is None \
                or not key_states.is_cuda or key_states.shape[0] != 1 \
                or (torch.is_grad_enabled() and query_states.requires_grad):
            return None
        dev = key_states.device
        rot = self._rotary(rotary_emb, dev)
        if rot is None:
            return None
        q, k, v, pos = query_states, key_states, value_states, position_ids
        n, D = k.shape[2], k.shape[3]
        dt = k.dtype
        P = 3 if pos.ndim == 3 else 1
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        if q.ndim != 4 or q.shape[0] != 1 or q.shape[2] != n or q.shape[3] != D or v.shape != k.shape \
                or q.dtype is not dt or v.dtype is not dt or dt not in (torch.float32, torch.bfloat16, torch.float16) \
                or pos.dtype is not torch.int64 or not pos.is_cuda or pos.shape[-1] != n or pos.shape[0] != P \
                or pos.stride(-1) != 1 or (pos.ndim == 3 and pos.shape[1] != 1) or pos.ndim not in (2, 3) \
                or q.get_device() != idx or v.get_device() != idx or pos.get_device() != idx or nv.current_device() != idx:
            return None
        qs, ks, vs = q.stride(), k.stride(), v.stride()
        if qs[3] != 1 or ks[3] != 1 or vs[3] != 1:
            return None
        # M-RoPE ids of a decode step are one row seen three times (`.expand(3, -1, -1)`, qwen2_vl.py:589: stride 0); the
        # kernel then moves t, h and w together, like the reference's in-place shift of the shared storage.  Rows that
        # overlap only partly are not served.
        if P == 3 and 0 < pos.stride(0) < n:
            return None
        st = self._store(layer_idx)
        if st.pending:
            self._flush()
        st = self.reserve(layer_idx, n, k)
        self._pos_reserve(st, P, pos.ndim, n, dev)
        if st.c.pos_len != st.c.length:   # a cache whose earlier rows carry no ids (filled without reforging): not ours
            return None
        io = self._aio
        io.q, io.q_stride_h, io.q_stride_l = q.data_ptr(), qs[1], qs[2]
        io.k, io.k_stride_h, io.k_stride_l = k.data_ptr(), ks[1], ks[2]
        io.v, io.v_stride_h, io.v_stride_l = v.data_ptr(), vs[1], vs[2]
        io.pos, io.pos_stride = pos.data_ptr(), pos.stride(0)
        io.q_rot, io.qr_stride_h, io.qr_stride_l, io.flags = io.q, qs[1], qs[2], 0
        nsec = len(mrope_section) if mrope_section else 0
        sec = (C.c_int * nsec)(*mrope_section) if nsec else None
        rc = nv.lib.rtk_pivotkv_append_rope(st.cref, C.addressof(io), q.shape[1], k.shape[1], n, D, nv.dtype_code(k), P,
                                            rot.inv.data_ptr(), rot.scaling, sec, nsec, nv.round_mode(dt),
                                            int(bool(shift_ids_in_place)), nv.raw_stream(idx))
        if rc == nv.RTK_EUNSUPPORTED:
            return None
        nv.check(rc, "rtk_pivotkv_append_rope")
        self._pos_layers = max(self._pos_layers, layer_idx + 1)
        m = st.c.length
        return q, st._k.narrow(2, 0, m), st._v.narrow(2, 0, m)

    def _bind_rotary(self, b: _Batch, rotary_emb_fn, mrope_section, rot: Optional[_Rotary]):
        """The rotary module / M-RoPE sections the batch's pending units were (and its next units will be) rotated
        with; units of different rotaries never share a flush."""
        same_fn = (b.rotary_emb_fn is rotary_emb_fn and b.rot is rot) or (rot is not None and b.rot is rot)
        if b.pending and (not same_fn or b.mrope_section != mrope_section):
            self._flush()
        b.rotary_emb_fn, b.rot = rotary_emb_fn, rot
        if b.mrope_section != mrope_section or b.c.inv_freq != (rot.inv.data_ptr() if rot is not None else None):
            b.mrope_section = list(mrope_section) if mrope_section is not None else None
            c = b.c
            c.nsec = len(mrope_section) if mrope_section else 0
            if c.nsec > 8:
                raise ValueError("mrope_section has more than 8 entries")
            for i in range(c.nsec):
                c.sections[i] = int(mrope_section[i])
            c.inv_freq = rot.inv.data_ptr() if rot is not None else None
            c.attention_scaling = rot.scaling if rot is not None else 1.0

    def _update_general(
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
        if not self._warned:  # the reference's logger.warning_once (:232): a log line on stderr, once per process
            self._warned = True
            _warn_once("Enable PivotKVCache compression: length after compression %.2f" % (self.compression_ratio))
        cache_kwargs = cache_kwargs if cache_kwargs is not None else {}
        position_ids = cache_kwargs.pop("position_ids", None)
        cache_kwargs.pop("shift_next_position_ids", None)   # (the build's own key: only the one-call route acts on it)
        nv.require_device(key_states, value_states)
        assert key_states.shape[0] == 1, "PivotKVCache supports bsz == 1 only"

        # 1) append: the next layer's hidden states see the uncompressed chunk (reference :238)
        n_new = key_states.shape[2]
        if len(self._layers) > layer_idx and self._layers[layer_idx].pending:
            self._flush()

        if not self.kvcache_compression:  # text prefill / decode (reference :319-321)
            st = self.reserve(layer_idx, n_new, key_states)
            P0 = st.length
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
        if keep_len > q_len:   # compression_ratio > 1: the reference's topk refuses it (:276)
            raise RuntimeError(f"PivotKVCache.update: selected index k out of range (keep {keep_len} of {q_len} tokens)")
        reforge = bool(self.pos_embed_reforge)

        mask = getattr(self, "keypatches_mask_chunk", None)
        if mask is not None:
            nv.require_device(mask)
            if mask.dtype != torch.bool or not mask.is_contiguous():
                mask = mask.to(torch.bool).contiguous()
            assert mask.numel() == L, "keypatches_mask_chunk must have one entry per chunk token"
        Pn = 0
        if position_ids is not None:
            nv.require_device(position_ids)
            Pn = 3 if position_ids.ndim == 3 else 1
        batch = self._get_batch(layer_idx, Hq, Hkv, L, D, keep_len, Pn, key_states.dtype, dev)
        rot = self._rotary(rotary_emb_fn, dev) if reforge else None
        self._bind_rotary(batch, rotary_emb_fn, mrope_section, rot)
        if batch.pending and batch.c.pre_rope:   # units of the prologue route are flushed among themselves
            self._flush()
        batch.c.pre_rope = 0
        slot = self._claim_slot(batch, layer_idx)
        batch.x_like = value_states[:, :, :1]
        st = self.reserve(layer_idx, n_new, key_states)
        P0 = st.length
        cap = st.k.shape[2]
        esz = st.k.element_size()

        a_scale = float(getattr(rotary_emb_fn, "attention_scaling", 1.0)) if reforge else 1.0
        ws_bytes = batch.ws_bytes
        sel_bytes = nv.lib.rtk_pivotkv_select_workspace_bytes(L)
        keep_idx = batch.keep_idx[slot]
        shared = {}   # values handed from one stage to the next

        defer_select = L >= 512   # the chip-wide selection kernels; smaller chunks select inside update

        def ws_pointer(ws):
            if batch.batched_passes:   # the slot's own workspace: q~ must survive until the batched passes of the flush
                return batch.score_ws_base + slot * batch.ws_stride
            wsb = self._buf("score_ws", (ws_bytes + 256,), torch.uint8, dev, ws)
            return (wsb.data_ptr() + 255) & ~255

        def score_stage(ws, stages):
            ws_ptr = ws_pointer(ws)
            score = batch.score[slot]
            k_unrot = batch.k_unrot[slot] if reforge else None
            # the matrix passes of a per-update launch know the chunk's key-patch mask: pass 2 skips the columns the
            # selection overwrites with 1.0 anyway (reference :272-274)
            live = mask if (stages & nv.SCORE_PASSES) and self.skip_masked_columns else None
            kidx = self._buf("key_index", (L + 1,), torch.int32, dev, ws) if live is not None else None
            nv.check(nv.lib.rtk_pivotkv_score_stages_masked(
                nv.ptr(query_states), query_states.stride(1), query_states.stride(2),
                nv.ptr(key_states), key_states.stride(1), key_states.stride(2),
                Hq, Hkv, L, D, batch.score_dt, nv.ptr(shared.get("cos")), nv.ptr(shared.get("sin")), a_scale,
                nv.ptr(score), nv.ptr(k_unrot), C.c_void_p(ws_ptr), ws_bytes, stages, nv.ptr(batch.partials[slot]),
                nv.ptr(live), nv.ptr(kidx), nv.stream()), "rtk_pivotkv_score")
            return score

        def stage_pre(ws, pos_in):
            """RoPE tables of the chunk's ids + un-rotate / pack (reference :248-259), on the CURRENT stream."""
            batch.ensure_scoring()
            if reforge:
                cos_t = self._buf("old_cos", (L, D), torch.float32, dev, ws)
                sin_t = self._buf("old_sin", (L, D), torch.float32, dev, ws)
                self._rope_tables(cos_t, sin_t, rotary_emb_fn, value_states, pos_in, L, position_ids.ndim, mrope_section,
                                  L, D)
                shared["cos"], shared["sin"] = cos_t, sin_t
            score_stage(ws, nv.SCORE_PREPARE)

        def stage_big(ws):
            """the two matrix passes (reference :260-268): deferred to the flush (all layers of the chunk in one
            launch per kernel) whenever the batched form supports the shape"""
            if batch.keep_all:         # nothing to choose: no scores
                batch.scored.add(layer_idx)
                return
            if batch.batched_passes:
                return
            score_stage(ws, nv.SCORE_PASSES)
            batch.scored.add(layer_idx)

        def stage_post(ws, pos_in):
            """small chunks only: column-mass reduction, mask override + top-k + position ids (reference :269-295)
            right away; larger chunks leave this to the batched selection of the flush"""
            if batch.keep_all:         # keep_idx is the identity (set once per batch); ids x 1.0 = the ids (:288-292)
                if pos_in is not None:
                    batch.pos_new[:, slot].copy_(pos_in)
                batch.selected.add(layer_idx)
                return
            if defer_select:
                batch.masks[layer_idx] = mask
                return
            score = score_stage(ws, nv.SCORE_FINALIZE)
            rank = self._buf("rank", (L,), torch.int32, dev, ws)
            pos_out = batch.pos_new[:, slot] if pos_in is not None else None
            sel_ws = self._buf("select_ws", (sel_bytes,), torch.uint8, dev, ws)
            nv.check(nv.lib.rtk_pivotkv_select(nv.ptr(score), nv.ptr(mask), L, keep_len, nv.ptr(pos_in), Pn,
                                               int(reforge), nv.ptr(keep_idx), nv.ptr(rank), nv.ptr(pos_out),
                                               batch.slots * keep_len, nv.ptr(sel_ws), sel_bytes, nv.stream()),
                     "rtk_pivotkv_select")
            batch.selected.add(layer_idx)

        def pos_2d(snapshot: bool):
            if position_ids is None:
                return None
            p2 = position_ids.reshape(Pn, L)
            # worker streams read the ids later than the caller's stream runs on: the attention patch shifts the
            # SAME ids tensor in place for the next layer (qwen2_vl.py:73), so they get a private copy
            return p2.clone() if snapshot else (p2 if p2.is_contiguous() else p2.contiguous())

        k_tail = C.c_void_p(st.k.data_ptr() + P0 * D * esz)
        v_tail = C.c_void_p(st.v.data_ptr() + P0 * D * esz)

        def append_tail():   # reference :238 — the uncompressed view this layer's attention reads
            nv.check(nv.lib.rtk_pivotkv_append(
                nv.ptr(key_states), key_states.stride(1), key_states.stride(2),
                nv.ptr(value_states), value_states.stride(1), value_states.stride(2), Hkv, L, D, dt,
                k_tail, v_tail, cap * D, nv.stream()), "rtk_pivotkv_append")

        def fused_prepare(ws, pos_in) -> bool:
            """append + native RoPE tables + un-rotate in ONE launch (k read once); False if the shape needs the
            separate kernels."""
            if not (reforge and rot is not None and pos_in is not None):
                return False
            inv = rot.inv
            # keep-all chunks are not scored: no q~, no score workspace (any aligned address will do)
            k_only = batch.keep_all and batch.partials is None
            ws_ptr = batch.c.score_ws if k_only else ws_pointer(ws)
            sec = (C.c_int * len(mrope_section))(*mrope_section) if mrope_section else None
            rc = nv.lib.rtk_pivotkv_prepare(
                nv.ptr(query_states), query_states.stride(1), query_states.stride(2),
                nv.ptr(key_states), key_states.stride(1), key_states.stride(2),
                nv.ptr(value_states), value_states.stride(1), value_states.stride(2),
                Hq, Hkv, L, D, batch.prep_dt | (nv.RTK_PREPARE_K_ONLY if k_only else 0), nv.ptr(pos_in), L, Pn,
                nv.ptr(inv), a_scale, sec,
                len(mrope_section) if mrope_section else 0, nv.round_mode(key_states.dtype),
                nv.ptr(batch.k_unrot[slot]), C.c_void_p(ws_ptr), ws_bytes, k_tail, v_tail, cap * D,
                nv.ptr(batch.pos_old[slot]) if defer_select else None, nv.stream())
            if rc == nv.RTK_EUNSUPPORTED:
                return False
            nv.check(rc, "rtk_pivotkv_prepare")
            return True

        with torch.cuda.device(dev):
            side = self._next_side(dev)
            if side is None:
                pos_in = pos_2d(False)
                if not fused_prepare(self._ws, pos_in):
                    append_tail()
                    stage_pre(self._ws, pos_in)
                    if defer_select and pos_in is not None:
                        batch.pos_old[slot].copy_(pos_in)
                stage_big(self._ws)
                stage_post(self._ws, pos_in)
            else:
                append_tail()   # on the caller's stream: all this layer's attention needs
                main = torch.cuda.current_stream()
                pos_in = pos_2d(True)
                ready = torch.cuda.Event()
                ready.record(main)
                for t in (query_states, key_states, value_states, pos_in, mask):
                    if t is not None:
                        t.record_stream(side.stream)
                with torch.cuda.stream(side.stream):
                    side.stream.wait_event(ready)
                    stage_pre(side.ws, pos_in)
                    if defer_select and pos_in is not None:
                        batch.pos_old[slot].copy_(pos_in)
                    stage_big(side.ws)
                    stage_post(side.ws, pos_in)
                    done = torch.cuda.Event()
                    done.record(side.stream)
                st.pending_event = done
        self.update_num_evicted_tokens(k_len - keep_len, layer_idx)  # reference :310
        st.pending = n_new
        st.pending_keep = keep_len
        batch.pending.append(layer_idx)
        self._last_slot = (batch, slot)
        return st.k[:, :, :P0 + n_new], st.v[:, :, :P0 + n_new]


def build_kvcache(config, reserve_tokens: Optional[int] = None):
    """DynamicCache unless longvideo_kwargs enables 'pivotkv' compression (reference :326-334).  reserve_tokens: optional
    capacity hint for the pre-allocated PivotKV cache (see PivotKVCache.__init__)."""
    if getattr(config, "longvideo_kwargs", None) is None or not config.longvideo_kwargs.get("kvcache_compression", False):
        return DynamicCache()
    compression_method = config.longvideo_kwargs["kvcache_compression_kwargs"]["compression_method"]
    if compression_method.lower() == "pivotkv":
        return PivotKVCache(config, reserve_tokens=reserve_tokens)
    raise NotImplementedError
