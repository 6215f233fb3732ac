"""ctypes binding of libretake_hip.so (C ABI in include/retake_hip.h).

The HIP library is the product path: there is NO CPU or eager fallback.  Importing this module
without the built library, or calling into it with non-ROCm tensors, raises immediately.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# RETAKE_HIP_LIB lets kernel developers A/B an alternative build of the same ABI (tools/variants.sh)
LIB_PATH = os.environ.get("RETAKE_HIP_LIB") or os.path.join(_HERE, "_lib", "libretake_hip.so")
ABI_VERSION = 17

RTK_F32, RTK_BF16, RTK_BF16_REFROUND, RTK_BF16_FAST, RTK_F16, RTK_F16_REFROUND = 0, 1, 2, 3, 4, 5
RTK_SCORE_MANY_UNITS = 0x100   # flag for the dtype argument of the scoring entry points (split policy of batched launches)
RTK_PREPARE_K_ONLY = 0x200     # flag for the dtype argument of rtk_pivotkv_prepare: keep-all chunk, no q~
RTK_UPDATE_PRE_ROPE = 1        # rtk_update_io.flags: q / k are the pre-RoPE projections (attention prologue)
RTK_UPDATE_Q_IN_PLACE = 2      # ... and the score passes read q where it is (no packed copy)
RTK_UPDATE_ROUNDTRIP = 4       # ... or: q~ / k~ = the reference's un-rotation of the rotated rows (its bf16 round trip)
RTK_UPDATE_SHIFT_NEXT = 8      # rotated q / k: the launch also applies the NEXT layer's continuity shift to the caller's ids
SCORE_PREPARE, SCORE_PASSES, SCORE_FINALIZE = 1, 2, 4
RTK_EINVAL, RTK_EUNSUPPORTED, RTK_EWORKSPACE, RTK_EHIP, RTK_EREFCRASH = -1, -2, -3, -4, -5

_vp, _i, _i64, _f, _sz = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t


class EvictUnit(C.Structure):
    """rtk_evict_unit (include/retake_hip.h)."""
    _fields_ = [("k_src", _vp), ("k_src_stride_h", _i64), ("v_src", _vp), ("v_src_stride_h", _i64),
                ("keep_idx", _vp), ("cos_new", _vp), ("sin_new", _vp),
                ("k_dst", _vp), ("k_dst_stride_h", _i64), ("v_dst", _vp), ("v_dst_stride_h", _i64),
                ("pos_src", _vp), ("pos_src_stride", _i64), ("pos_dst", _vp), ("pos_dst_stride", _i64)]


class SelectUnit(C.Structure):
    """rtk_select_unit (include/retake_hip.h)."""
    _fields_ = [("partial", _vp), ("score", _vp), ("mask", _vp), ("pos", _vp), ("keep_idx", _vp), ("rank", _vp),
                ("pos_out", _vp), ("workspace", _vp)]


class PlaceUnit(C.Structure):
    """rtk_place_unit (include/retake_hip.h)."""
    _fields_ = [("stage", _vp), ("stage_stride_h_bytes", _i64), ("tail", _vp), ("tail_stride_h_bytes", _i64),
                ("keep_idx", _vp)]


class CompactUnit(C.Structure):
    """rtk_compact_unit (include/retake_hip.h)."""
    _fields_ = [("k_src", _vp), ("k_src_stride_h", _i64), ("k_tail", _vp), ("k_tail_stride_h", _i64), ("v_tail", _vp),
                ("v_tail_stride_h", _i64), ("keep_idx", _vp), ("pos_src", _vp), ("pos_src_stride", _i64), ("pos_dst", _vp),
                ("pos_dst_stride", _i64)]


COMPACT_K_ROTATE, COMPACT_K_COPY, COMPACT_K_INPLACE = 0, 1, 2
P2P_MAX_RANKS, IPC_HANDLE_BYTES = 16, 64


class P2PPeers(C.Structure):
    """rtk_p2p_peers (include/retake_hip.h)."""
    _fields_ = [("buf", _vp * P2P_MAX_RANKS), ("flag", _vp * P2P_MAX_RANKS)]


class CopyUnit(C.Structure):
    """rtk_copy_unit (include/retake_hip.h)."""
    _fields_ = [("src", _vp), ("src_stride_h_bytes", _i64), ("dst", _vp), ("dst_stride_h_bytes", _i64)]


_u64, _i32 = C.c_uint64, C.c_int32


class LayerState(C.Structure):
    """rtk_layer_state (include/retake_hip.h): one layer's pre-allocated cache, shared with the library."""
    _fields_ = [("k", _vp), ("v", _vp), ("cap", _i64), ("length", _i64), ("pending", _i64), ("pending_keep", _i64),
                ("pos", _vp), ("pos_cap", _i64), ("pos_len", _i64), ("mask", _vp)]


class PivotKVBatch(C.Structure):
    """rtk_pivotkv_batch (include/retake_hip.h)."""
    _fields_ = [("Hq", _i32), ("Hkv", _i32), ("L", _i32), ("D", _i32), ("keep", _i32), ("P", _i32), ("slots", _i32),
                ("dtype", _i32), ("score_dtype", _i32), ("prep_dtype", _i32), ("reforge", _i32), ("keep_all", _i32),
                ("round_mode", _i32), ("nsec", _i32), ("sections", _i32 * 8), ("rs_n", _i32), ("skip_masked", _i32),
                ("attention_scaling", _f), ("defer_rot", _i32), ("inv_freq", _vp),
                ("score_ws", _vp), ("score_ws_stride", _u64), ("score_ws_bytes", _u64),
                ("k_unrot", _vp), ("partials", _vp), ("partial_floats", _u64), ("score", _vp), ("pos_old", _vp),
                ("keep_idx", _vp), ("pos_new", _vp), ("sel_ws", _vp), ("sel_ws_stride", _u64), ("key_index", _vp),
                ("v_stage", _vp), ("k_stage", _vp), ("shift_row", _vp), ("q_units", _vp), ("q_stride_h", _i64),
                ("q_stride_l", _i64), ("pre_rope", _i32), ("batched_passes", _i32), ("compact_sync", _vp),
                ("compact_sync_ints", _u64)]


class UpdateIO(C.Structure):
    """rtk_update_io (include/retake_hip.h)."""
    _fields_ = [("q", _vp), ("q_stride_h", _i64), ("q_stride_l", _i64), ("k", _vp), ("k_stride_h", _i64),
                ("k_stride_l", _i64), ("v", _vp), ("v_stride_h", _i64), ("v_stride_l", _i64), ("pos", _vp),
                ("pos_stride", _i64), ("q_rot", _vp), ("qr_stride_h", _i64), ("qr_stride_l", _i64), ("flags", _i32),
                ("pad0", _i32), ("next_prev", _vp), ("ticket", _vp), ("ticket_ints", _i64), ("status", _vp)]


_SIGNATURES = {
    "rtk_version": (C.c_int, []),
    "rtk_last_error": (C.c_char_p, []),
    "rtk_arch": (C.c_char_p, []),
    "rtk_dpselect_dis": (C.c_int, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "rtk_dpselect_select": (C.c_int, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "rtk_gather_frames": (C.c_int, [_vp, _i, _i, _i, _i, _vp, _i, _i, _vp, _vp]),
    "rtk_adjacent_cosine": (C.c_int, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "rtk_mallm_argmax": (C.c_int, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "rtk_mallm_merge": (C.c_int, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "rtk_mallm_hard_chain": (C.c_int, [_vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "rtk_rope_merge": (C.c_int, [_vp, _vp, _i, _i, _i, _i, _vp, _i, _vp, _vp, _vp]),
    "rtk_rope_table": (C.c_int, [_vp, _i64, _i, _i, _vp, _i, _f, _vp, _i, _i, _vp, _vp, _vp]),
    "rtk_rope_rotate_rows": (C.c_int, [_vp, _i64, _i64, _i, _i, _i, _i, _i, _vp, _i64, _i64, _i, _vp, _f, _vp, _i, _i, _vp]),
    "rtk_rope_shift": (C.c_int, [_vp, _i64, _i, _i, _i, _i, _vp, _vp, _i, _vp, _i, _vp]),
    "rtk_rope_shift_segments": (C.c_int, [_vp, _i64, _i64, _i, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp, _i, _vp]),
    "rtk_pivotkv_score_workspace_bytes": (C.c_size_t, [_i, _i, _i, _i, _i]),
    "rtk_pivotkv_score": (C.c_int, [_vp, _i64, _i64, _vp, _i64, _i64, _i, _i, _i, _i, _i, _vp, _vp, _f, _vp, _vp,
                                    _vp, _sz, _vp]),
    "rtk_pivotkv_score_stages": (C.c_int, [_vp, _i64, _i64, _vp, _i64, _i64, _i, _i, _i, _i, _i, _vp, _vp, _f, _vp, _vp,
                                           _vp, _sz, _i, _vp, _vp]),
    "rtk_pivotkv_score_stages_masked": (C.c_int, [_vp, _i64, _i64, _vp, _i64, _i64, _i, _i, _i, _i, _i, _vp, _vp, _f, _vp,
                                                  _vp, _vp, _sz, _i, _vp, _vp, _vp, _vp]),
    "rtk_pivotkv_score_passes_batched": (C.c_int, [_vp, _sz, _vp, _sz, _vp, _sz, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "rtk_pivotkv_score_passes_batched_q": (C.c_int, [_vp, _sz, _vp, _sz, _vp, _sz, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i64,
                                                     _i64, _vp]),
    "rtk_pivotkv_score_partials": (C.c_size_t, [_i, _i, _i, _i, _i, C.POINTER(C.c_int)]),
    "rtk_pivotkv_select_batched": (C.c_int, [_vp, _i, _i, _i, _i, _i, _i, _i, _i, _i64, _i, _vp]),
    "rtk_pivotkv_prepare": (C.c_int, [_vp, _i64, _i64, _vp, _i64, _i64, _vp, _i64, _i64, _i, _i, _i, _i, _i, _vp, _i64, _i,
                                      _vp, _f, _vp, _i, _i, _vp, _vp, _sz, _vp, _vp, _i64, _vp, _vp]),
    "rtk_pivotkv_select_workspace_bytes": (C.c_size_t, [_i]),
    "rtk_pivotkv_select": (C.c_int, [_vp, _vp, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _i64, _vp, _sz, _vp]),
    "rtk_pivotkv_evict": (C.c_int, [_vp, _i64, _i64, _vp, _i64, _i64, _vp, _i, _i, _i, _i, _vp, _i, _vp, _vp, _vp,
                                    _vp, _i64, _vp, _vp, _i64, _vp]),
    "rtk_copy_rows": (C.c_int, [_vp, _i64, _vp, _i64, _i, _i, _i, _i, _vp]),
    "rtk_pivotkv_commit": (C.c_int, [_vp, _vp, _i64, _vp, _vp, _i64, _i, _i, _i, _i, _vp]),
    "rtk_pivotkv_append": (C.c_int, [_vp, _i64, _i64, _vp, _i64, _i64, _i, _i, _i, _i, _vp, _vp, _i64, _vp]),
    "rtk_pivotkv_evict_batched": (C.c_int, [_vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "rtk_pivotkv_evict_batched_rope": (C.c_int, [_vp, _i, _i, _i, _i, _i, _i, _vp, C.c_float, _vp, _i, _i, _i, _vp]),
    "rtk_pivotkv_place_batched": (C.c_int, [_vp, _i, _i, _i, _i, _i, _vp]),
    "rtk_pivotkv_commit_batched": (C.c_int, [_vp, _i, _i, _i, _i, _i, _vp]),
    "rtk_pivotkv_compact_sync_ints": (C.c_size_t, [_i, _i, _i, _i, _i]),
    "rtk_pivotkv_compact_batched": (C.c_int, [_vp, _i, _i, _i, _i, _i, _i, _i, _vp, _f, _vp, _i, _i, _vp, _sz, _vp]),
    "rtk_position_shift": (C.c_int, [_vp, _i, _vp, _vp]),
    "rtk_pivotkv_shift_ticket_ints": (_sz, [_i, _i]),
    "rtk_pivotkv_update": (C.c_int, [_vp, _vp, _i, _vp, _vp]),
    "rtk_pivotkv_flush": (C.c_int, [_vp, _vp, _vp, _i, _vp]),
    "rtk_pivotkv_append_rope": (C.c_int, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _f, _vp, _i, _i, _i, _vp]),
    "rtk_p2p_alloc": (C.c_int, [_sz, _i, C.POINTER(_vp)]),
    "rtk_p2p_free": (C.c_int, [_vp]),
    "rtk_p2p_export": (C.c_int, [_vp, _vp, C.POINTER(_sz)]),
    "rtk_p2p_open": (C.c_int, [_vp, C.POINTER(_vp)]),
    "rtk_p2p_close": (C.c_int, [_vp]),
    "rtk_p2p_push": (C.c_int, [_vp, _sz, _i, _sz, C.POINTER(P2PPeers), _i, _i, _sz, _sz, C.c_uint32, _vp]),
    "rtk_p2p_wait": (C.c_int, [_vp, _i, C.c_uint32, _i, _vp, _vp]),
    "rtk_profile_enable": (C.c_int, [_i]),
    "rtk_profile_enable_mask": (C.c_int, [C.c_uint]),
    "rtk_profile_collect": (C.c_int, []),
    "rtk_profile_reset": (C.c_int, []),
    "rtk_profile_num_kernels": (C.c_int, []),
    "rtk_profile_kernel_name": (C.c_char_p, [_i]),
    "rtk_profile_read": (C.c_int, [_i, C.POINTER(C.c_longlong), C.POINTER(C.c_double)]),
    "rtk_profile_copy": (C.c_int, [_vp, _vp, _sz, _vp]),
}

EXPORTS = tuple(_SIGNATURES)


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python __graft_entry__.py build` "
            "(or `make -C video-retake_amd/csrc`). The retake package has no CPU/eager fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = stale library
        fn.restype, fn.argtypes = res, args
    if lib.rtk_version() != ABI_VERSION:
        raise ImportError(f"libretake_hip.so ABI {lib.rtk_version()} != expected {ABI_VERSION}: rebuild")
    return lib


lib = _load()


class RetakeHipError(RuntimeError):
    pass


def check(rc: int, what: str):
    """Map an rtk_status to the exception class the reference raises for the same condition."""
    if rc == 0:
        return
    msg = f"{what}: {lib.rtk_last_error().decode(errors='replace')}"
    if rc == RTK_EREFCRASH:
        raise IndexError(msg)
    if rc == RTK_EUNSUPPORTED:
        raise NotImplementedError(msg)
    if rc in (RTK_EINVAL, RTK_EWORKSPACE):
        raise ValueError(msg)
    raise RetakeHipError(msg)


def dtype_code(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return RTK_F32
    if t.dtype == torch.bfloat16:
        return RTK_BF16
    if t.dtype == torch.float16:
        return RTK_F16
    raise NotImplementedError(f"retake HIP kernels support float32, bfloat16 and float16, got {t.dtype}")


def round_mode(dtype: torch.dtype) -> int:
    """The `round_bf16` argument of the C ABI: which 16-bit format a rotary module's cos / sin tables (and the
    intermediate results of the model dtype) are rounded to - 0 = none (fp32), 1 = bf16, 2 = fp16."""
    return 1 if dtype == torch.bfloat16 else (2 if dtype == torch.float16 else 0)


def require_device(*tensors: torch.Tensor):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("retake HIP path needs ROCm device tensors (no CPU fallback); got a "
                               f"{t.device} tensor")


def ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


try:  # the raw handle of the current stream without building a torch.cuda.Stream object (hot path of update)
    _raw_stream = torch._C._cuda_getCurrentRawStream
    _cur_device = torch._C._cuda_getDevice
except AttributeError:  # a torch without these private helpers: the public route
    def _raw_stream(index):
        return torch.cuda.current_stream(index).cuda_stream

    def _cur_device():
        return torch.cuda.current_device()


def raw_stream(index: int) -> int:
    """hipStream_t of torch's current stream on device `index`, as an int."""
    return _raw_stream(index)


def current_device() -> int:
    return _cur_device()


def profile_kernel_ids() -> dict:
    return {lib.rtk_profile_kernel_name(k).decode(): k for k in range(lib.rtk_profile_num_kernels())}


def profile_read() -> dict:
    """{kernel name: (launches, total_ms)} accumulated since the last rtk_profile_reset()."""
    check(lib.rtk_profile_collect(), "rtk_profile_collect")
    out = {}
    for kid in range(lib.rtk_profile_num_kernels()):
        n, ms = C.c_longlong(0), C.c_double(0.0)
        check(lib.rtk_profile_read(kid, C.byref(n), C.byref(ms)), "rtk_profile_read")
        if n.value:
            out[lib.rtk_profile_kernel_name(kid).decode()] = (n.value, ms.value)
    return out
