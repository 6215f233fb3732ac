"""DPSelect on MI355X.  Same entry point as the reference's retake/visual_compression.py.

`memory_bank_compress_keyframe` keeps the reference's signature and return tuple
(visual_compression.py:86, :177); the three stages run as HIP kernels:
    rtk_dpselect_dis     adjacent-frame cosine distance          (:100-106)
    rtk_dpselect_select  peaks + bonus + top-k + sorted indices  (:108-135 / :142-169)
    rtk_gather_frames    frame gather                            (:138 / :173)
`memory_bank_compress_MALLM` / `memory_bank_compress_MALLM_hard` (:5-83) are one merge step each, like the
reference's: rtk_adjacent_cosine -> rtk_mallm_argmax -> rtk_mallm_merge.
"""
from __future__ import annotations

import torch

from . import _native as nv

__all__ = ["memory_bank_compress_keyframe", "memory_bank_compress_MALLM", "memory_bank_compress_MALLM_hard",
           "memory_bank_compress_MALLM_hard_to", "dpselect_stages"]


def dpselect_stages(memory_bank: torch.Tensor, tgt_mem_len: int, window_size: int = 3, sync: bool = True):
    """Runs the three kernels and also returns the intermediates (dis [T,N], idx, keys) for tests/bench."""
    if memory_bank.ndim != 4:
        raise ValueError(f"memory_bank must be [B,T,N,C], got {tuple(memory_bank.shape)}")
    B, T, N, Cc = memory_bank.shape
    tgt_mem_len = int(tgt_mem_len)
    # the reference's failures on degenerate calls, same exception classes, before anything is launched
    if T < 2:
        # no adjacent pair: the distance rows are empty and max_pool1d refuses them (visual_compression.py:100-123)
        raise RuntimeError("memory_bank_compress_keyframe: a single frame (T == 1) has no adjacent-frame distances to pool")
    if not sync and N == 1:
        # `.squeeze()` drops the patch axis (visual_compression.py:153-156)
        raise IndexError("memory_bank_compress_keyframe: sync=False with a single patch position (N == 1)")
    if not 0 <= tgt_mem_len <= T:
        raise RuntimeError(f"memory_bank_compress_keyframe: selected index k out of range (tgt_mem_len {tgt_mem_len}, "
                           f"T {T})")                                      # torch.topk's error (:134 / :167)
    nv.require_device(memory_bank)
    # Only the first batch entry's distances choose the frames (visual_compression.py:101).  sync: every batch entry is
    # gathered at those frames (:138); async: torch.gather with a [1, t, N, C] index returns batch entry 0 only (:173).
    xs = [memory_bank[b] if memory_bank[b].is_contiguous() else memory_bank[b].contiguous()
          for b in range(B if sync else 1)]
    x = xs[0]
    dt = nv.dtype_code(x)
    dev = x.device
    with torch.cuda.device(dev):
        st = nv.stream()
        dis = torch.empty((T, N), dtype=torch.float32, device=dev)
        nv.check(nv.lib.rtk_dpselect_dis(nv.ptr(x), T, N, Cc, dt, nv.ptr(dis), st), "rtk_dpselect_dis")
        idx = torch.empty((tgt_mem_len,) if sync else (tgt_mem_len, N), dtype=torch.int64, device=dev)
        mask = torch.empty((tgt_mem_len, N), dtype=torch.bool, device=dev)
        out = torch.empty((len(xs), tgt_mem_len, N, Cc), dtype=x.dtype, device=dev)
        if tgt_mem_len == 0:  # topk(k=0): empty selections (the callers never ask for it: tgt = max(1, ...))
            return out, mask, idx, dis, None
        keys = torch.empty((2, T) if sync else (N, T), dtype=torch.float32, device=dev)
        nv.check(nv.lib.rtk_dpselect_select(nv.ptr(dis), T, N, tgt_mem_len, int(window_size), int(bool(sync)),
                                            nv.ptr(idx), nv.ptr(mask), nv.ptr(keys), st), "rtk_dpselect_select")
        for b, xb in enumerate(xs):
            nv.check(nv.lib.rtk_gather_frames(nv.ptr(xb), T, N, Cc, dt, nv.ptr(idx), tgt_mem_len, int(bool(sync)),
                                              nv.ptr(out[b]), st), "rtk_gather_frames")
    return out, mask, idx, dis, keys


def memory_bank_compress_keyframe(memory_bank: torch.Tensor, tgt_mem_len: int, window_size: int = 3,
                                  sync: bool = True) -> tuple:
    """DPSelect (reference: visual_compression.py:86-177).

    Args:
        memory_bank: [B, T, N, C] frame embeddings (float32, bfloat16 or float16, on the ROCm device); the callers
            pass B = 1, and like the reference only batch entry 0 chooses the frames
        tgt_mem_len: number of frames to keep
        window_size: argrelmax window (the callers pass 3)
        sync: True = one frame set for all patch positions; False = per-patch frame sets
    Returns:
        compressed_memory_bank [B, t, N, C] (sync) / [1, t, N, C] (async) (fresh tensor),
        keypatches_mask.flatten() [t*N] bool
    """
    out, mask, _, _, _ = dpselect_stages(memory_bank, tgt_mem_len, window_size, sync)
    return out, mask.flatten()


def _mallm_step(memory_bank: torch.Tensor, compression_size, sync: bool, hard: bool):
    if memory_bank.ndim != 4:
        raise ValueError(f"memory_bank must be [B,T,N,C], got {tuple(memory_bank.shape)}")
    B, T, N, Cc = memory_bank.shape
    if T < 2:  # argmax over an empty similarity axis (visual_compression.py:19-23 / :62-66)
        raise IndexError("memory_bank_compress_MALLM: max(): a single frame (T == 1) has no adjacent pair to merge")
    nv.require_device(memory_bank)
    dev, dt = memory_bank.device, nv.dtype_code(memory_bank)
    out = torch.empty((B, T - 1, N, Cc), dtype=memory_bank.dtype, device=dev)
    sizes_out = None if hard else torch.empty((B, T - 1, N), dtype=memory_bank.dtype, device=dev)
    with torch.cuda.device(dev):
        st = nv.stream()
        cosv = torch.empty((T - 1, N), dtype=torch.float32, device=dev)
        idx = torch.empty((N,), dtype=torch.int64, device=dev)
        for b in range(B):  # the reference's ops are batched over B; the callers pass B = 1
            x = memory_bank[b]
            if not x.is_contiguous():
                x = x.contiguous()
            sz = None
            if not hard:
                sz = compression_size[b].to(memory_bank.dtype)
                if not sz.is_contiguous():
                    sz = sz.contiguous()
            nv.check(nv.lib.rtk_adjacent_cosine(nv.ptr(x), T, N, Cc, dt, nv.ptr(cosv), st), "rtk_adjacent_cosine")
            nv.check(nv.lib.rtk_mallm_argmax(nv.ptr(cosv), T - 1, N, int(bool(sync)), nv.round_mode(memory_bank.dtype),
                                             nv.ptr(idx), st), "rtk_mallm_argmax")
            nv.check(nv.lib.rtk_mallm_merge(nv.ptr(x), nv.ptr(sz), nv.ptr(idx), T, N, Cc, dt, int(hard), nv.ptr(out[b]),
                                            None if hard else nv.ptr(sizes_out[b]), st), "rtk_mallm_merge")
    return out, sizes_out


def memory_bank_compress_MALLM(memory_bank: torch.Tensor, compression_size: torch.Tensor, sync: bool = False) -> tuple:
    """MA-LLM merge step (reference: visual_compression.py:5-47): the most similar adjacent frame pair of every
    patch position (of the patch mean when `sync`) is merged into its first frame, weighted by how many frames
    each already holds.  Returns (compressed_memory_bank [B,T-1,N,C], compressed_size [B,T-1,N])."""
    return _mallm_step(memory_bank, compression_size, sync, hard=False)


def memory_bank_compress_MALLM_hard(memory_bank: torch.Tensor, sync: bool = False) -> torch.Tensor:
    """MA-LLM-hard merge step (reference: visual_compression.py:50-83): the first frame of the most similar
    adjacent pair is replaced by the second.  Returns compressed_memory_bank [B,T-1,N,C]."""
    return _mallm_step(memory_bank, None, sync, hard=True)[0]


def memory_bank_compress_MALLM_hard_to(memory_bank: torch.Tensor, tgt_mem_len: int, sync: bool = False) -> torch.Tensor:
    """`while bank.shape[1] > tgt: bank = memory_bank_compress_MALLM_hard(bank, sync)` (the reference's loop,
    qwen2_vl.py:406-408) in one pass over the bank: rtk_mallm_hard_chain finds the surviving frames - per step the only
    new adjacent cosine is the pair that closes over the dropped frame - and rtk_gather_frames copies them.  Same
    values as the loop, bit for bit; shapes the chain does not serve (odd row sizes, T beyond ~12 000) take the loop."""
    if memory_bank.ndim != 4:
        raise ValueError(f"memory_bank must be [B,T,N,C], got {tuple(memory_bank.shape)}")
    B, T, N, Cc = memory_bank.shape
    tgt = int(tgt_mem_len)
    if T <= tgt:
        return memory_bank
    nv.require_device(memory_bank)
    dev, dt = memory_bank.device, nv.dtype_code(memory_bank)
    if tgt >= 1:
        out = torch.empty((B, tgt, N, Cc), dtype=memory_bank.dtype, device=dev)
        with torch.cuda.device(dev):
            st = nv.stream()
            cosv = torch.empty((T - 1, N), dtype=torch.float32, device=dev)
            idx = torch.empty((tgt,) if sync else (tgt, N), dtype=torch.int64, device=dev)
            ok = True
            for b in range(B):  # batched in the reference; the callers pass B = 1
                x = memory_bank[b] if memory_bank[b].is_contiguous() else memory_bank[b].contiguous()
                rc = nv.lib.rtk_mallm_hard_chain(nv.ptr(x), T, N, Cc, dt, tgt, int(bool(sync)), nv.ptr(cosv), nv.ptr(idx), st)
                if rc == nv.RTK_EUNSUPPORTED:
                    ok = False
                    break
                nv.check(rc, "rtk_mallm_hard_chain")
                nv.check(nv.lib.rtk_gather_frames(nv.ptr(x), T, N, Cc, dt, nv.ptr(idx), tgt, int(bool(sync)), nv.ptr(out[b]),
                                                  st), "rtk_gather_frames")
            if ok:
                return out
    bank = memory_bank
    while bank.shape[1] > tgt:
        bank = memory_bank_compress_MALLM_hard(bank, sync=sync)
    return bank
