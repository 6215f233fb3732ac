"""One long video sharded over the GPUs of a node (BASELINE.json configs[3]).

One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI; "gloo" in the CPU tests).
The video's frame chunks are split into contiguous blocks, one per rank:

  DPSelect   every rank computes the distance rows of its frames (one halo frame on the left), the
             [T, N] distance matrix is all-gathered (T*N*4 bytes, 1.6 MB at 2048x196), and the cheap
             stencil + top-k runs redundantly on every rank — deterministic, so all ranks hold the
             same indices and key-patch mask.  Frame gather: at ratio 1.0 (the shipped default) every
             frame stays in place; at ratio < 1 a rank gathers the kept frames it owns into a padded
             block, the blocks are all-gathered, and one more gather puts them in output order
             (`plan_frame_exchange`: kept indices ascend, so a rank's frames are one run per patch).
  PivotKV    (chunk, layer) scoring only reads the chunk's own q/k (longvideo_cache.py:264), so a rank
             runs `PivotKVCache.update` on its chunks with PROVISIONAL temporal ids starting at 0.
             Selection and the id rescale are translation invariant; the only coupling between blocks
             is the temporal offset of a block = last compressed temporal id of the previous block + 1
             (qwen2_vl.py:68-73).  `finalize` all-gathers one int64 per (rank, layer), takes the
             exclusive prefix, rotates each rank's kept keys by R(delta) (RoPE composes:
             R(p + delta) = R(delta) R(p)), shifts the stored ids, and (optionally) all-gathers the
             kept K / V / ids so every rank holds the full compressed cache.

No collective sits inside the scoring path; the exchanges are one small all-gather per video
(distances), one tiny all-gather per video (offsets) and, for the cache assembly, either two all-gathers per
video (K and V of every layer in one, the ids in the other) or - `gather_chunk` - one asynchronous all-gather per
chunk that overlaps the next chunk's scoring, plus the ids at the end.

Transport: `torch.distributed` collectives (RCCL) by default.  `enable_p2p(group)` switches every device-side exchange
of that group to the direct peer-to-peer pushes of retake/p2p.py (`rtk_p2p_*`: mapped peer buffers, one hop on every
xGMI link at once, no ring); with it `gather_chunk` pushes a chunk's kept rows straight into their final position of
every rank's assembled cache, so `finalize` has nothing left to copy.
"""
from __future__ import annotations

import ctypes as C
import json
import math
import os
import sys
import time
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


# ---------------------------------------------------------------------------------------------------
# transport: torch.distributed collectives, or the direct p2p pushes of retake/p2p.py for groups that enabled them
# ---------------------------------------------------------------------------------------------------
_P2P = {}   # process group (None = the default group) -> retake.p2p.P2PGroup


def enable_p2p(group=None, device=None):
    """Route this group's device-side exchanges through mapped peer buffers (collective: every rank must call it)."""
    from .p2p import P2PGroup

    if group not in _P2P:
        _P2P[group] = P2PGroup(group, device)
    return _P2P[group]


def disable_p2p(group=None):
    g = _P2P.pop(group, None)
    if g is not None:
        g.close()


def _p2p_for(t: torch.Tensor, group):
    g = _P2P.get(group)
    return g if g is not None and t.is_cuda else None


def _gather_stack(t: torch.Tensor, group=None) -> torch.Tensor:
    """t (same shape on every rank) -> [world, *t.shape] in rank order: THE collective of this module."""
    t = t.contiguous()
    g = _p2p_for(t, group)
    if g is not None:
        return g.all_gather(t).clone()   # the landing buffer is only valid until this rank's next exchange
    world = dist.get_world_size(group)
    if t.is_cuda and dist.get_backend(group) == "gloo":
        # device tensors over a gloo group (tests/mp_sharded_gpu.py with RETAKE_TEST_TRANSPORT=host, a debugging aid: the
        # sharded path with NO device-side transport at all - no RCCL, no peer mapping): staged through the host
        h = t.cpu()
        parts = [torch.empty_like(h) for _ in range(world)]
        dist.all_gather(parts, h, group=group)
        return torch.stack(parts).to(t.device)
    recv = torch.empty((world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
    if t.is_cuda:
        dist.all_gather_into_tensor(recv, t, group=group)
    else:  # gloo has no all_gather_into_tensor on every build
        dist.all_gather(list(recv.unbind(0)), t, group=group)
    return recv


# ---------------------------------------------------------------------------------------------------
# host logic (device independent; exercised by the gloo tests)
# ---------------------------------------------------------------------------------------------------
def shard_chunks(n_chunks: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous, balanced chunk blocks [c0, c1) per rank (first `n_chunks % world` ranks get one more)."""
    base, extra = divmod(n_chunks, world)
    out, c = [], 0
    for r in range(world):
        n = base + (1 if r < extra else 0)
        out.append((c, c + n))
        c += n
    return out


def exchange_temporal_offsets(local_last: torch.Tensor, first_start: int = 0, group=None,
                              all_ranks: bool = False) -> torch.Tensor:
    """local_last [layers] int64 = last PROVISIONAL temporal id of this rank's block per layer (block ids
    start at 0; -1 if the block kept nothing).  Returns delta [layers] int64 for this rank: the true start
    of the block, i.e. first_start + sum over previous ranks of (last + 1); with all_ranks the whole
    [world, layers] table (every rank computes the same one)."""
    rank = dist.get_rank(group)
    spans = _gather_stack(local_last, group) + 1           # [world, layers]
    prefix = torch.cumsum(spans, dim=0) - spans            # exclusive
    table = (prefix + first_start).contiguous()
    return table if all_ranks else table[rank]


def all_gather_cat(t: torch.Tensor, dim: int, group=None) -> torch.Tensor:
    """All-gather equally shaped tensors and concatenate along `dim` in rank order."""
    return torch.cat(list(_gather_stack(t, group).unbind(0)), dim=dim)


def all_gather_rows(local: torch.Tensor, group=None) -> torch.Tensor:
    """[rows_local, ...] -> [world*rows_local, ...] in rank order (distance rows)."""
    full = _gather_stack(local, group)
    return full.reshape((full.shape[0] * local.shape[0],) + tuple(local.shape[1:]))


def gather_counts(n: int, device, group=None) -> List[int]:
    """How many rows every rank is about to contribute (one tiny all-gather).  Equal-size collectives
    (`all_gather_into_tensor`) hang or corrupt memory when the ranks disagree, so every assembly asks first."""
    mine = torch.tensor([int(n)], dtype=torch.int64, device=device)
    counts = [int(c) for c in _gather_stack(mine, group).reshape(-1).tolist()]
    g = _P2P.get(group)
    if g is not None and mine.is_cuda:
        g.check()   # the .tolist() above synchronised: a bounded p2p wait that gave up (lost / late peer) raises HERE
    return counts


def all_gather_rows_ragged(local: torch.Tensor, group=None, counts: Optional[List[int]] = None) -> torch.Tensor:
    """[rows_r, ...] with a different rows_r per rank -> [sum rows, ...] in rank order: the blocks are padded to the
    longest one, exchanged with ONE equal-size all-gather and trimmed (a video whose chunk count is not a multiple of
    the world size gives the last ranks one chunk less)."""
    counts = counts if counts is not None else gather_counts(local.shape[0], local.device, group)
    if len(set(counts)) == 1:
        return all_gather_rows(local, group)
    nmax = max(counts)
    padded = torch.zeros((nmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    padded[:local.shape[0]] = local
    full = all_gather_rows(padded, group).reshape((len(counts), nmax) + tuple(local.shape[1:]))
    return torch.cat([full[r, :n] for r, n in enumerate(counts)], dim=0)


def _all_gather_flat(send: torch.Tensor, group=None) -> torch.Tensor:
    """send [...] -> [world, ...] in rank order, one collective.  With the p2p transport the result is the landing
    buffer itself (no copy): it is only valid until this rank's NEXT exchange on the group, so the callers below copy
    out of it (permute + reshape / cat) before they return - except at world size 1, where those are views: cloned."""
    g = _p2p_for(send, group)
    if g is not None:
        out = g.all_gather(send.contiguous())
        return out.clone() if g.world == 1 else out
    return _gather_stack(send, group)


def all_gather_caches(keys: List[torch.Tensor], values: List[torch.Tensor], pos: List[torch.Tensor], group=None):
    """Cache assembly for ALL layers with two collectives instead of three per layer (xGMI is point to point: few
    large all-gathers use the links far better than 84 small ones, and each result needs one permute copy instead
    of a per-tensor scatter + cat).

    keys / values: per layer [1, Hkv, n, D]; pos: per layer [3, 1, n] or [1, n] int64.  The ranks first exchange their
    row counts: equal counts (every chunk keeps exactly `keep` tokens and the ranks hold equally many chunks) move
    exactly the rows, unequal ones are padded to the longest block and trimmed after the gather.  Returns the
    per-layer lists [1, Hkv, sum n, D] / [..., sum n] in rank order."""
    world = dist.get_world_size(group)
    n_layers = len(keys)
    _, Hkv, n, D = keys[0].shape
    counts = gather_counts(n, keys[0].device, group)
    nmax = max(counts)
    alloc = torch.empty if n == nmax and len(set(counts)) == 1 else torch.zeros
    send = alloc((2, n_layers, Hkv, nmax, D), dtype=keys[0].dtype, device=keys[0].device)
    if n == nmax:   # stack straight into the send buffer
        torch.stack([k[0] for k in keys], out=send[0])
        torch.stack([v[0] for v in values], out=send[1])
    else:
        send[0, :, :, :n].copy_(torch.stack([k[0] for k in keys]))
        send[1, :, :, :n].copy_(torch.stack([v[0] for v in values]))
    recv = _all_gather_flat(send, group)                                        # [W, 2, layers, Hkv, nmax, D]
    if len(set(counts)) == 1:
        kv = recv.permute(1, 2, 3, 0, 4, 5).reshape(2, n_layers, Hkv, world * n, D)   # one copy: rank-major inside a head
    else:   # ragged blocks (chunk count not a multiple of the world size): drop every rank's padding
        kv = torch.cat([recv[r, :, :, :, :c] for r, c in enumerate(counts)], dim=3)
    return ([kv[0, l][None] for l in range(n_layers)], [kv[1, l][None] for l in range(n_layers)],
            all_gather_ids(pos, group, counts))


def all_gather_ids(pos: List[torch.Tensor], group=None, counts: Optional[List[int]] = None) -> List[torch.Tensor]:
    """Per-layer position ids [..., n_r] of every rank -> [..., sum n_r] in rank order, one collective (blocks padded
    to the longest one when the ranks hold different numbers of rows)."""
    world = dist.get_world_size(group)
    n_layers, n = len(pos), pos[0].shape[-1]
    counts = counts if counts is not None else gather_counts(n, pos[0].device, group)
    nmax, total = max(counts), sum(counts)
    pshape = tuple(pos[0].shape[:-1])
    rows = 1
    for d in pshape:
        rows *= d
    psend = torch.zeros((n_layers, rows, nmax), dtype=pos[0].dtype, device=pos[0].device)
    psend[:, :, :n] = torch.stack([p.reshape(rows, n) for p in pos])            # [layers, rows, nmax]
    precv = _all_gather_flat(psend, group)                                      # [W, layers, rows, nmax]
    if len(set(counts)) == 1:
        pall = precv.permute(1, 2, 0, 3).reshape(n_layers, rows, world * n)
    else:
        pall = torch.cat([precv[r, :, :, :c] for r, c in enumerate(counts)], dim=2)
    return [pall[l].reshape(pshape + (total,)) for l in range(n_layers)]


class _Done:
    def wait(self):
        return True


class ChunkGather:
    """Cache assembly overlapped with the compression of the following chunks: after each chunk's flush the rows it
    kept in every layer go out in ONE asynchronous all-gather (RCCL runs it on its own stream beside the score
    kernels of the next chunk); `finish` waits for the lot and lays the blocks out rank-major.  The rows travel at
    their PROVISIONAL temporal position - the caller applies R(delta_r) to rank r's segment afterwards."""

    def __init__(self, group=None):
        self.group = group
        self.items = []   # (work, recv [W, 2, layers, Hkv, n, D], send)

    def start(self, k_new: List[torch.Tensor], v_new: List[torch.Tensor]):
        """k_new / v_new: per layer [Hkv, n, D] (any strides), the rows one chunk added.  Every rank must call this
        equally often with equally many rows (callers whose ranks hold different chunk counts assemble at the end
        instead: `finalize` checks the totals before it trusts the overlapped gathers)."""
        n_layers = len(k_new)
        Hkv, n, D = k_new[0].shape
        send = torch.empty((2, n_layers, Hkv, n, D), dtype=k_new[0].dtype, device=k_new[0].device)
        torch.stack(list(k_new), out=send[0])
        torch.stack(list(v_new), out=send[1])
        world = dist.get_world_size(self.group)
        if send.is_cuda and dist.get_backend(self.group) == "gloo":   # host-staged (see _gather_stack): done when it returns
            self.items.append((_Done(), _gather_stack(send, self.group), send))
            return
        recv = torch.empty((world,) + tuple(send.shape), dtype=send.dtype, device=send.device)
        if send.is_cuda:
            work = dist.all_gather_into_tensor(recv, send, group=self.group, async_op=True)
        else:
            work = dist.all_gather(list(recv.unbind(0)), send, group=self.group, async_op=True)
        self.items.append((work, recv, send))

    def rows(self) -> int:
        return sum(it[2].shape[3] for it in self.items)

    def pushes(self) -> int:
        return len(self.items)

    def drop(self, lo: int = 0, hi: int = 0):
        """Abandon the started gathers.  They are collectives: every rank must have started equally many (`lo == hi`),
        which `sharded_video_step` guarantees by only overlapping even chunk splits - a mismatch cannot be repaired
        after the fact (the extra all-gather of one rank has no partner), so it raises instead of hanging in `wait`."""
        if lo != hi:
            raise RuntimeError(f"ranks started between {lo} and {hi} per-chunk all-gathers: unmatched collectives")
        for work, _, _ in self.items:
            work.wait()
        self.items = []

    def finish(self) -> torch.Tensor:
        """-> [2, layers, Hkv, world * rows, D]: rank-major, then chunk order, inside every head."""
        for work, _, _ in self.items:
            work.wait()
        world, _, n_layers, Hkv, _, D = self.items[0][1].shape
        total = self.rows()
        out = torch.empty((2, n_layers, Hkv, world, total, D), dtype=self.items[0][1].dtype,
                          device=self.items[0][1].device)
        at = 0
        for _, recv, _ in self.items:
            n = recv.shape[4]
            out[:, :, :, :, at:at + n] = recv.permute(1, 2, 3, 0, 4, 5)
            at += n
        self.items = []
        return out.view(2, n_layers, Hkv, world * total, D)


class ChunkGatherP2P:
    """`ChunkGather` over mapped peer buffers: the rows a chunk kept are pushed (side stream, beside the next chunk's
    scoring) straight to their FINAL position in every rank's assembled cache `[2, layers, Hkv, world * total, D]` -
    rank-major, then chunk order, inside every head - so `finish` only waits for the arrival flags; there is no receive
    buffer to permute.  Needs the rows per rank and video (`total`) up front.

    Lifetime: two landing buffers alternate between videos.  A peer starts pushing video n+2 into the buffer of video n
    as soon as ITS video n+1 is assembled, i.e. once this rank's last push of video n+1 has landed there - so the cache
    `finish` returned for video n is guaranteed intact only for reads this rank enqueued BEFORE its last `start` of video
    n+1.  Treat it as valid until this rank begins the next video; clone what must live longer.

    Memory: the two buffers are byte pools sized for the LARGEST video seen so far (`capacity` bytes each); a shorter
    or differently shaped video of at most that many bytes reuses them with its own strides, a larger one allocates a
    new pair 1.5x the need (mapped buffers are never freed before `P2PGroup.close()`, see SymmetricBuffer.close: the
    total mapped memory is bounded by ~3x the largest pair, not by the number of distinct shapes)."""

    def __init__(self, p2p, total: int, n_layers: int, Hkv: int, D: int, dtype, capacity_bytes: int = 0):
        self.p2p = p2p
        need = self._bytes(p2p.world, total, n_layers, Hkv, D, dtype)
        self.capacity = max(int(capacity_bytes), need)
        self.bufs = [p2p.symmetric(self.capacity), p2p.symmetric(self.capacity)]
        self.side = torch.cuda.Stream(device=p2p.device)
        self.gen, self.at = 1, 0
        self._sources: List[Tuple[torch.cuda.Event, torch.Tensor]] = []   # push sources whose copy kernel may still run
        self.begin_video(total, n_layers, Hkv, D, dtype)

    @staticmethod
    def _bytes(world, total, n_layers, Hkv, D, dtype) -> int:
        return 2 * n_layers * Hkv * world * int(total) * D * torch.empty((), dtype=dtype).element_size()

    def matches(self, total, n_layers, Hkv, D, dtype) -> bool:
        """Can a video of this shape land in the mapped buffers?"""
        return self._bytes(self.p2p.world, total, n_layers, Hkv, D, dtype) <= self.capacity

    def begin_video(self, total=None, n_layers=None, Hkv=None, D=None, dtype=None):
        if total is not None:
            if not self.matches(total, n_layers, Hkv, D, dtype):
                raise ValueError("p2p chunk gather: the video does not fit the mapped landing buffers")
            self.total, self.dtype = int(total), dtype
            self.shape = (2, n_layers, Hkv, self.p2p.world, self.total, D)
            self.es = torch.empty((), dtype=dtype).element_size()
        self.gen ^= 1
        self.at = 0
        self._release_sources()

    def start(self, k_new: List[torch.Tensor], v_new: List[torch.Tensor]):
        n_layers = len(k_new)
        Hkv, n, D = k_new[0].shape
        if self.at + n > self.total:
            raise ValueError(f"p2p chunk gather: {self.at + n} rows pushed, the landing buffer was sized for {self.total}")
        send = torch.empty((2, n_layers, Hkv, n, D), dtype=k_new[0].dtype, device=k_new[0].device)
        torch.stack(list(k_new), out=send[0])
        torch.stack(list(v_new), out=send[1])
        self.side.wait_stream(torch.cuda.current_stream(send.device))
        send.record_stream(self.side)
        row = D * self.es
        world, rank = self.p2p.world, self.p2p.rank
        self.bufs[self.gen].push(send, n * row, 2 * n_layers * Hkv, n * row, (rank * self.total + self.at) * row,
                                 world * self.total * row, stream=self.side)
        # the push kernel reads `send` on the side stream after this function has returned (record_stream above only
        # protects it under torch's caching allocator): the tensor is kept until an event behind its push has completed -
        # one or two chunks' rows at a time, not a second copy of everything the rank kept over the video
        ev = torch.cuda.Event()
        ev.record(self.side)
        self._release_sources()
        self._sources.append((ev, send))
        self.at += n

    def _release_sources(self):
        self._sources = [(e, t) for e, t in self._sources if not e.query()]

    def rows(self) -> int:
        return self.at

    def pushes(self) -> int:
        return self.bufs[self.gen].epoch

    def drop(self, lo: Optional[int] = None, hi: Optional[int] = None):
        """Abandon the per-chunk pushes of this video (`finalize` reaches that decision for all ranks at once and hands
        over the smallest / largest push count of the landing buffer over the ranks).  The ranks may have pushed
        DIFFERENT numbers of chunks (one skipped a `gather_chunk`, or `start` raised on one rank only): each waits for
        the smallest count - which every sender has reached, so nobody stalls into the timeout and no error is latched -
        and continues from the largest, so that the epochs agree again for the next video."""
        torch.cuda.current_stream(self.p2p.device).wait_stream(self.side)
        buf = self.bufs[self.gen]
        buf.resync(buf.epoch if lo is None else lo, buf.epoch if hi is None else hi)

    def finish(self) -> torch.Tensor:
        """-> [2, layers, Hkv, world * total, D], a view of the landing buffer (lifetime: class docstring)."""
        if self.at != self.total:
            raise ValueError(f"p2p chunk gather: {self.at} of {self.total} rows pushed")
        buf = self.bufs[self.gen]
        torch.cuda.current_stream(self.p2p.device).wait_stream(self.side)
        buf.wait()
        two, n_layers, Hkv, world, total, D = self.shape
        n = two * n_layers * Hkv * world * total * D * self.es
        return buf.local[:n].view(self.dtype).view(two, n_layers, Hkv, world * total, D)


def plan_frame_exchange(idx: torch.Tensor, T_own: int, world: int):
    """Who sends which kept frame where (DPSelect at ratio < 1 on frames sharded `T_own` per rank).

    idx: the global selection, [t] (sync) or [t, N] (per patch position), ascending along dim 0
    (visual_compression.py:135,169).  Because it ascends, the kept frames of patch n that live on rank r are
    the run j in [start[r, n], start[r+1, n]).  Returns
      start  [world+1, N'] int64 (N' = 1 for sync),
      cmax   int, the longest run (the padded block length every rank sends),
      local  [world, cmax, N'] int64: rank r's local frame index for slot (jl, n); padding repeats a valid frame,
      place  [t, N'] int64: row of the rank-ordered concatenation of the padded blocks that holds output (j, n).
    Pure index arithmetic on idx's device; identical on every rank because idx is."""
    i2 = idx if idx.ndim == 2 else idx[:, None]
    t, n_pos = i2.shape
    dev = i2.device
    bounds = torch.arange(world + 1, device=dev, dtype=torch.int64) * T_own
    start = torch.searchsorted(i2.t().contiguous(), bounds[None, :].expand(n_pos, -1).contiguous()).t().contiguous()
    cmax = max(1, int((start[1:] - start[:-1]).max().item()))
    jl = torch.arange(cmax, device=dev, dtype=torch.int64)[None, :, None]          # [1, cmax, 1]
    j = start[:-1, None, :] + jl                                                     # [world, cmax, N']
    valid = j < start[1:, None, :]
    src = torch.gather(i2[None].expand(world, -1, -1), 1, j.clamp(max=t - 1))       # global frame per slot
    r0 = torch.arange(world, device=dev, dtype=torch.int64)[:, None, None] * T_own
    local = torch.where(valid, src - r0, torch.zeros_like(src)).clamp_(0, T_own - 1)
    owner = torch.div(i2, T_own, rounding_mode="floor")                              # [t, N']
    jj = torch.arange(t, device=dev, dtype=torch.int64)[:, None]
    place = owner * cmax + (jj - torch.gather(start, 0, owner))
    return start, cmax, local.contiguous(), place.contiguous()


class PhaseTimer:
    """Where a sharded step spends its time, per rank: marks on the CURRENT stream (HIP events on a GPU, the host clock
    under gloo), one list per step.  A phase is the span between two consecutive marks as the launch stream sees it, so
    it includes what that stream waited for (a collective, the side stream's pushes, a host round trip that left it
    idle).  `sharded_video_step` marks  dpselect | blocks | offsets | rotate | assembly;  read `per_step_ms()` after a
    device synchronisation."""

    ORDER = ("dpselect", "blocks", "offsets", "rotate", "assembly")

    def __init__(self, device):
        self.cuda = torch.device(device).type == "cuda"
        self.steps: List[list] = []
        self.bytes = {}

    def begin(self):
        self.steps.append([])
        self.mark("start")

    def mark(self, name: str):
        if not self.steps:
            return
        if self.cuda:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            self.steps[-1].append((name, e))
        else:
            self.steps[-1].append((name, time.perf_counter()))

    def note_bytes(self, name: str, n: int):
        self.bytes[name] = int(n)

    def per_step_ms(self) -> dict:
        """{phase: mean ms per step} over the recorded steps (+ "step": start -> last mark)."""
        tot = {k: 0.0 for k in self.ORDER}
        tot["step"] = 0.0
        n = 0
        for marks in self.steps:
            if len(marks) < 2:
                continue
            n += 1
            for (_, a), (name, b) in zip(marks[:-1], marks[1:]):
                tot[name] = tot.get(name, 0.0) + (a.elapsed_time(b) if self.cuda else (b - a) * 1e3)
            a, b = marks[0][1], marks[-1][1]
            tot["step"] += a.elapsed_time(b) if self.cuda else (b - a) * 1e3
        return {k: v / max(1, n) for k, v in tot.items()}


def _mark(phases, name):
    if phases is not None:
        phases.mark(name)


# ---------------------------------------------------------------------------------------------------
# device path
# ---------------------------------------------------------------------------------------------------
def dpselect_sharded(frames_local: torch.Tensor, has_halo: bool, tgt_mem_len: int, window_size: int = 3,
                     sync: bool = False, group=None):
    """frames_local [1, T_loc (+1 halo frame in front), N, C]; every rank holds the same number of own frames.
    Returns (compressed, mask_flat_global, idx_global, dis_global).  At ratio 1.0 (tgt_mem_len == total frames)
    `compressed` is the rank's own frames [1, T_loc, N, C]; at ratio < 1 it is the full compressed bank
    [1, tgt_mem_len, N, C], assembled on every rank from one all-gather of the kept frames."""
    from . import _native as nv

    nv.require_device(frames_local)
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    x = frames_local[0].contiguous()
    Tl, N, Cc = x.shape
    dev = x.device
    dt = nv.dtype_code(x)
    with torch.cuda.device(dev):
        st = nv.stream()
        dis_l = torch.empty((Tl, N), dtype=torch.float32, device=dev)
        nv.check(nv.lib.rtk_dpselect_dis(nv.ptr(x), Tl, N, Cc, dt, nv.ptr(dis_l), st), "rtk_dpselect_dis")
        own = dis_l[1:] if has_halo else dis_l      # the halo frame only feeds the first own row
        counts = gather_counts(own.shape[0], dev, group)
        dis = all_gather_rows_ragged(own.contiguous(), group, counts)
        T = dis.shape[0]
        t = int(tgt_mem_len)
        if not 1 <= t <= T:
            raise ValueError(f"sharded DPSelect: tgt_mem_len {t} outside [1, {T}]")
        idx = torch.empty((t,) if sync else (t, N), dtype=torch.int64, device=dev)
        mask = torch.empty((t, N), dtype=torch.bool, device=dev)   # key-patch mask of the kept frames
        keys = torch.empty((2, T) if sync else (N, T), dtype=torch.float32, device=dev)
        nv.check(nv.lib.rtk_dpselect_select(nv.ptr(dis), T, N, t, int(window_size), int(bool(sync)), nv.ptr(idx),
                                            nv.ptr(mask), nv.ptr(keys), st), "rtk_dpselect_select")
        own_x = (x[1:] if has_halo else x).contiguous()
        T_own = own_x.shape[0]
        f0 = sum(counts[:rank])                      # first global frame of this rank
        if t == T:
            # ratio 1.0: every frame is kept at its own index -> the local output is a copy of the own frames
            idx_l = (idx[f0:f0 + T_own] - f0).contiguous()
            out = torch.empty((1, T_own, N, Cc), dtype=x.dtype, device=dev)
            nv.check(nv.lib.rtk_gather_frames(nv.ptr(own_x), T_own, N, Cc, dt, nv.ptr(idx_l), T_own,
                                              int(bool(sync)), nv.ptr(out), st), "rtk_gather_frames")
            return out, mask.flatten(), idx, dis
        # ratio < 1: own kept frames -> padded block -> all-gather -> output order
        if len(set(counts)) != 1:
            raise ValueError(f"sharded DPSelect at ratio < 1 needs equally many frames per rank, got {counts}")
        _, cmax, local, place = plan_frame_exchange(idx, T_own, world)
        mine = local[rank, :, 0].contiguous() if sync else local[rank].contiguous()
        block = torch.empty((cmax, N, Cc), dtype=x.dtype, device=dev)
        nv.check(nv.lib.rtk_gather_frames(nv.ptr(own_x), T_own, N, Cc, dt, nv.ptr(mine), cmax, int(bool(sync)),
                                          nv.ptr(block), st), "rtk_gather_frames")
        blocks = all_gather_rows(block, group)                       # [world * cmax, N, C] in rank order
        plc = place[:, 0].contiguous() if sync else place
        out = torch.empty((1, t, N, Cc), dtype=x.dtype, device=dev)
        nv.check(nv.lib.rtk_gather_frames(nv.ptr(blocks), world * cmax, N, Cc, dt, nv.ptr(plc), t, int(bool(sync)),
                                          nv.ptr(out), st), "rtk_gather_frames")
    return out, mask.flatten(), idx, dis


class ShardedPivotKV:
    """A rank's share of one video's PivotKV compression (see the module docstring)."""

    def __init__(self, config, group=None, first_start: int = 0, expected_rows: Optional[int] = None,
                 chunk_gather: Optional[ChunkGatherP2P] = None, reserve_tokens: Optional[int] = None):
        """expected_rows: rows this rank will keep per layer over the whole video (equal on all ranks); with the p2p
        transport it lets `gather_chunk` push rows to their final position (else the assembly happens at the end).
        chunk_gather: the ChunkGatherP2P of the previous video of the same shape, to reuse its mapped buffers."""
        from .longvideo_cache import PivotKVCache

        self.cache = PivotKVCache(config, reserve_tokens=reserve_tokens)   # capacity hint: kept rows + one chunk
        # the kept keys stay un-rotated (ids provisional) until `finalize` knows the block's temporal offset and rotates
        # them ONCE at their final ids - the same single rotation the sequential cache applies (reference :297-306)
        self.cache.defer_rerotation = bool(self.cache.pos_embed_reforge)
        # ... and the attention prologue scores the PRE-RoPE operands: the reference's own operands carry the rounding of a
        # rotation / un-rotation round trip at the chunk's FINAL ids (longvideo_cache.py:248-259), which a block does not
        # know before the offset scan - q0 / k0 never see an id, so every rank keeps what a single GPU with
        # prologue_operands="pre_rope" keeps, bit for bit, in every dtype
        self.cache.prologue_operands = "pre_rope"
        self.group = group
        self.first_start = first_start
        self.expected_rows = expected_rows
        self.chunk_gather = chunk_gather
        self._gather = None
        self._seen: List[int] = []

    def update(self, key_states, value_states, layer_idx, cache_kwargs):
        """PivotKVCache.update on this rank's block.  NOT attention-ready: while the re-rotation is deferred the
        returned keys are [un-rotated kept prefix | rotated current chunk] (the prefix takes its rotation in `finalize`,
        once its final ids are known), so only the current chunk's rows of the returned K may be attended to - which is
        all the sharded prefill does (scores are chunk-local, longvideo_cache.py:264)."""
        return self.cache.update(key_states, value_states, layer_idx, cache_kwargs)

    def update_pre_rope(self, *args, **kwargs):
        """PivotKVCache.update_pre_rope on this rank's block: with pre-RoPE operands the scores - and so the kept set -
        do not depend on the (provisional) ids at all, in any dtype."""
        return self.cache.update_pre_rope(*args, **kwargs)

    def gather_chunk(self):
        """Optional, after `after_forward()` of a chunk: start the all-gather of the rows that chunk kept (all
        layers, one collective) so that it overlaps the next chunk's scoring; `finalize(assemble=True)` then only
        waits, rotates every rank's segment to its true temporal position and exchanges the ids."""
        cache = self.cache
        cache.after_forward()
        kc, vc = cache.key_cache, cache.value_cache
        if self._gather is None:
            p2p = _p2p_for(kc[0], self.group)
            if p2p is None:
                self._gather = ChunkGather(self.group)
            elif self.expected_rows is None:
                return          # p2p pushes need their final position: assemble at the end instead
            else:
                _, Hkv, _, D = kc[0].shape
                cg = self.chunk_gather
                if cg is None or not cg.matches(self.expected_rows, len(kc), Hkv, D, kc[0].dtype):
                    need = ChunkGatherP2P._bytes(p2p.world, self.expected_rows, len(kc), Hkv, D, kc[0].dtype)
                    cg = ChunkGatherP2P(p2p, self.expected_rows, len(kc), Hkv, D, kc[0].dtype,
                                        capacity_bytes=0 if cg is None else need * 3 // 2)
                else:
                    cg.begin_video(self.expected_rows, len(kc), Hkv, D, kc[0].dtype)
                self._gather = self.chunk_gather = cg
            self._seen = [0] * len(kc)
        ks, vs = [], []
        for layer in range(len(kc)):
            ks.append(kc[layer][0, :, self._seen[layer]:])
            vs.append(vc[layer][0, :, self._seen[layer]:])
            self._seen[layer] = kc[layer].shape[2]
        self._gather.start(ks, vs)

    def finalize(self, inv_freq: torch.Tensor, mrope_section: Optional[List[int]], assemble: bool = True,
                 attention_scaling: Optional[float] = None, phases: Optional["PhaseTimer"] = None):
        """Exchange offsets, shift this rank's ids to their true temporal position, rotate the kept keys - un-rotated
        until now - ONCE at those final ids (what the sequential cache does per chunk, reference :297-306: same tables,
        same roundings, so a block's keys carry the bits they would have had in a single-GPU run), and optionally
        all-gather the full compressed cache.  Returns (keys, values, position_ids) lists per layer.
        attention_scaling: of the rotary module the keys belong to (default: the one the cache's updates were given)."""
        from . import _native as nv

        cache = self.cache
        cache.after_forward()
        n_layers = len(cache.position_cache)
        dev = cache.position_cache[0].device
        last = torch.stack([pc.reshape(-1, pc.shape[-1])[0, -1] for pc in cache.position_cache])   # [layers]
        table = exchange_temporal_offsets(last, self.first_start, self.group, all_ranks=True)      # [world, layers]
        delta = table[dist.get_rank(self.group)]                                                   # [layers]
        _mark(phases, "offsets")
        sec = (C.c_int * len(mrope_section))(*mrope_section) if mrope_section else None
        nsec = len(mrope_section) if mrope_section else 0
        if attention_scaling is None:   # (a cache that never saw a compressed update has no batch: plain RoPE scaling)
            b = cache._batch
            rot = b.rot if b is not None else None
            attention_scaling = rot.scaling if rot is not None else float(
                getattr(b.rotary_emb_fn if b is not None else None, "attention_scaling", 1.0))
        keys, values, pos = [], [], []
        with torch.cuda.device(dev):
            st = nv.stream()
            inv = inv_freq.to(device=dev, dtype=torch.float32).contiguous()
            for layer in range(n_layers):
                k, v = cache.key_cache[layer], cache.value_cache[layer]
                pc = cache.position_cache[layer]
                if pc.ndim == 3:
                    pc[0] += delta[layer]
                else:
                    pc += delta[layer]
                keys.append(k)
                values.append(v)
                pos.append(pc)

            def rotate_own():
                """this rank's kept rows of every layer, in place in the layers' cache buffers, at their final ids"""
                for layer in range(n_layers):
                    st_l = cache._layers[layer]
                    kbuf, k = st_l.k, keys[layer]
                    P = 3 if pos[layer].ndim == 3 else 1
                    nv.check(nv.lib.rtk_rope_rotate_rows(
                        nv.ptr(kbuf), 0, kbuf.shape[2] * kbuf.shape[3], 1, k.shape[1], k.shape[2], k.shape[3],
                        nv.dtype_code(k), nv.ptr(st_l.pos), 0, st_l.pos.shape[1], P, nv.ptr(inv), attention_scaling, sec, nsec,
                        nv.round_mode(k.dtype), st), "rtk_rope_rotate_rows")

            g = self._gather
            self._gather = None
            # ONE decision for all ranks, taken before anybody waits on anything: the overlapped per-chunk gathers are
            # used only if every rank started one for every chunk (its rows == the rows it kept) and every rank kept
            # equally many rows; otherwise every rank that started any abandons them (ChunkGatherP2P.drop re-aligns the
            # push epochs, so ranks that pushed different numbers of chunks neither stall nor latch a timeout) and the
            # cache is assembled by the padded gather at the end
            counts = gather_counts(keys[0].shape[2], dev, self.group)
            mine_ok = g is not None and assemble and g.rows() == keys[0].shape[2] and len(set(counts)) == 1
            flags = _gather_stack(torch.tensor([int(mine_ok), g.pushes() if g is not None else -1], dtype=torch.int64,
                                               device=dev), self.group)
            use_all = bool(flags[:, 0].min().item())
            if not use_all:
                pushes = [int(x) for x in flags[:, 1].tolist()]
                started = [x for x in pushes if x >= 0]
                if started:
                    # `flags` is the same tensor on every rank, so every rank takes the same branch here - before anybody
                    # enters another collective.  Two mismatches cannot be repaired after the fact and raise on ALL ranks:
                    # a rank without per-chunk gathers among ranks that started some (its peers' pushes would wait for an
                    # epoch it never publishes / its landing buffers keep another video's epoch), and unequal numbers of
                    # all-gathers on the collective transport (the extra one has no partner).
                    if len(started) != len(pushes):
                        raise RuntimeError(f"per-chunk gathers were started on some ranks only (pushes per rank: {pushes}): "
                                           "every rank must call gather_chunk for every chunk, or none")
                    if min(started) != max(started) and _P2P.get(self.group) is None:
                        raise RuntimeError(f"ranks started between {min(started)} and {max(started)} per-chunk all-gathers: "
                                           "unmatched collectives")
                    g.drop(min(started), max(started))
                g = None
            if assemble and g is not None:
                kv = g.finish()                                   # [2, layers, Hkv, world*n, D], keys still un-rotated
                pos = all_gather_ids(pos, self.group, counts)     # the final ids of every rank's rows
                _mark(phases, "assembly")                         # (the rows travelled beside the blocks: this is the wait)
                if phases is not None:
                    phases.note_bytes("assembly_rows_received", kv.numel() * kv.element_size() * (len(counts) - 1) // len(counts))
                # every row of every layer -> rotated at its final ids, ONE launch over the assembled cache
                P = 3 if pos[0].ndim == 3 else 1
                rows = kv.shape[3]
                ids_all = torch.stack([p_.reshape(P, rows) for p_ in pos]).contiguous()      # [layers, P, world*n]
                nv.check(nv.lib.rtk_rope_rotate_rows(C.c_void_p(kv[0].data_ptr()), kv.stride(1), kv.stride(2), n_layers,
                                                     kv.shape[2], rows, kv.shape[4], nv.dtype_code(kv), nv.ptr(ids_all),
                                                     P * rows, rows, P, nv.ptr(inv), attention_scaling, sec, nsec,
                                                     nv.round_mode(kv.dtype), st), "rtk_rope_rotate_rows")
                keys = [kv[0, layer][None] for layer in range(n_layers)]
                values = [kv[1, layer][None] for layer in range(n_layers)]
                _mark(phases, "rotate")
                p2p = _P2P.get(self.group)
                if p2p is not None:
                    p2p.check()   # synchronises; a wait that timed out left a partly filled landing buffer: raise
            else:
                rotate_own()
                _mark(phases, "rotate")
                if assemble:
                    own = sum(k.numel() * k.element_size() + v.numel() * v.element_size() for k, v in zip(keys, values))
                    keys, values, pos = all_gather_caches(keys, values, pos, self.group)
                    _mark(phases, "assembly")
                    if phases is not None:
                        phases.note_bytes("assembly_rows_received", own * (len(counts) - 1))
        return keys, values, pos


# ---------------------------------------------------------------------------------------------------
# bench.py --gpus N  (strong scaling: one video, chunks sharded over the ranks)
# ---------------------------------------------------------------------------------------------------
def sharded_video_step(frames, has_halo: bool, T: int, c0: int, c1: int, layers: int, pool, pos_base, rotary, overlap: bool,
                       group=None, state: Optional[dict] = None, inputs=None, pre_rope: bool = False):
    """One rank's share of one video (what `bench.py --gpus N` times and tests/mp_sharded_gpu.py checks): DPSelect on
    the rank's frames with the distance rows all-gathered, PivotKV on chunks [c0, c1) at provisional temporal ids, then
    offsets + cache assembly.  `overlap`: the rows a chunk kept leave in one asynchronous all-gather right after its
    flush (needs equally many chunks on every rank).  `state`: a dict the caller keeps between videos (the p2p
    transport reuses its mapped landing buffers through it).  `inputs(c, layer, pos) -> (q, k, v)`: the layer's rotated
    q / k at the ids `pos` the update will see (what a model produces; tests/mp_sharded_gpu.py rotates fixed contents
    with it).  Without it the resident pool set is taken AS the rotated input, whatever the ids: right for timing, but
    a later block then scores different content than the single-GPU run (its scores, kept set and K differ; sizes agree).
    `pre_rope`: `inputs` returns the PRE-RoPE projections and every update is the attention prologue
    (PivotKVCache.update_pre_rope): the scores never see the ids, so the sharded and the sequential run keep the same
    tokens in every dtype, and with the single rotation of `finalize` the same key bits.
    Returns (retained tokens of this rank, (keys, values, ids))."""
    import bench as B

    L = B.FRAMES_PER_CHUNK * B.N_PATCH
    phases = (state or {}).get("phases")     # a PhaseTimer the caller wants filled (bench.py --gpus N)
    watch = (state or {}).get("watch")       # debugging aid: called with a label after every chunk / the assembly
    if phases is not None:
        phases.begin()
    out, mask, idx, dis = dpselect_sharded(frames, has_halo, T, 3, sync=False, group=group)
    _mark(phases, "dpselect")
    keep = max(1, int(B.RATIO * L))
    sh = ShardedPivotKV(B.make_cache_config(layers), group=group, expected_rows=(c1 - c0) * keep if overlap else None,
                        chunk_gather=(state or {}).get("chunk_gather"), reserve_tokens=(c1 - c0) * keep + L)
    cache = sh.cache
    q_rot = None
    for ci, c in enumerate(range(c0, c1)):
        cache.keypatches_mask_chunk = mask[c * L:(c + 1) * L]
        cache.kvcache_compression = True
        pos = pos_base[ci].clone()
        for layer in range(layers):
            if pre_rope:
                q0, k0, v = inputs(c, layer, pos)
                if q_rot is None or q_rot.shape != q0.shape or q_rot.dtype != q0.dtype:
                    q_rot = torch.empty_like(q0)   # where the rotated queries go: one scratch per video, not one per update
                if cache.update_pre_rope(q0, k0, v, layer, pos, rotary, B.MROPE, query_out=q_rot) is None:
                    raise RuntimeError("update_pre_rope declined a video chunk of the sharded step")
                continue
            cache.shift_temporal_ids_(pos, layer)       # block-local ids start at 0 (provisional)
            q, k, v = pool[(c * layers + layer) % len(pool)] if inputs is None else inputs(c, layer, pos)
            cache.update(k, v, layer, {"query_states": q, "position_ids": pos, "rotary_emb": rotary,
                                       "mrope_section": B.MROPE, "shift_next_position_ids": True})   # as the Qwen2-VL patch does
        cache.after_forward()
        if overlap:
            sh.gather_chunk()   # this chunk's kept rows leave now, beside the next chunk's scoring
        if watch is not None:
            watch(f"after chunk {c} of the block (its push started: {overlap})")
    _mark(phases, "blocks")
    keys, values, pos = sh.finalize(rotary.inv_freq, B.MROPE, assemble=True, phases=phases)
    if watch is not None:
        watch("after finalize")
    if state is not None:
        state["chunk_gather"] = sh.chunk_gather
    return (c1 - c0) * layers * keep, (keys, values, pos)


def _rotate_at(x0: torch.Tensor, pos: torch.Tensor, rotary, mrope_section):
    """What the model's attention does before it calls the cache (third-party HF formula): x0 [1, H, L, D] rotated at
    the ids `pos` with the module's cos / sin (M-RoPE sections merged when given)."""
    cos, sin = rotary(x0, pos)
    if mrope_section:
        sec = list(mrope_section) * 2
        cos = torch.cat([m[i % 3] for i, m in enumerate(cos.split(sec, dim=-1))], dim=-1).unsqueeze(1)
        sin = torch.cat([m[i % 3] for i, m in enumerate(sin.split(sec, dim=-1))], dim=-1).unsqueeze(1)
    else:
        cos, sin = cos.unsqueeze(1), sin.unsqueeze(1)
    D = x0.shape[-1]
    return x0 * cos + torch.cat((-x0[..., D // 2:], x0[..., : D // 2]), dim=-1) * sin


class _PoolWatch:
    """RETAKE_VERIFY_POOL_WATCH=1 (debugging aid of the multi-rank check): did a LIVE input tensor of this process change,
    when, where and into what?  One int64 fingerprint (sum of the bit patterns) per resident tensor at construction;
    every call recomputes them and, on the first difference, regenerates that tensor from its seed, reports the damaged
    256-byte rows (count, runs, first values), the tensor's address beside the ranges this process has MAPPED from its
    peers, whether the guard regions of its own landing buffers are intact (RETAKE_P2P_GUARD=1) - and raises."""

    def __init__(self, rank, world, pool, frames_all, mask, layers, n_chunks, dname, dev, td, group):
        self.rank, self.world, self.pool, self.layers, self.dname, self.dev, self.td, self.group = rank, world, pool, layers, dname, dev, td, group
        self.n_chunks = n_chunks
        self.extra = [("frames", frames_all), ("mask", mask.view(torch.uint8))]
        self.calls = 0
        self.ref = self._sums()

    @staticmethod
    def _bits(t):
        t = t.contiguous() if t.is_contiguous() else t.transpose(1, 2).contiguous()
        v = t.view(torch.int16) if t.element_size() == 2 else (t.view(torch.int32) if t.element_size() == 4 else t.view(torch.uint8))
        return v.sum(dtype=torch.int64)

    def _sums(self):
        s = [self._bits(t) for trip in self.pool for t in trip] + [self._bits(t) for _, t in self.extra]
        return torch.stack(s).cpu()

    def __call__(self, label):
        self.calls += 1
        now = self._sums()
        bad = (now != self.ref).nonzero().flatten().tolist()
        p2p = _P2P.get(self.group)
        guards = p2p.guards_intact() if p2p is not None else None
        if not bad and guards in (None, True):
            return
        import bench as B

        torch.cuda.synchronize(self.dev)
        again = self._sums()                 # the same fingerprints, computed once more
        still = (again != self.ref).nonzero().flatten().tolist()
        lines = [f"POOLWATCH rank {self.rank} ({self.dname}, {self.n_chunks} chunks): fingerprints of {len(bad)} resident tensor(s) "
                 f"differ, first seen {label} (look {self.calls}); recomputed at once: {len(still)} differ; guard regions intact: {guards}"]
        persistent = False
        for i in bad[:3]:
            if i >= 3 * len(self.pool):
                lines.append(f"  {self.extra[i - 3 * len(self.pool)][0]}: fingerprint differed")
                continue
            si, j = divmod(i, 3)
            t = self.pool[si][j]
            fresh = B.pool_set(si, self.dev, self.td, projection_layout=True)[j]
            mem = t.transpose(1, 2).contiguous().view(-1)         # the projection layout's memory order [L, H, D]
            ref = fresh.transpose(1, 2).contiguous().view(-1)
            D = t.shape[-1]
            rows = (mem.view(-1, D) != ref.view(-1, D)).any(dim=1).nonzero().flatten()
            if rows.numel() == 0:
                # the tensor IS what its seed says: it was the READ that went wrong (the reduction that fingerprinted it saw
                # other bytes, or lost its scratch), not the tensor
                lines.append(f"  pool set {si} (chunk {si // self.layers}, layer {si % self.layers}) {'qkv'[j]}: fingerprint was "
                             f"{int(now[i])}, is {int(again[i])} on recomputation, reference {int(self.ref[i])}; the tensor EQUALS its "
                             f"regeneration element by element -> a transient wrong READ, nothing was written")
                continue
            persistent = True
            r0, r1, n = int(rows[0]), int(rows[-1]), int(rows.numel())
            runs = int((rows[1:] - rows[:-1] != 1).sum().item()) + 1
            row_b = D * t.element_size()
            lines.append(f"  pool set {si} (chunk {si // self.layers}, layer {si % self.layers}) {'qkv'[j]}: {n} of {mem.numel() // D} rows of "
                         f"{row_b} B differ, rows {r0}..{r1} in {runs} run(s); tensor at {t.data_ptr():#x} (+{r0 * row_b:#x}), "
                         f"{t.numel() * t.element_size()} B; first damaged row now {mem.view(-1, D)[r0, :6].float().tolist()} "
                         f"was {ref.view(-1, D)[r0, :6].float().tolist()}; all-zero damaged rows "
                         f"{int((mem.view(-1, D)[rows] == 0).all(dim=1).sum())}")
        if p2p is not None:
            rng = p2p.mapped_ranges()
            lines.append("  ranges mapped from peers (buffer, peer, address, bytes): "
                         + ", ".join(f"({b},{r},{a:#x},{nb})" for b, r, a, nb, _ in rng if r != self.rank)[:1500])
        print("\n".join(lines), flush=True)
        if persistent or still or guards is False:
            raise AssertionError(lines[0])
        self.transients = getattr(self, "transients", 0) + 1


def verify_sharded_equals_sequential(rank: int, world: int, dev, rotary, layers: int = 2, chunk_counts=None,
                                     group=None, state: Optional[dict] = None, log=None, dtypes=("fp32", "bf16")) -> dict:
    """Equality of the sharded and the sequential compression, over the transport `group` is configured for (RCCL
    collectives, or the p2p pushes after `enable_p2p`) and through the very function `bench.py --gpus N` times
    (`sharded_video_step`), in the parity dtype AND in the dtype the bench times.  For every dtype and every chunk count
    in `chunk_counts` (default: 2 * world - even blocks, per-chunk overlapped gathers - and 2 * world + 1 - ragged blocks,
    padded assembly at the end) every rank

      * builds the SEQUENTIAL cache of the whole small video on its own (`layers` layers, bench.py's deterministic
        tensors as the pre-RoPE projections, through PivotKVCache.update_pre_rope at the ids a single-GPU run sees), then
      * compresses its block through `sharded_video_step` (same projections, the block's PROVISIONAL ids) and compares
        the ASSEMBLED cache with the sequential one: position ids exact, V exact, K EXACT - bit patterns, in bf16 too:
        the scores are computed from the un-rotated operands (they never see an id) and `finalize` rotates every kept
        key once, at its final id, with the tables and roundings the sequential flush uses.

    Raises AssertionError on any mismatch (on the rank that sees it); returns a summary dict."""
    import bench as B
    from . import longvideo_cache as lc
    from . import visual_compression as vc

    L = B.FRAMES_PER_CHUNK * B.N_PATCH
    keep = max(1, int(B.RATIO * L))
    state = {} if state is None else state
    chunk_counts = tuple(chunk_counts) if chunk_counts is not None else (2 * world, 2 * world + 1)
    cases = []
    for dname in dtypes:
      td = B.TORCH_DTYPE[dname]
      for n_chunks in chunk_counts:
        T = n_chunks * B.FRAMES_PER_CHUNK
        pool = [B.pool_set(i, dev, td, projection_layout=True) for i in range(n_chunks * layers)]

        def inputs(c, l, pos):   # what q_proj / k_proj / v_proj hand the attention patch
            return pool[(c * layers + l) % len(pool)]

        frames_all = torch.cat([B.chunk_frames(c, dev, td) for c in range(n_chunks)])[None]
        _, mask = vc.memory_bank_compress_keyframe(frames_all, T, 3, sync=False)
        def sequential():
            q_rot = torch.empty_like(pool[0][0])   # where the rotated queries go (one scratch per build)
            seq = lc.build_kvcache(B.make_cache_config(layers))
            seq.prologue_operands = "pre_rope"    # what the blocks score (ShardedPivotKV): operands that never see an id
            for c in range(n_chunks):
                seq.keypatches_mask_chunk = mask[c * L:(c + 1) * L]
                seq.kvcache_compression = True
                pos = B.chunk_position_ids(c, dev)
                for l in range(layers):
                    q0, k0, v = inputs(c, l, pos)
                    if seq.update_pre_rope(q0, k0, v, l, pos, rotary, B.MROPE, query_out=q_rot) is None:
                        raise AssertionError("update_pre_rope declined a chunk of the sequential reference run")
                seq.after_forward()
            return seq

        if os.environ.get("RETAKE_VERIFY_POOL_WATCH") == "1":
            # debugging aid (profiles/r15_p2p_hunt.log): fingerprints of every resident input tensor of this rank, taken now
            # and looked at again after every push epoch - the first label at which one differs, which tensor, where in it
            # and what was written there
            state["watch"] = _PoolWatch(rank, world, pool, frames_all, mask, layers, n_chunks, dname, dev, td, group)
        seq = sequential()
        if state.get("watch") is not None:
            state["watch"]("after the sequential build")
        if os.environ.get("RETAKE_VERIFY_SEQ_TWICE") == "1":   # debugging aid: is the single-GPU build itself reproducible here?
            again = sequential()
            for l in range(layers):
                for what, a, b in (("ids", seq.position_cache[l], again.position_cache[l]),
                                   ("V", seq.value_cache[l], again.value_cache[l]), ("K", seq.key_cache[l], again.key_cache[l])):
                    if not torch.equal(a, b):
                        d = (a != b).reshape(-1, a.shape[-1]) if what == "ids" else (a != b).any(dim=3).any(dim=1)
                        raise AssertionError(f"rank {rank} layer {l} {dname}: two SEQUENTIAL builds of the same video differ in {what}; "
                                             f"differing entries per kept chunk {d.any(0).reshape(-1, keep).sum(1).tolist()}")
            del again
        blocks = shard_chunks(n_chunks, world)
        c0, c1 = blocks[rank]
        even = len({b - a for a, b in blocks}) == 1
        halo = c0 > 0 and c1 > c0
        parts = ([B.chunk_frames(c0 - 1, dev, td)[-1:]] if halo else []) + [B.chunk_frames(c, dev, td) for c in range(c0, c1)]
        fr = torch.cat(parts)[None] if parts else torch.empty((1, 0, B.N_PATCH, B.C_EMB), dtype=td, device=dev)
        pos_base = [B.chunk_position_ids(c, dev) for c in range(c0, c1)]
        _, (keys, values, pos) = sharded_video_step(fr, halo, T, c0, c1, layers, pool, pos_base, rotary, even, group=group,
                                                    state=state, inputs=inputs, pre_rope=True)
        state.pop("watch", None)
        if _P2P.get(group) is not None:
            _P2P.get(group).check()   # a bounded wait that ran out is reported as that, not as the mismatch it leaves behind
        for l in range(layers):
            assert keys[l].shape[2] == n_chunks * keep, (keys[l].shape, n_chunks * keep)
            if not torch.equal(pos[l], seq.position_cache[l]):
                bad = (pos[l] != seq.position_cache[l]).reshape(-1, pos[l].shape[-1]).any(0).reshape(-1, keep).sum(1)
                if os.environ.get("RETAKE_VERIFY_LOOK_AGAIN") == "1":   # debugging aid: late, or never?  ids only, or the rows too?
                    torch.cuda.synchronize(dev)
                    time.sleep(2.0)
                    torch.cuda.synchronize(dev)
                    again = (pos[l] != seq.position_cache[l]).reshape(-1, pos[l].shape[-1]).any(0).reshape(-1, keep).sum(1)
                    vrows = (values[l] != seq.value_cache[l]).any(dim=3).any(dim=1)[0].reshape(-1, keep).sum(1)
                    c = int(bad.nonzero()[0])

                    def bits(t):
                        return int(t.contiguous().view(torch.int16 if t.element_size() == 2 else torch.int32).to(torch.int64).sum())

                    fresh = B.pool_set((c * layers + l) % len(pool), dev, td, projection_layout=True)
                    sums = {"q k v as used": [bits(t) for t in inputs(c, l, None)], "q k v regenerated": [bits(t) for t in fresh],
                            "frames as used": bits(frames_all[0, c * B.FRAMES_PER_CHUNK:(c + 1) * B.FRAMES_PER_CHUNK]),
                            "frames regenerated": bits(B.chunk_frames(c, dev, td)), "mask": int(mask[c * L:(c + 1) * L].sum())}
                    print(f"LOOK rank {rank} chunk {c} layer {l}: {sums}", flush=True)
                    got = pos[l].reshape(-1, pos[l].shape[-1])[:, c * keep:(c + 1) * keep]
                    want = seq.position_cache[l].reshape(-1, pos[l].shape[-1])[:, c * keep:(c + 1) * keep]
                    raise AssertionError(f"rank {rank} layer {l} {dname}: ids differ; wrong ids per kept chunk "
                                         f"{[(i, int(x)) for i, x in enumerate(bad.tolist()) if x]}; two seconds later "
                                         f"{[(i, int(x)) for i, x in enumerate(again.tolist()) if x]}; V rows wrong "
                                         f"{[(i, int(x)) for i, x in enumerate(vrows.tolist()) if x]}; chunk {c}: got "
                                         f"{got[:, :6].tolist()} .. {got[:, -3:].tolist()} want {want[:, :6].tolist()} .. {want[:, -3:].tolist()}")
                raise AssertionError(f"rank {rank} layer {l} {dname}: ids differ; wrong ids per kept chunk {bad.tolist()}; first "
                                     f"rows {pos[l].reshape(-1, pos[l].shape[-1])[0, ::keep].tolist()} vs "
                                     f"{seq.position_cache[l].reshape(-1, pos[l].shape[-1])[0, ::keep].tolist()}")
            if not torch.equal(values[l], seq.value_cache[l]):   # say where: differing rows per kept chunk of the assembled rows
                rows = (values[l] != seq.value_cache[l]).any(dim=3).any(dim=1)[0].reshape(-1, keep).sum(1)
                if os.environ.get("RETAKE_VERIFY_LOOK_AGAIN") == "1":   # debugging aid: late, or never?
                    bad = (values[l] != seq.value_cache[l]).any(dim=3).any(dim=1)[0].nonzero().flatten()
                    torch.cuda.synchronize(dev)
                    time.sleep(2.0)
                    torch.cuda.synchronize(dev)
                    again = (values[l] != seq.value_cache[l]).any(dim=3).any(dim=1)[0].reshape(-1, keep).sum(1)
                    elems = (values[l] != seq.value_cache[l])[0, :, bad[0]].sum(-1).tolist()
                    raise AssertionError(f"rank {rank} layer {l} {dname}: V differs; wrong rows per kept chunk "
                                         f"{[(i, int(x)) for i, x in enumerate(rows.tolist()) if x]}; two seconds later "
                                         f"{[(i, int(x)) for i, x in enumerate(again.tolist()) if x]}; first wrong rows "
                                         f"{bad[:12].tolist()} (+{int(bad.numel())}), wrong elements per head in the first {elems}")
                raise AssertionError(f"rank {rank} layer {l} {dname}: V differs; wrong rows per kept chunk {rows.tolist()}; "
                                     f"NaN entries {int(torch.isnan(values[l].float()).sum())}, all-zero rows "
                                     f"{int((values[l] == 0).all(dim=3).all(dim=1).sum())}")
            if not torch.equal(keys[l], seq.key_cache[l]):   # say where: per kept chunk of the assembled rows
                d = (keys[l].float() - seq.key_cache[l].float()).abs().amax(dim=(0, 1, 3)).reshape(-1, keep).amax(dim=1)
                raise AssertionError(f"rank {rank} layer {l} {dname}: K differs; max |diff| per kept chunk {d.tolist()}")
        a = B.cache_checksum(keys, values, pos)
        b = B.cache_checksum([seq.key_cache[l] for l in range(layers)], [seq.value_cache[l] for l in range(layers)],
                             seq.position_cache)
        assert a["ids_sum"] == b["ids_sum"] and a["v_bits_sum"] == b["v_bits_sum"] and a["tokens_per_layer"] == b["tokens_per_layer"]
        assert a["k_abs_sum"] == b["k_abs_sum"]
        p2p = _P2P.get(group)
        if p2p is not None:
            p2p.check()
        torch.cuda.synchronize(dev)
        dist.barrier(group=group)
        cases.append({"dtype": dname, "chunks": n_chunks, "blocks": blocks, "overlapped_gathers": even})
        if log is not None and rank == 0:
            log(f"{dname}: chunks {n_chunks} on {world} rank(s): blocks {blocks}, overlapped gathers {even}: assembled == "
                f"sequential, bit for bit")
        del seq, pool, frames_all, keys, values, pos
    # every rank passed its own comparison (a failing rank raised and took the job down with it); make it explicit
    ok = torch.ones(1, dtype=torch.int64, device=dev if dist.get_backend(group) != "gloo" else "cpu")
    dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
    assert int(ok.item()) == 1
    return {"equal": True, "dtype": list(dtypes), "layers": layers, "cases": cases, "max_rel_k_diff": 0.0,
            "checked": "assembled cache == sequential cache on every rank, pre-RoPE operands, one rotation at the final ids: "
                       "ids, V and K bit patterns equal in every listed dtype"}


def measure_sharded(args, rank: int, world: int, dev, transport: str, share: bool, steps: int, warmup: int,
                    verify: bool = True) -> Optional[dict]:
    """The timed sharded run on an initialised process group: -> the report dict on rank 0, None elsewhere.
    Besides the whole-job rate the report says where a step's time goes on the ranks (`phase_ms`: [max, min] over the
    ranks of the mean ms per step of each PhaseTimer phase, of the host's wall time per step and of the idle time at the
    closing barrier) - what the first real multi-GPU run needs to explain a shortfall."""
    import bench as B
    from . import _native as nv

    red_dev = torch.device("cpu") if share else dev     # gloo reduces host tensors
    p2p = _P2P.get(None)
    tdtype = B.TORCH_DTYPE[args.dtype]
    T = args.frames
    n_chunks = T // B.FRAMES_PER_CHUNK
    L = B.FRAMES_PER_CHUNK * B.N_PATCH
    blocks = shard_chunks(n_chunks, world)
    c0, c1 = blocks[rank]
    even = len({b - a for a, b in blocks}) == 1     # else: padded assembly at the end instead of per-chunk gathers
    # the synthetic inputs are functions of the chunk / update index (bench.chunk_frames, bench.pool_set), so every rank
    # regenerates its share - halo frame included - and any world size compresses the very same video
    halo = 1 if c0 > 0 and c1 > c0 else 0
    parts = ([B.chunk_frames(c0 - 1, dev, tdtype)[-1:]] if halo else []) + [B.chunk_frames(c, dev, tdtype) for c in range(c0, c1)]
    frames = torch.cat(parts)[None] if parts else torch.empty((1, 0, B.N_PATCH, B.C_EMB), dtype=tdtype, device=dev)
    pool = [B.pool_set(i, dev, tdtype) for i in range(min(args.pool, n_chunks * args.layers))]
    pos_base = [B.chunk_position_ids(c, dev) for c in range(c0, c1)]
    rotary = B.Rotary(dev)

    state = {}

    def step():
        return sharded_video_step(frames, halo == 1, T, c0, c1, args.layers, pool, pos_base, rotary, even, state=state)

    # communicator set-up (RCCL builds its rings / the p2p scratch is mapped on the first exchange) is not part of a step:
    # run one tiny exchange before anything is timed, whatever --warmup says
    _gather_stack(torch.zeros(4, dtype=torch.int64, device=dev))
    # Before anything is timed: is the cache this transport assembles the cache one GPU builds?  (tests/mp_sharded_gpu.py's
    # check, in process, over the transport of the timed loop.)  A mismatch raises: no value is printed for a wrong path.
    verdict = None
    if verify:
        verdict = verify_sharded_equals_sequential(rank, world, dev, rotary, layers=2, state=state)
        torch.cuda.empty_cache()
    for _ in range(warmup):
        step()
    ids = nv.profile_kernel_ids()   # HIP events around the dominant kernels, on their launch stream, in the timed region
    nv.check(nv.lib.rtk_profile_reset(), "profile_reset")
    nv.check(nv.lib.rtk_profile_enable_mask((1 << ids["score_pass1"]) | (1 << ids["score_pass2"])), "profile_enable")
    phases = state["phases"] = PhaseTimer(dev)
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    retained = 0
    for _ in range(steps):
        r, (keys, values, pos) = step()
        retained += r
    torch.cuda.synchronize()
    t_done = time.perf_counter()
    dist.barrier()
    torch.cuda.synchronize()
    t_end = time.perf_counter()
    state.pop("phases")
    dt = torch.tensor([t_end - t0], dtype=torch.float64, device=red_dev)
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    tot = torch.tensor([float(retained)], dtype=torch.float64, device=red_dev)
    dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    dt = float(dt.item())
    nv.check(nv.lib.rtk_profile_enable(0), "profile_enable")
    kern = {k: {"launches": n, "avg_us": ms / n * 1e3, "total_ms": ms} for k, (n, ms) in nv.profile_read().items()}
    # per-phase times of this rank (mean per step) -> [max, min] over the ranks
    pm = phases.per_step_ms()
    pm["finalize"] = pm["offsets"] + pm["rotate"] + pm["assembly"]
    pm["host_wall"] = (t_done - t0) / steps * 1e3
    pm["barrier_idle"] = (t_end - t_done) / steps * 1e3
    names = sorted(pm)
    hi = torch.tensor([pm[k] for k in names], dtype=torch.float64, device=red_dev)
    lo = hi.clone()
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    phase_ms = {k: [float(f"{a:.4g}"), float(f"{b:.4g}")] for k, a, b in zip(names, hi.tolist(), lo.tolist())}
    # untimed; the token count equals the N = 1 line's, the sums only at N = 1 (see sharded_video_step on `inputs`)
    checksum = B.cache_checksum(keys, values, pos) if rank == 0 else None
    if p2p is not None:
        p2p.check()   # a bounded wait that gave up would have left garbage: fail the run instead
    if rank != 0:
        return None
    return {
        "metric": "frames/sec through DPSelect+PivotKV @2048 frames; retained-KV-tokens/sec",
        "value": T * steps / dt, "unit": "frames/s", "n_gpus": world, "steps": steps,
        "warmup": warmup, "ms_per_step": dt / steps * 1e3, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "retained_kv_tokens_per_s": float(tot.item()) / dt,
        "config": {"workload": f"BASELINE configs[3]: {T}-frame video sharded by frame chunk over {world} GPU(s), "
                               f"{n_chunks} chunks x {args.layers} layers, L={L}",
                   "workload_detail": f"Qwen2-VL-7B geometry: DPSelect (distance rows all-gathered) + PivotKV 4x; offsets + "
                                      f"whole-cache all-gather over "
                                      f"{('RCCL' if transport == 'rccl' else 'gloo, host-staged (tests)') if p2p is None else 'direct xGMI pushes (retake/p2p.py)'}",
                   "frames": T, "chunks": n_chunks, "layers": args.layers, "chunk_tokens": L,
                   "parallelism": f"chunk-sharded x{world}", "transport": transport,
                   "assembled_cache_tokens": int(keys[0].shape[2])},
        "cache_checksum": checksum,
        # untimed, before the timed region, over the same transport and through the same function the loop times
        "sharded_equals_sequential": bool(verdict and verdict["equal"]),
        "sharded_check": verdict,
        ("rccl_world_size" if transport == "rccl" else ("p2p_world_size" if p2p is not None else "host_staged_world_size")):
            dist.get_world_size(),
        "cpu_baseline": None,   # timed on rank 0 of the N = 1 run only (bench contract); see that line
        "kernels_timed_region_rank0": kern,
        "roofline": B.score_roofline(kern, args.dtype, L, T, (c1 - c0) * args.layers * steps),
        # [max, min] over the ranks, mean ms per step: PhaseTimer phases on the launch stream (dpselect: local distances +
        # all-gather + select; blocks: the rank's chunks x layers incl. starting the per-chunk gathers; offsets: the id scan;
        # rotate: kept keys at their final ids; assembly: waiting for / gathering the other ranks' rows; finalize = the
        # last three), the host's wall time per step and the idle time at the closing barrier
        "phase_ms": phase_ms,
        "phase_bytes_rank0": dict(phases.bytes),
    }


def bench_main(args, rank: int, world: int, local_rank: int):
    import bench as B

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")   # world size 1 without a launcher (RETAKE_FORCE_SHARDED=1)
    os.environ.setdefault("MASTER_PORT", "29544")
    transport = getattr(args, "transport", "rccl")
    # RETAKE_BENCH_SHARE_GPU=1 (tests only): every rank runs on GPU 0 over a gloo control plane, so that the multi-rank bench
    # path can run on a 1-GPU box (RCCL refuses two ranks on one device) - with the p2p transport as the data plane, or
    # (`--transport host`) with the collective-transport code path (ChunkGather, all_gather_caches) staged through the host
    share = os.environ.get("RETAKE_BENCH_SHARE_GPU") == "1" and transport in ("p2p", "host")
    if transport == "host" and not share:
        raise SystemExit("--transport host is the tests' form of the collective path (RETAKE_BENCH_SHARE_GPU=1); use rccl or p2p")
    dev = torch.device("cuda", 0 if share else local_rank)
    torch.cuda.set_device(dev)
    if not dist.is_initialized():
        if share:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    p2p = enable_p2p(device=dev) if transport == "p2p" else None
    out = measure_sharded(args, rank, world, dev, transport, share, args.steps, args.warmup,
                          verify=not getattr(args, "no_self_check", False))
    dist.barrier()
    if p2p is not None:
        disable_p2p()
    dist.destroy_process_group()
    if rank == 0:
        # RCCL writes its version banner to stdout through C stdio, which a pipe only sees at exit - after anything
        # Python printed.  Flush it out first so that the JSON line is the last line of rank 0's stdout.
        try:
            C.CDLL(None).fflush(None)
        except OSError:
            pass
        sys.stdout.flush()
        B.emit(out, getattr(args, "report", None))
