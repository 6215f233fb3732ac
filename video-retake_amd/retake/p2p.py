"""Direct peer-to-peer all-gather over xGMI for the ranks of one node (`rtk_p2p_*`, include/retake_hip.h).

xGMI is point to point: every GPU has its own link to every other GPU of the node.  A ring all-gather walks
world - 1 serial steps; the exchanges of this path are small (distance rows 1.6 MB, id offsets a few hundred bytes, one
chunk's kept rows 11 MB per rank at 8 ranks), so they are bound by those steps, not by the links.  Here every rank maps
its peers' landing buffers once (hipIpc handles, exchanged through the process group's control plane) and then PUSHES
its block into all of them with one kernel (+ a one-wave launch behind it for the arrival flags): one hop, all links at once, no intermediate copies - the landing buffer can
be the final layout (`SymmetricBuffer.push` takes strided segments).

    p2p = P2PGroup(group)                 # once per process group
    out = p2p.all_gather(x)               # [world, *x.shape], valid until this rank's NEXT all_gather (clone to keep)
    buf = p2p.symmetric(nbytes)           # same-sized landing buffer on every rank
    buf.push(src, seg_bytes, nseg, src_stride, dst_offset, dst_stride); ...; buf.wait()

No collective library is involved; `torch.distributed` is only used to hand the 64-byte handles around
(`all_gather_object`), so any backend works for that - the 2-process tests run it over gloo with both ranks on one GPU.

Environment: the host driver of this pool only supports dmabuf IPC, so `HSA_ENABLE_IPC_MODE_LEGACY=0` must be in the
environment BEFORE the HIP runtime initialises (it is exported on the build and GPU boxes; bench.py and the tests set it
as a default too) - without it `hipIpcGetMemHandle` fails with "invalid argument".  This module sets the default at
import, which is early enough only if nothing has touched the GPU yet.

Status: opt-in.  The protocol has run with 2, 3 AND 8 rank processes sharing ONE GPU (tests/mp_p2p_gpu.py,
tests/mp_sharded_gpu.py, tests/test_00_world8_gpu.py); what has never been exercised is a LINK: the landing buffers are
ordinary coarse-grained `hipMalloc` memory written by remote GPUs, and a consumer kernel launched after the wait kernel
sees that data through the kernel-boundary cache invalidate, which runs on one GPU cannot show - run those scripts on a
node with >= 2 GPUs before relying on it.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional

import torch
import torch.distributed as dist

import os

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # see the module docstring

from . import _native as nv  # noqa: E402

_TIMEOUT_MS = 20000


_GUARD = (2 << 20) if os.environ.get("RETAKE_P2P_GUARD") == "1" else 0    # debugging aid: pattern-filled guard regions
_NO_REMOTE = os.environ.get("RETAKE_P2P_NO_REMOTE_WRITES") == "1"          # debugging aid: mappings opened, never written


class _DeviceMem:
    """Device memory owned by the library (rtk_p2p_alloc: whole 2 MiB granules, never a fragment of a block shared with
    other allocations), viewable as a torch tensor without a copy.  RETAKE_P2P_GUARD=1 (debugging aid) puts 2 MiB of a
    fixed byte pattern either side of it inside the same allocation - what peers map is then guard | buffer | guard - and
    `guards_intact()` says whether anything wrote there."""

    PATTERN = 0xC3

    def __init__(self, nbytes: int, uncached: bool = False):
        p = C.c_void_p()
        nv.check(nv.lib.rtk_p2p_alloc(nbytes + 2 * _GUARD, int(uncached), C.byref(p)), "rtk_p2p_alloc")
        self.base, self.guard = int(p.value), _GUARD
        self.ptr, self.nbytes = self.base + _GUARD, int(nbytes)
        self._guards = None

    def fill_guards(self, device):
        if self.guard:
            raw = _RawView(self.base, self.nbytes + 2 * self.guard).tensor(device)
            raw[: self.guard].fill_(self.PATTERN)
            raw[self.guard + self.nbytes:].fill_(self.PATTERN)
            torch.cuda.synchronize(device)
            self._guards = (raw[: self.guard], raw[self.guard + self.nbytes:])

    def guards_intact(self) -> bool:
        return self._guards is None or all(bool((g == self.PATTERN).all().item()) for g in self._guards)

    @property
    def __cuda_array_interface__(self):
        return {"shape": (self.nbytes,), "typestr": "|u1", "data": (self.ptr, False), "version": 3, "strides": None}

    def tensor(self, device) -> torch.Tensor:
        t = torch.as_tensor(self, device=device)
        if t.data_ptr() != self.ptr:
            raise RuntimeError("torch copied the p2p buffer instead of viewing it")
        return t

    def free(self):
        if self.ptr:
            self._guards = None
            nv.lib.rtk_p2p_free(C.c_void_p(self.base))
            self.ptr = self.base = 0


class _RawView:
    def __init__(self, ptr, nbytes):
        self.ptr, self.nbytes = ptr, nbytes

    @property
    def __cuda_array_interface__(self):
        return {"shape": (self.nbytes,), "typestr": "|u1", "data": (self.ptr, False), "version": 3, "strides": None}

    def tensor(self, device):
        return torch.as_tensor(self, device=device)


class SymmetricBuffer:
    """One landing buffer of `nbytes` on every rank, mapped into every rank.  `push` writes this rank's segments into
    all of them; `wait` returns (on the current stream) once every rank's pushes up to the same count have landed here.
    Every rank must call push / wait equally often.  A sender can be one exchange ahead of a receiver, never two, so a
    caller that reuses regions alternates two of them; what landed in a region is then guaranteed intact only for reads
    this rank enqueued BEFORE its own next push (a peer's push after next, which rewrites the region, is ordered behind
    that push through the peer's wait - and behind nothing later)."""

    def __init__(self, owner: "P2PGroup", nbytes: int):
        self.owner = owner
        g, dev = owner, owner.device
        self.nbytes = (int(nbytes) + 15) // 16 * 16
        with torch.cuda.device(dev):
            self.mem = _DeviceMem(self.nbytes, uncached=os.environ.get("RETAKE_P2P_UNCACHED_LANDING") == "1")   # (A/B aid)
            self.flags = _DeviceMem(4 * nv.P2P_MAX_RANKS, uncached=True)
            self.local = self.mem.tensor(dev)                        # uint8 [nbytes]
            self.mem.fill_guards(dev)
            self.flags.fill_guards(dev)
            self.status = torch.zeros(1, dtype=torch.int32, device=dev)
            handles = []
            for m in (self.mem, self.flags):
                h = (C.c_char * nv.IPC_HANDLE_BYTES)()
                off = C.c_size_t()
                nv.check(nv.lib.rtk_p2p_export(C.c_void_p(m.ptr), h, C.byref(off)), "rtk_p2p_export")
                handles.append((bytes(h), int(off.value)))
            table: List[Optional[list]] = [None] * g.world
            dist.all_gather_object(table, handles, group=g.group)
            self.peers = nv.P2PPeers()
            self._opened = []
            for r, hs in enumerate(table):
                ptrs = []
                for (h, off), own in zip(hs, (self.mem, self.flags)):
                    if r == g.rank:
                        ptrs.append(own.ptr)
                        continue
                    base = C.c_void_p()
                    nv.check(nv.lib.rtk_p2p_open((C.c_char * nv.IPC_HANDLE_BYTES).from_buffer_copy(h), C.byref(base)),
                             "rtk_p2p_open")
                    self._opened.append(int(base.value))
                    ptrs.append(int(base.value) + off)
                self.peers.buf[r], self.peers.flag[r] = ptrs
            self.mapped = [(r, int(self.peers.buf[r]), self.nbytes, int(self.peers.flag[r])) for r in range(g.world)]
            if _NO_REMOTE:   # this rank's kernels only ever see its OWN buffers; the peers' data comes through the host
                self.peers_all, self.peers = self.peers, nv.P2PPeers()
                for r in range(g.world):
                    self.peers.buf[r], self.peers.flag[r] = self.peers_all.buf[g.rank], self.peers_all.flag[g.rank]
        self.epoch = 0
        dist.barrier(group=g.group)   # nobody pushes before everybody has mapped everybody

    def push(self, src: torch.Tensor, seg_bytes: int, nseg: int, src_stride: int, dst_offset: int, dst_stride: int,
             stream=None):
        """`nseg` segments of `seg_bytes` from src.data_ptr() + s * src_stride -> byte dst_offset + s * dst_stride of
        every rank's buffer (all byte counts multiples of 16), then this rank's arrival flag on every rank."""
        if dst_offset + (nseg - 1) * dst_stride + seg_bytes > self.nbytes and nseg > 0:
            raise ValueError("p2p push outside the symmetric buffer")
        g = self.owner
        self.epoch += 1
        if _NO_REMOTE:
            return self._push_through_host(src, seg_bytes, nseg, src_stride, dst_offset, dst_stride, stream)
        st = nv.stream() if stream is None else C.c_void_p(stream.cuda_stream)
        nv.check(nv.lib.rtk_p2p_push(C.c_void_p(src.data_ptr()), seg_bytes, nseg, src_stride, C.byref(self.peers),
                                     g.rank, g.world, dst_offset, dst_stride, self.epoch, st),
                 "rtk_p2p_push")

    def _push_through_host(self, src, seg_bytes, nseg, src_stride, dst_offset, dst_stride, stream):
        """RETAKE_P2P_NO_REMOTE_WRITES=1 (debugging aid): the same call sequence with the peers' buffers MAPPED but never
        written - every rank's segments travel through the host (gloo all-gather, blocking) and each rank stores all of
        them, and all arrival flags, into its OWN buffers.  If a failure that needs the mappings disappears here, the
        writes through them are implicated; if it stays, they are not."""
        g = self.owner
        ts = torch.cuda.current_stream(g.device) if stream is None else stream
        with torch.cuda.stream(ts):
            raw = _RawView(src.data_ptr(), max(16, (nseg - 1) * src_stride + seg_bytes)).tensor(g.device) if nseg and seg_bytes \
                else torch.empty(0, dtype=torch.uint8, device=g.device)
            if nseg and seg_bytes:
                segs = torch.as_strided(raw, (nseg, seg_bytes), (src_stride, 1)).contiguous()
            else:
                segs = torch.empty((0, 0), dtype=torch.uint8, device=g.device)
            meta = torch.tensor([dst_offset, dst_stride, nseg, seg_bytes], dtype=torch.int64)
            ts.synchronize()
            host = segs.cpu()
        metas = [torch.empty_like(meta) for _ in range(g.world)]
        dist.all_gather(metas, meta, group=g.group)
        blobs: List[Optional[torch.Tensor]] = [None] * g.world
        dist.all_gather_object(blobs, host, group=g.group)
        with torch.cuda.stream(ts):
            for r in range(g.world):
                off, stride, n, sb = (int(v) for v in metas[r].tolist())
                if n and sb:
                    dst = torch.as_strided(self.local, (n, sb), (stride, 1), off)
                    dst.copy_(blobs[r].to(g.device))
            fl = _RawView(self.flags.ptr, 4 * g.world).tensor(g.device).view(torch.int32)
            fl.fill_(self.epoch if self.epoch < 2 ** 31 else self.epoch - 2 ** 32)

    def wait(self, stream=None, timeout_ms: int = _TIMEOUT_MS):
        g = self.owner
        st = nv.stream() if stream is None else C.c_void_p(stream.cuda_stream)
        nv.check(nv.lib.rtk_p2p_wait(C.c_void_p(self.flags.ptr), g.world, self.epoch, timeout_ms, nv.ptr(self.status),
                                     st), "rtk_p2p_wait")

    def resync(self, lo: int, hi: int):
        """After an exchange the ranks entered with DIFFERENT push counts (lo / hi = the smallest / largest `epoch` over
        the ranks, agreed through the control plane): wait for what every sender has published, then count on from the
        largest so that the next push / wait pair matches on every rank again (published epochs only grow)."""
        g = self.owner
        nv.check(nv.lib.rtk_p2p_wait(C.c_void_p(self.flags.ptr), g.world, int(lo), _TIMEOUT_MS, nv.ptr(self.status),
                                     nv.stream()), "rtk_p2p_wait")
        self.epoch = int(hi)

    def check(self):
        """Host-side check of the bounded waits (synchronises): raises if a sender never arrived."""
        s = int(self.status.item())
        if s:
            raise RuntimeError(f"p2p wait on rank {self.owner.rank} timed out: nothing arrived from rank {s - 1}")

    def close(self):
        """Unmap and free.  Only at the very end (P2PGroup.close): re-allocating after a free can hand a peer a handle its
        runtime still associates with the freed memory (seen on ROCm 7.2 as pushes that never arrive), so buffers are
        never recycled mid-run."""
        for base in self._opened:
            nv.lib.rtk_p2p_close(C.c_void_p(base))
        self._opened = []
        self.local = None
        self.mem.free()
        self.flags.free()


class P2PGroup:
    """The ranks of `group` (one process per GPU, all on one node) with their buffers mapped into each other."""

    def __init__(self, group=None, device=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        if self.world > nv.P2P_MAX_RANKS:
            raise ValueError(f"p2p all-gather supports up to {nv.P2P_MAX_RANKS} ranks, got {self.world}")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self._scratch: Optional[SymmetricBuffer] = None
        self._slot = 0
        self._calls = 0
        self._buffers: List[SymmetricBuffer] = []
        self._last_src: Optional[torch.Tensor] = None

    def symmetric(self, nbytes: int) -> SymmetricBuffer:
        b = SymmetricBuffer(self, nbytes)
        self._buffers.append(b)
        return b

    def all_gather(self, x: torch.Tensor) -> torch.Tensor:
        """x (same shape and dtype on every rank) -> [world, *x.shape] in rank order.  The result is a view of the landing
        buffer and stays valid until this rank's NEXT call: the two halves alternate because a peer may already push
        call n+1 while this rank still reads call n, and that peer pushes call n+2 into this half as soon as it has seen
        this rank's push n+1 - reads enqueued after that push are ordered behind nothing.  Clone what must live longer."""
        nv.require_device(x)
        x = x.contiguous()
        nb = x.numel() * x.element_size()
        slot = (nb + 15) // 16 * 16
        if self._scratch is None or slot > self._slot:   # (re)size: collective, like the call itself
            # a grown payload gets a new, larger buffer; the old one stays mapped until close() (results handed out
            # earlier may still be read, and nothing is gained by unmapping mid-run)
            self._slot = max(slot, 1 << 16)
            self._scratch = self.symmetric(2 * self.world * self._slot)
            self._calls = 0
        buf = self._scratch
        half = (self._calls & 1) * self.world * self._slot
        self._calls += 1
        with torch.cuda.device(self.device):
            src = x
            if nb != slot or x.data_ptr() % 16:   # pad to the 16-byte granule of the push kernel
                src = torch.zeros(slot, dtype=torch.uint8, device=x.device)
                src[:nb] = x.reshape(-1).view(torch.uint8)
            if slot:
                buf.push(src, slot, 1, slot, half + self.rank * self._slot, self._slot)
            else:
                buf.push(src, 0, 0, 16, half, 16)
            buf.wait()
            self._last_src = src   # read by the push kernel after this call has returned: kept until the next call
        out = buf.local[half:half + self.world * self._slot].view(self.world, self._slot)[:, :nb]
        if nb == slot:
            return out.view(x.dtype).view((self.world,) + tuple(x.shape))
        return out.contiguous().view(x.dtype).view((self.world,) + tuple(x.shape))

    def check(self):
        for b in self._buffers:
            b.check()

    def guards_intact(self) -> bool:
        """RETAKE_P2P_GUARD=1: the guard regions either side of every landing / flag buffer of this rank still hold their pattern."""
        return all(b.mem.guards_intact() and b.flags.guards_intact() for b in self._buffers)

    def mapped_ranges(self):
        """[(buffer index, peer, address of the peer's landing buffer in THIS process, bytes, address of its flags)]."""
        return [(i,) + m for i, b in enumerate(self._buffers) for m in b.mapped]

    def close(self):
        torch.cuda.synchronize(self.device)
        dist.barrier(group=self.group)
        for b in self._buffers:
            b.close()
        self._buffers, self._scratch = [], None
