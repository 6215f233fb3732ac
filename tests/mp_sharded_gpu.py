#!/usr/bin/env python3
"""Multi-rank GPU check of the chunk-sharded path, launched by tests/test_hip_parity.py::test_sharded_multi_rank_rccl as
    python -m torch.distributed.run --nproc-per-node W --master-addr 127.0.0.1 --master-port P tests/mp_sharded_gpu.py
Every rank compresses its block of one small synthetic video (bench.py's deterministic tensors as the pre-RoPE
projections, through the attention prologue at the block's provisional ids) with retake.sharded.sharded_video_step - RCCL
all-gathers of the distance rows, the temporal offsets and the compressed cache - and compares the assembled cache with
the cache the same rank builds sequentially on its own: ids, V and K bit patterns equal, in fp32 and in bf16.  Two shapes
per dtype: chunks divisible by the world size (per-chunk overlapped gathers) and one chunk more (ragged blocks, padded
assembly at the end).  Prints MP_SHARDED_OK on rank 0.

RETAKE_TEST_TRANSPORT=p2p runs the same check over the direct peer-to-peer pushes of retake/p2p.py with a gloo control
plane and rank r on GPU r % device_count - two ranks can then share the one GPU of a test box (RCCL refuses that) - and
repeats the even case so that the landing buffers of the per-chunk pushes are reused across videos.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "video-retake_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import torch.distributed as dist


def dpselect_blocks(rank, world, dev):
    """DPSelect sharded by frame block over the live transport: every rank holds 12 frames (+ the halo frame in front of its
    block), the distance rows are all-gathered, the selection runs redundantly, and at ratio < 1 the kept frames are
    exchanged - output and key-patch mask must equal the unsharded call on the whole video, sync and per-patch, fp32 and
    bf16, ratio 1 and ~1/3."""
    import numpy as np

    import retake.visual_compression as vc
    import synth
    from retake import sharded

    per, N, C = 12, 6, 256
    T = per * world
    for dtype in (torch.float32, torch.bfloat16):
        x = torch.from_numpy(synth.frames_video(500 + world, T, N, C)).to(dev).to(dtype)     # [1, T, N, C], same on every rank
        f0 = rank * per
        local = x[:, max(f0 - 1, 0):f0 + per].contiguous()
        for sync in (True, False):
            for t in (T, T // 3 + 1):
                ref_out, ref_mask = vc.memory_bank_compress_keyframe(x, t, 3, sync=sync)
                out, mask, idx, dis = sharded.dpselect_sharded(local, rank > 0, t, 3, sync=sync)
                assert torch.equal(mask, ref_mask), (str(dtype), sync, t, "mask")
                want = ref_out if t < T else x[:, f0:f0 + per]
                assert out.shape == want.shape and torch.equal(out, want), (str(dtype), sync, t, "frames")
    torch.cuda.synchronize(dev)
    dist.barrier()
    if rank == 0:
        print(f"sharded DPSelect over {world} ranks (halo frames, distance rows gathered, frame exchange at ratio < 1) == unsharded",
              flush=True)


def main():
    import bench as B
    from retake import sharded

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    if os.environ.get("RETAKE_TEST_POISON") == "1":
        # debugging aid: every torch.empty / empty_like of the process starts as NaN (floats) / a sentinel (ints), so a read
        # of memory nobody wrote shows up in the comparison instead of depending on what the allocator handed out
        real_empty, real_like = torch.empty, torch.empty_like

        def poison(t):
            if t.is_cuda and t.numel():
                if t.dtype.is_floating_point:
                    t.fill_(float("nan"))
                elif t.dtype in (torch.int64, torch.int32):
                    t.fill_(-77)
                elif t.dtype == torch.uint8:
                    t.fill_(0xA5)
            return t

        torch.empty = lambda *a, **k: poison(real_empty(*a, **k))
        torch.empty_like = lambda *a, **k: poison(real_like(*a, **k))
    p2p = os.environ.get("RETAKE_TEST_TRANSPORT") == "p2p"
    dev = torch.device("cuda", 0 if os.environ.get("RETAKE_TEST_ONE_GPU") == "1"
                       else int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    if p2p:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        sharded.enable_p2p(device=dev)
    elif os.environ.get("RETAKE_TEST_TRANSPORT") == "host":
        # debugging aid: no device-side transport at all (no RCCL, no peer mapping) - every exchange staged through the host
        dist.init_process_group("gloo", rank=rank, world_size=world)
    else:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    # the comparison itself lives in the library (bench.py --gpus N runs it before its timed region as well)
    counts = (2 * world, 2 * world + 1) + ((2 * world, 2 * world, 2 * world) if p2p else ())
    if os.environ.get("RETAKE_TEST_ONLY_MORE_CASES") == "1":
        counts = ()
    state = {}
    res = sharded.verify_sharded_equals_sequential(rank, world, dev, B.Rotary(dev), layers=2, chunk_counts=counts,
                                                   state=state, log=lambda m: print(m, flush=True))
    assert res["equal"] and len(res["cases"]) == 2 * len(counts) and (not counts or {c["dtype"] for c in res["cases"]} == {"fp32", "bf16"})
    # RETAKE_TEST_MORE_CASES="bf16:64,65": further chunk counts per dtype (world size 8: BASELINE's 64-chunk video in blocks
    # of 8 chunks, and the ragged 65-chunk split)
    for spec in filter(None, os.environ.get("RETAKE_TEST_MORE_CASES", "").split(";")):
        dname, cc = spec.split(":")
        cc = tuple(int(x) for x in cc.split(","))
        for one in cc:
            res = sharded.verify_sharded_equals_sequential(rank, world, dev, B.Rotary(dev), layers=2, chunk_counts=(one,), state=state,
                                                           log=lambda m: print(m, flush=True), dtypes=(dname,))
            assert res["equal"] and len(res["cases"]) == 1
    if os.environ.get("RETAKE_TEST_DPSELECT") == "1":
        dpselect_blocks(rank, world, dev)
    if p2p:
        sharded.disable_p2p()
    dist.destroy_process_group()
    if rank == 0:
        print("MP_SHARDED_OK", flush=True)


if __name__ == "__main__":
    main()
