"""Multi-process check of the direct p2p all-gather (retake/p2p.py, rtk_p2p_*), launched by
tests/test_hip_parity.py::test_p2p_allgather_processes as

    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 tests/mp_p2p_gpu.py

The control plane is gloo; rank r uses GPU r % device_count, so on a 1-GPU box both ranks map each other's buffers
on the same device (same protocol, same kernels; the stores just do not cross a link)."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-retake_amd"))


def main():
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dev = torch.device("cuda", 0 if os.environ.get("RETAKE_TEST_ONE_GPU") == "1" else rank % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from retake.p2p import P2PGroup

    if os.environ.get("RETAKE_TEST_HANG_RANK") == str(rank):
        # test hook (tests/test_00_world8_gpu.py): this rank never joins the group's set-up - a deadlocked rank.  Its peers
        # block in the control plane's collectives (handle exchange): what the caller sees is a job that does not finish.
        import time

        time.sleep(3600)
    g = P2PGroup(device=dev)

    def block(r, i, shape, dtype):
        gen = torch.Generator().manual_seed(1000 * i + r)
        x = torch.randint(-100, 100, shape, generator=gen).to(dtype)
        return x

    # 1. all_gather: shapes with odd byte counts, a growing payload (buffer regrowth), an empty one, repeated calls
    cases = [((3,), torch.int64), ((5, 7), torch.float32), ((1,), torch.bfloat16), ((0, 4), torch.float32),
             ((128, 196), torch.float32), ((4, 1568, 128), torch.bfloat16), ((33,), torch.uint8), ((3,), torch.int64)]
    held = []
    for i, (shape, dtype) in enumerate(cases):
        out = g.all_gather(block(rank, i, shape, dtype).to(dev))
        want = torch.stack([block(r, i, shape, dtype) for r in range(world)])
        assert out.shape == want.shape and out.dtype == want.dtype, (out.shape, want.shape)
        assert torch.equal(out.cpu(), want), f"all_gather case {i} differs"
        # contract (retake/p2p.py): a result is a view of the landing buffer, valid until THIS rank's next all_gather -
        # a peer may rewrite its half right after seeing this rank's next push, so it is consumed (copied to the host
        # above) before the next call and never read again; whoever needs it longer clones it
        held.append((out.clone(), want, i))
    for kept, want, i in held:
        assert torch.equal(kept.cpu(), want), f"clone of all_gather case {i} changed"
    if os.environ.get("RETAKE_TEST_BIG") == "1":
        # the sharded prefill's ragged assembly at 8 ranks: tiny exchanges (counts, flags, offsets), then a payload that
        # makes the landing buffer grow to ~1 GB, then a small one right behind it (the ids) - all on alternating halves
        seq = [((1,), torch.int64), ((2,), torch.int64), ((2,), torch.int64), ((1,), torch.int64),
               ((2, 2, 4, 14112, 128), torch.bfloat16), ((2, 3, 14112), torch.int64), ((1,), torch.int64),
               ((2, 2, 4, 14112, 128), torch.bfloat16), ((2, 3, 14112), torch.int64)]
        for j, (shape, dtype) in enumerate(seq):
            i = 1000 + j
            mine = block(rank, i, shape, dtype).to(dev)
            out = g.all_gather(mine)
            got = torch.cat([out[r].reshape(-1)[: 1 << 22] for r in range(world)]).clone()   # a copy kernel reads the landing buffer
            full = out.clone()
            del mine
            for r in range(world):
                want = block(r, i, shape, dtype)
                if not torch.equal(full[r].cpu(), want):
                    bad = (full[r].cpu().reshape(-1) != want.reshape(-1)).nonzero().reshape(-1)
                    raise AssertionError(f"rank {rank}: big all_gather step {j} {shape}: block of rank {r} differs in {bad.numel()} "
                                         f"elements, first at {int(bad[0])}, last at {int(bad[-1])}")
            del got, full
        if rank == 0:
            print("MP_P2P_BIG_OK", flush=True)
    for i in range(8, 40):   # many epochs through the same halves
        out = g.all_gather(block(rank, i, (64, 128), torch.float32).to(dev))
        assert torch.equal(out.cpu(), torch.stack([block(r, i, (64, 128), torch.float32) for r in range(world)]))

    # 2. strided pushes into the final layout: [heads, world, rows, D] assembled from per-rank [heads, rows, D] in two
    #    row batches, one wait at the end (the cache-assembly pattern)
    H, rows, D = 4, 48, 128
    buf = g.symmetric(H * world * rows * D * 2)
    mine = block(rank, 77, (H, rows, D), torch.bfloat16).to(dev)
    row_b = D * 2
    for r0, n in ((0, 16), (16, 32)):
        part = mine[:, r0:r0 + n].contiguous()
        buf.push(part, n * row_b, H, n * row_b, (rank * rows + r0) * row_b, world * rows * row_b)
    buf.wait()
    got = buf.local.view(torch.bfloat16).view(H, world, rows, D)
    want = torch.stack([block(r, 77, (H, rows, D), torch.bfloat16) for r in range(world)], dim=1)
    assert torch.equal(got.cpu(), want), "strided push layout differs"
    g.check()

    # 2b. ranks that entered an exchange with different push counts (one rank pushed a chunk the others skipped): after
    #     the counts have been agreed through the control plane, resync waits for the smallest - no stall, no latched
    #     error - and continues from the largest, and the next push / wait pair matches again
    rs = g.symmetric(world * 64)
    probe = torch.full((16,), rank + 1, dtype=torch.int32, device=dev)
    rs.push(probe, 64, 1, 64, rank * 64, 64)
    if rank == 0:
        rs.push(probe, 64, 1, 64, rank * 64, 64)       # one push more than everybody else
    counts = [None] * world
    dist.all_gather_object(counts, rs.epoch)
    rs.resync(min(counts), max(counts))
    rs.check()
    probe2 = torch.full((16,), 100 + rank, dtype=torch.int32, device=dev)
    rs.push(probe2, 64, 1, 64, rank * 64, 64)
    rs.wait()
    got = rs.local.view(torch.int32).view(world, 16)[:, 0].cpu().tolist()
    assert got == [100 + r for r in range(world)], got
    rs.check()
    torch.cuda.synchronize()
    dist.barrier()

    # 3. a sender that never arrives is an error, not a hang: rank 0 waits 100 ms for pushes the others never make
    lost = g.symmetric(1024)
    if rank == 0:
        lost.epoch += 1           # pretend a push round happened
        lost.wait(timeout_ms=100)
        try:
            lost.check()
            raise AssertionError("the bounded wait did not report the missing sender")
        except RuntimeError as e:
            assert "rank 0" in str(e) or "rank 1" in str(e), str(e)
        lost.status.zero_()
    g.close()
    dist.barrier()
    if rank == 0:
        print("MP_P2P_OK", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
