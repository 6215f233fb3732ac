#!/usr/bin/env python3
"""Do kernels of a process READ WRONG DATA from its own intact, resident tensors when several processes share one GPU and
release / allocate device memory at every torch op (PYTORCH_NO_CUDA_MEMORY_CACHING=1)?  (profiles/r15_p2p_hunt.log)

    [PYTORCH_NO_CUDA_MEMORY_CACHING=1] python -m torch.distributed.run --nproc-per-node N tools/read_glitch_probe.py
        [--seconds 60] [--gb 4] [--map]

Every rank (all on GPU 0, gloo control plane only) holds `--gb` of resident bf16 tensors, fingerprints them once (sum of the
bit patterns), then for `--seconds`: allocates and frees temporaries the way a torch op sequence does and fingerprints the
resident tensors again.  A fingerprint that differs is checked element by element against a regeneration from the seed:
"transient read" = the tensor is intact, the read was wrong.  NO peer mapping exists unless --map (which opens the p2p
transport's landing buffers between the ranks and never touches them)."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-retake_amd"))
import torch
import torch.distributed as dist


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60)
    ap.add_argument("--gb", type=float, default=4.0)
    ap.add_argument("--map", action="store_true")
    ap.add_argument("--push", type=float, default=0.0, metavar="MB",
                    help="with --map: every look also all-gathers a payload of this size through the mapped buffers (peers WRITE "
                         "into each other's landing buffers) and checks what arrived")
    a = ap.parse_args()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    keep = None
    if a.map:
        from retake.p2p import P2PGroup

        keep = P2PGroup(device=dev)
        keep.symmetric(64 << 20)
    n_el = 28 * 6272 * 128                       # one q projection of the bench (45 MB in bf16)
    n_t = max(2, int(a.gb * 1e9 / (2 * n_el)))

    def gen(i):
        g = torch.Generator(device=dev).manual_seed(77000 + 1000 * rank + i)
        x = torch.randn(n_el, generator=g, device=dev, dtype=torch.float32)
        y = (1.7 * x).to(torch.bfloat16)
        torch.cuda.synchronize()
        return y

    pool = [gen(i) for i in range(n_t)]

    def sums():
        return torch.stack([t.view(torch.int16).sum(dtype=torch.int64) for t in pool]).cpu()

    ref = sums()
    again = sums()
    assert torch.equal(ref, again), "fingerprints not reproducible before anybody competes"
    dist.barrier()
    t_end = time.perf_counter() + a.seconds
    looks = reads = transient = persistent = landed_wrong = 0
    payload = want = None
    if a.map and a.push > 0:
        n_pay = int(a.push * 1e6 / 2) // 8 * 8

        def pay(r):
            g = torch.Generator(device=dev).manual_seed(555 + r)
            return torch.randn(n_pay, generator=g, device=dev, dtype=torch.float32).to(torch.bfloat16)

        payload = pay(rank)
        want = torch.stack([pay(r).view(torch.int16).sum(dtype=torch.int64) for r in range(world)]).cpu()
        torch.cuda.synchronize()
    stop = torch.zeros(1, dtype=torch.int64)
    while True:
        # (the pushes are collective: every rank leaves the loop at the same look)
        stop[0] = int(time.perf_counter() >= t_end)
        dist.all_reduce(stop, op=dist.ReduceOp.MAX)
        if int(stop[0]):
            break
        if payload is not None:
            got = keep.all_gather(payload)
            s = torch.stack([got[r].view(torch.int16).sum(dtype=torch.int64) for r in range(world)]).cpu()
            for r in (s != want).nonzero().flatten().tolist():
                landed_wrong += 1
                print(f"rank {rank}: look {looks + 1}: the block of rank {r} read from the landing buffer has fingerprint {int(s[r])} != {int(want[r])}",
                      flush=True)
        for i in range(0, n_t, 7):               # the churn of an op sequence: temporaries come and go
            tmp = pool[i].float()
            tmp2 = tmp * 2.0
            del tmp, tmp2
        now = sums()
        looks += 1
        reads += n_t
        for i in (now != ref).nonzero().flatten().tolist():
            torch.cuda.synchronize()
            same = torch.equal(pool[i], gen(i))
            transient += int(same)
            persistent += int(not same)
            print(f"rank {rank}: look {looks}: tensor {i} fingerprint {int(now[i])} != {int(ref[i])}; tensor equals its regeneration: {same}",
                  flush=True)
    res = torch.tensor([reads, transient, persistent, landed_wrong], dtype=torch.int64)
    dist.all_reduce(res)
    dist.barrier()
    if rank == 0:
        mode = "no caching allocator" if os.environ.get("PYTORCH_NO_CUDA_MEMORY_CACHING") == "1" else "caching allocator"
        print(f"READ_GLITCH {world} process(es), {mode}, peer mappings {('WRITTEN (%.0f MB all-gather per look; %d wrong blocks read from landing buffers)' % (a.push, int(res[3]))) if payload is not None else ('OPEN (idle)' if a.map else 'none')}, {n_t} x {2 * n_el / 1e6:.0f} MB "
              f"resident per process, {a.seconds:.0f} s: {int(res[0])} tensor reads fingerprinted, {int(res[1])} TRANSIENT wrong reads "
              f"(tensor intact), {int(res[2])} tensors really changed", flush=True)
    if keep is not None:
        keep.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
