#!/usr/bin/env python3
"""Does releasing a tensor WITHOUT torch's caching allocator (PYTORCH_NO_CUDA_MEMORY_CACHING=1: `del` = hipFree) wait for
the kernels that are still queued to read it?  Queue ~25 ms of copies out of a 1 GiB tensor, drop the tensor, time the drop.
If the drop returns long before the queue drains, the memory went back to the driver while kernels were reading it - with
several processes on one GPU another process can be handed those pages (profiles/r15_p2p_hunt.log)."""
import os
import sys
import time

os.environ["PYTORCH_NO_CUDA_MEMORY_CACHING"] = sys.argv[1] if len(sys.argv) > 1 else "1"
import torch

dev = torch.device("cuda", 0)
for label, stream in (("default stream", None), ("side stream", torch.cuda.Stream(dev))):
    with torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream(dev)):
        x = torch.randn(1 << 28, device=dev)
        y = torch.empty_like(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(60):
            y.copy_(x)
        t1 = time.perf_counter()
        del x
        t2 = time.perf_counter()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        del y
    print(f"no_caching={os.environ['PYTORCH_NO_CUDA_MEMORY_CACHING']} {label}: enqueue {1e3 * (t1 - t0):.1f} ms, drop of the source "
          f"{1e3 * (t2 - t1):.1f} ms, queue drained {1e3 * (t3 - t2):.1f} ms later", flush=True)
