"""A job whose rank 1 never joins a gloo barrier (tests/test_proc_util_cpu.py: the time limit of proc_util.run_job)."""
import os
import time

import torch.distributed as dist

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
print(f"rank {rank} up", flush=True)
if rank == 1:
    time.sleep(3600)
dist.barrier()
