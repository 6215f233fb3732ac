"""tests/proc_util.run_job: a multi-process job that hangs is killed - launcher AND ranks - within the time limit (CPU)."""
import os
import sys
import time

import psutil
import pytest

import proc_util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cmd(script, world):
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.join(ROOT, "tests", script)]


def test_a_hung_job_is_killed_with_all_its_ranks_and_reported():
    t0 = time.perf_counter()
    with pytest.raises(proc_util.JobTimeout) as ei:
        proc_util.run_job(_cmd("mp_hang_cpu.py", 2), timeout=25, cwd=ROOT)
    took = time.perf_counter() - t0
    assert took < 25 + 45, took                       # the kill and the final read of the pipes are bounded too
    e = ei.value
    assert len(e.killed) >= 2                          # the ranks live in their own sessions: they were found and killed
    assert all(not psutil.pid_exists(pid) or psutil.Process(pid).status() == psutil.STATUS_ZOMBIE for pid in e.killed)
    assert "rank 0 up" in e.stdout and "rank 1 up" in e.stdout   # the output up to the hang is in the report


def test_a_finished_job_is_returned_as_is():
    r = proc_util.run_job([sys.executable, "-c", "print('hello')"], timeout=60)
    assert r.returncode == 0 and r.stdout.strip() == "hello"
