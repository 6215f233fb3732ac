"""WORLD SIZE 8 on ONE GPU (BASELINE configs[3]'s rank count; what the driver's 8-GPU run launches first).

Eight rank processes share GPU 0 over the p2p transport (mapped peer buffers, gloo control plane; RCCL refuses two ranks
on one device).  This file sorts FIRST on purpose and its tests must run before the pytest process itself has created a
HIP context: the KFD hardware scheduler runs at most 8 processes with compute queues concurrently (hws_max_conc_proc);
with a ninth - a pytest parent that has already run GPU tests - the runlist is oversubscribed and time-sliced, and the same
tests take 18 + 7 MINUTES instead of 14 + 8 seconds (measured, profiles/r14_parity_stats.txt).  Run out of order they
skip with that explanation instead.
"""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fresh_parent():
    if torch.cuda.is_initialized():
        pytest.skip("this process already holds a HIP context: a ninth GPU process oversubscribes the hardware scheduler "
                    "(8 concurrent processes) - run tests/test_00_world8_gpu.py first / on its own")


def _run(cmd, env=None, timeout=300):
    """Run `cmd` (launcher + ranks); normal run times here are 4-15 s.  If it is not done after `timeout` seconds the launcher
    and every rank are killed (tests/proc_util.py: torchrun's ranks live in their own sessions) and the test FAILS with the
    ranks' last output: this file is the dress rehearsal of the first 8-GPU run, and a deadlock of the eight-rank path must
    be red, not a skip `pytest -x` sails past.  (The one benign cause of slowness - a ninth GPU process oversubscribing the
    hardware scheduler - is this pytest process itself, and `_fresh_parent` has skipped before we get here if it holds a
    HIP context.)"""
    import proc_util

    try:
        return proc_util.run_job(cmd, env, timeout, cwd=ROOT)
    except proc_util.JobTimeout as e:
        pytest.fail(f"eight ranks on one GPU did not finish in {timeout} s (normally < 20 s) - a hang of the eight-rank path "
                    f"(or a foreign GPU process oversubscribing the hardware scheduler); killed pids {e.killed}.  "
                    f"stdout tail: {e.stdout[-1500:]!r}  stderr tail: {e.stderr[-3000:]!r}")


def _launch_ranks(script, world, env=None, timeout=300):
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", script)]
    return _run(cmd, env, timeout)


def test_p2p_allgather_8_processes():
    """retake/p2p.py over the C ABI (rtk_p2p_*) at eight ranks: all_gather of odd-sized / empty / growing payloads over 40
    epochs, strided pushes into a final layout, push-count resync, the bounded wait (tests/mp_p2p_gpu.py)."""
    _fresh_parent()
    r = _launch_ranks("mp_p2p_gpu.py", 8, env={"RETAKE_TEST_ONE_GPU": "1"})
    assert r.returncode == 0 and "MP_P2P_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


def test_sharded_prefill_8_ranks_over_p2p():
    """The chunk-sharded path at world size 8 (tests/mp_sharded_gpu.py): 16- and 17-chunk videos in fp32 and bf16 (even
    blocks with per-chunk pushes, landing buffers reused over four videos; ragged blocks, padded assembly), then the real
    split - BASELINE's 64-chunk video in blocks of 8 chunks and the ragged 65-chunk one, bf16 - and DPSelect sharded over 8
    frame blocks with halo frames and the frame exchange at ratio < 1.  Assembled cache == sequential cache on every rank,
    bit for bit."""
    _fresh_parent()
    env = {"RETAKE_TEST_TRANSPORT": "p2p", "RETAKE_TEST_ONE_GPU": "1", "RETAKE_TEST_MORE_CASES": "bf16:64,65",
           "RETAKE_TEST_DPSELECT": "1"}
    r = _launch_ranks("mp_sharded_gpu.py", 8, env=env)
    assert r.returncode == 0 and "MP_SHARDED_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])
    assert "sharded DPSelect over 8 ranks" in r.stdout and "chunks 65 on 8 rank(s)" in r.stdout


def test_sharded_prefill_8_ranks_collective_code_path_host_staged():
    """World size 8 through the COLLECTIVE-transport code path (`ChunkGather`, `all_gather_caches`, `all_gather_ids` - what the
    first RCCL run on eight GPUs executes), every exchange staged through the host over gloo, all ranks on GPU 0: the 16- /
    17-chunk videos in fp32 and bf16 and the 64- / 65-chunk ones in bf16, assembled == sequential bit for bit on every rank."""
    _fresh_parent()
    r = _launch_ranks("mp_sharded_gpu.py", 8, env={"RETAKE_TEST_TRANSPORT": "host", "RETAKE_TEST_ONE_GPU": "1",
                                                   "RETAKE_TEST_MORE_CASES": "bf16:64,65"})
    assert r.returncode == 0 and "MP_SHARDED_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])
    assert "chunks 64 on 8 rank(s)" in r.stdout and "chunks 65 on 8 rank(s)" in r.stdout


def test_bench_eight_ranks_share_one_gpu_p2p(tmp_path):
    """`bench.py --gpus 8 --transport p2p` end to end with all eight ranks on GPU 0 (RETAKE_BENCH_SHARE_GPU=1) on a 512-frame /
    2-layer video (16 chunks: blocks of 2): world size 8 through rank start-up, halo frames, the in-process
    self-verification in fp32 and bf16 (16- and 17-chunk videos) and the timed loop."""
    _fresh_parent()
    common = ["--frames", "512", "--layers", "2", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"]
    rep = os.path.join(str(tmp_path), "report.json")
    r = _run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--transport", "p2p", "--report", rep] + common,
             env={"RETAKE_BENCH_SHARE_GPU": "1"})
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    last = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")][-1]
    assert len(last) < 4096, len(last)     # the short contract line; the report file holds the rest
    b, full = json.loads(last), json.load(open(rep))
    assert b["n_gpus"] == 8 and b["scaling"] == "strong" and b["config"]["transport"] == "p2p" and b["value"] > 0
    assert b["sharded_equals_sequential"] is True and b["p2p_world_size"] == 8
    assert [(c["dtype"], c["chunks"]) for c in full["sharded_check"]["cases"]] == [("fp32", 16), ("fp32", 17), ("bf16", 16), ("bf16", 17)]
    assert b["config"]["assembled_cache_tokens"] == 16 * 1568 and full["cache_checksum"]["tokens_per_layer"] == 16 * 1568
    # per-phase timing, [max, min] over the eight ranks, in the line itself
    for ph in ("dpselect", "blocks", "finalize", "step"):
        assert b["phase_ms"][ph][0] >= b["phase_ms"][ph][1] >= 0.0, b["phase_ms"]


def test_in_launch_id_shift_with_eight_competing_processes():
    """RTK_UPDATE_SHIFT_NEXT under contention (tests/mp_shift_contention_gpu.py): eight processes on GPU 0 run the update
    route with the in-launch shift at the same time - the regime in which a WALL-CLOCK bound on the watcher's wait could
    trip (a process that is switched out keeps ageing); the wait is bounded by polls now.  On every process: no run-out
    latched, arrival counters back at zero, ids and caches bitwise equal to the shift-launch-per-layer route."""
    _fresh_parent()
    r = _launch_ranks("mp_shift_contention_gpu.py", 8)
    assert r.returncode == 0 and r.stdout.count("SHIFT_CONTENTION_OK") == 8, (r.stdout[-2000:], r.stderr[-3000:])


def test_a_hung_rank_turns_the_rehearsal_red():
    """The policy of `_run`: a job that does not finish FAILS (group killed, output tails in the message) - shown on a
    deliberately deadlocked rank (tests/mp_p2p_gpu.py's RETAKE_TEST_HANG_RANK hook: rank 5 never joins the group's set-up)."""
    _fresh_parent()
    with pytest.raises(pytest.fail.Exception, match="did not finish in 40 s"):
        _launch_ranks("mp_p2p_gpu.py", 8, env={"RETAKE_TEST_ONE_GPU": "1", "RETAKE_TEST_HANG_RANK": "5"}, timeout=40)
