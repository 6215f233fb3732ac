"""Running a multi-process job (torchrun + ranks) from a test with a hard time limit.

`torch.distributed.run` starts every rank in its OWN session (subprocess_handler: start_new_session=True), so killing the
launcher's process group does not reach the ranks - and orphaned ranks keep the launcher's stdout / stderr pipes open, so a
`communicate()` after the kill never returns (seen: a test whose job hung waited on the dead launcher's pipes until the
box's time limit).  `run_job` therefore collects the launcher's descendants BEFORE it kills anything, kills those exact
pids, and bounds the final read of the pipes as well.
"""
import os
import signal
import subprocess


class JobTimeout(Exception):
    def __init__(self, timeout, stdout, stderr, killed):
        super().__init__(f"job did not finish in {timeout} s")
        self.timeout, self.stdout, self.stderr, self.killed = timeout, stdout, stderr, killed


def _descendants(pid):
    import psutil

    try:
        return psutil.Process(pid).children(recursive=True)
    except psutil.Error:
        return []


def run_job(cmd, env=None, timeout=300, cwd=None):
    """-> CompletedProcess, or raises JobTimeout (after the launcher AND every descendant have been killed)."""
    import psutil

    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=cwd,
                         env={**os.environ, **(env or {})}, start_new_session=True)
    try:
        out, err = p.communicate(timeout=timeout)
        return subprocess.CompletedProcess(cmd, p.returncode, out, err)
    except subprocess.TimeoutExpired:
        pass
    procs = _descendants(p.pid)          # before the launcher dies: afterwards its children belong to init
    for q in procs + _descendants(p.pid):
        try:
            q.send_signal(signal.SIGKILL)
        except psutil.Error:
            pass
    try:
        os.killpg(p.pid, signal.SIGKILL)
    except ProcessLookupError:
        pass
    psutil.wait_procs(procs, timeout=20)
    try:
        out, err = p.communicate(timeout=20)
    except subprocess.TimeoutExpired:    # something still holds the pipes: give up on the output, not on the test run
        for f in (p.stdout, p.stderr):
            try:
                f.close()
            except OSError:
                pass
        out, err = "", "(output unavailable: the pipes were still held after the kill)"
    raise JobTimeout(timeout, out or "", err or "", [q.pid for q in procs])
