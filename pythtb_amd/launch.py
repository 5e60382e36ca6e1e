"""One process per GPU: start the ranks of a multi-GPU run from a plain `python script.py --gpus N`.

The k-mesh path shards with no data-path collective (the loops being cut are
pythtb.py:2475-2497 -- mesh rows -- and :1047 -- the k list), so a "launcher" is only N copies of the same
script with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in their environment: the variables
`torch.distributed.run` sets, so a script behaves the same under either.

The parent must not have touched the GPU (no HIP call, no `torch.cuda.is_available()`): it only starts
children and waits.  Nothing is re-exec'ed.  If one rank fails, the others are terminated (by PID) and the
first non-zero status is returned, so a hung collective on the surviving ranks cannot outlive the failure.
"""
import os
import socket
import subprocess
import sys
import time

__all__ = ["under_launcher", "free_port", "spawn_ranks"]


def under_launcher():
    """True inside a rank started by spawn_ranks or torch.distributed.run."""
    return "WORLD_SIZE" in os.environ and "RANK" in os.environ


def free_port():
    s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(script, argv, nproc, env=None, timeout=None, poll=0.05):
    """Run `python script argv...` as `nproc` ranks on this node and wait for them.

    Rank r gets RANK = LOCAL_RANK = r, WORLD_SIZE = LOCAL_WORLD_SIZE = nproc, MASTER_ADDR = 127.0.0.1 and a free
    MASTER_PORT.  The children inherit stdout / stderr (rank 0's single JSON line is the parent's output).
    Returns the exit status: 0 if every rank returned 0, else the first non-zero one seen (124 on timeout)."""
    nproc = int(nproc)
    if nproc < 1:
        raise ValueError("nproc must be >= 1")
    base = dict(os.environ if env is None else env)
    base.setdefault("MASTER_ADDR", "127.0.0.1")
    base["MASTER_PORT"] = str(free_port())
    base["WORLD_SIZE"] = base["LOCAL_WORLD_SIZE"] = str(nproc)
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs across processes here
    procs = []
    for r in range(nproc):
        e = dict(base)
        e["RANK"] = e["LOCAL_RANK"] = e["GROUP_RANK"] = str(r)
        procs.append(subprocess.Popen([sys.executable, script] + list(argv), env=e))
    status = 0
    t_end = None if timeout is None else time.monotonic() + float(timeout)
    live = list(procs)
    try:
        while live:
            for p in list(live):
                rc = p.poll()
                if rc is None:
                    continue
                live.remove(p)
                if rc != 0 and status == 0:
                    status = rc if rc > 0 else 128 - rc
            if status != 0 or (t_end is not None and time.monotonic() > t_end):
                if status == 0:
                    status = 124
                break
            if live:
                time.sleep(poll)
    finally:
        # the ranks that are still running lost a peer: end them, by PID
        for p in live:
            if p.poll() is None:
                p.terminate()
        t_kill = time.monotonic() + 5.0
        for p in live:
            try:
                p.wait(timeout=max(0.1, t_kill - time.monotonic()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    return status
