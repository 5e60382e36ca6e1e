"""One process per GPU: start the ranks of a multi-GPU run from a plain `python script.py --gpus N`.

The k-mesh path shards with no data-path collective (the loops being cut are
pythtb.py:2475-2497 -- mesh rows -- and :1047 -- the k list), so a "launcher" is only N copies of the same
script with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in their environment: the variables
`torch.distributed.run` sets, so a script behaves the same under either.

The parent must not have touched the GPU (no HIP call, no `torch.cuda.is_available()`): it only starts
children and waits.  Nothing is re-exec'ed.  If one rank fails, the others are terminated (by PID) and the
first non-zero status is returned, so a hung collective on the surviving ranks cannot outlive the failure.
"""
import json
import os
import socket
import struct
import subprocess
import sys
import tempfile
import time

__all__ = ["under_launcher", "free_port", "spawn_ranks", "Rendezvous"]


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


class Rendezvous(object):
    """The control plane of a one-node, one-process-per-GPU run WITHOUT torch: the ranks of `spawn_ranks` (or of
    `python -m torch.distributed.run`, whose RANK / WORLD_SIZE / MASTER_* variables are the same) meet on a TCP socket of
    rank 0 and exchange small byte strings -- RCCL's 128-byte unique id, one-word agreements, barriers, and (as the labelled
    fallback when RCCL is unavailable) the few doubles a rank reports.  Nothing of the data path goes through here: the
    path's one collective is the RCCL gather.

    Where the ranks meet: rank 0 listens on MASTER_ADDR:MASTER_PORT if that port is free (our own launcher picks a free
    one and nobody listens on it); under torch.distributed.run the agent's store owns MASTER_PORT, so rank 0 takes any free
    port and publishes it in a small file named after MASTER_ADDR / MASTER_PORT (/ TORCHELASTIC_RUN_ID) in the temp dir --
    one node, one file system.  Every connection is checked with a token, so a stale file or a foreign listener is skipped.

    A star: every operation is an all-gather of byte strings through rank 0 (ranks send, rank 0 returns the list); barrier,
    broadcast and all_min are that with small payloads.  World sizes are <= 8 and payloads tiny: latency, not bandwidth."""
    MAGIC = b"TBKRDZV1"

    def __init__(self, rank=None, world=None, addr=None, port=None, timeout=600.0):
        env = os.environ
        self.rank = int(env.get("RANK", "0")) if rank is None else int(rank)
        self.world = int(env.get("WORLD_SIZE", "1")) if world is None else int(world)
        self.addr = addr or env.get("MASTER_ADDR", "127.0.0.1")
        self.port = int(port if port is not None else env.get("MASTER_PORT", "29500"))
        self.timeout = float(timeout)
        self.name = "socket"
        self._peers, self._sock, self._file = {}, None, None
        self._seq = 0
        if self.world > 1:
            self._connect()

    # -- wiring
    def _rdzv_file(self):
        tag = "%s_%d_%s" % (self.addr.replace(":", "-"), self.port, os.environ.get("TORCHELASTIC_RUN_ID", "none"))
        return os.path.join(tempfile.gettempdir(), "tbk_rdzv_%s.json" % "".join(c if c.isalnum() or c in "._-" else "-" for c in tag))

    def _connect(self):
        deadline = time.monotonic() + self.timeout
        path = self._rdzv_file()
        if self.rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            try:
                srv.bind((self.addr, self.port))
            except OSError:                                  # torch.distributed.run's store lives there
                srv.bind((self.addr, 0))
            srv.listen(self.world)
            token = os.urandom(16).hex()
            tmp = path + ".%d" % os.getpid()
            with open(tmp, "w") as f:
                json.dump({"port": srv.getsockname()[1], "token": token, "pid": os.getpid()}, f)
            os.replace(tmp, path)                             # atomic: a reader sees the old file or the new one
            self._file, self._sock = path, srv
            while len(self._peers) < self.world - 1:
                srv.settimeout(max(0.1, deadline - time.monotonic()))
                try:
                    c, _ = srv.accept()
                except socket.timeout:
                    raise RuntimeError("rendezvous: %d of %d ranks joined within %g s" % (len(self._peers) + 1, self.world, self.timeout))
                c.settimeout(10.0)
                try:
                    hello = self._recv_exact(c, len(self.MAGIC) + 32 + 4)
                except (OSError, RuntimeError):
                    c.close()
                    continue
                r = struct.unpack("<i", hello[-4:])[0]
                if hello[:len(self.MAGIC)] != self.MAGIC or hello[len(self.MAGIC):-4].decode("ascii", "replace") != token \
                        or not 0 < r < self.world or r in self._peers:
                    c.close()                                  # not one of ours
                    continue
                c.sendall(self.MAGIC)
                c.settimeout(self.timeout)
                c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                self._peers[r] = c
        else:
            last = "no rendezvous file %s" % path
            while True:
                if time.monotonic() > deadline:
                    raise RuntimeError("rendezvous: rank %d could not reach rank 0 within %g s (%s)" % (self.rank, self.timeout, last))
                try:
                    with open(path) as f:
                        info = json.load(f)
                    c = socket.create_connection((self.addr, int(info["port"])), timeout=5.0)
                    c.sendall(self.MAGIC + info["token"].encode("ascii") + struct.pack("<i", self.rank))
                    if self._recv_exact(c, len(self.MAGIC)) != self.MAGIC:
                        raise RuntimeError("foreign listener")
                    c.settimeout(self.timeout)
                    c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    self._sock = c
                    return
                except (OSError, ValueError, KeyError, RuntimeError) as e:      # stale file, nobody listening yet, wrong token
                    last = " ".join(str(e).split())
                    time.sleep(0.05)

    @staticmethod
    def _recv_exact(c, n):
        buf = bytearray()
        while len(buf) < n:
            chunk = c.recv(n - len(buf))
            if not chunk:
                raise RuntimeError("rendezvous: peer closed the connection")
            buf += chunk
        return bytes(buf)

    def _send_frame(self, c, payload):
        c.sendall(struct.pack("<IQ", self._seq, len(payload)) + payload)

    def _recv_frame(self, c):
        seq, n = struct.unpack("<IQ", self._recv_exact(c, 12))
        if seq != self._seq:
            raise RuntimeError("rendezvous: operation %d of a peer met operation %d here (the ranks took different paths)" % (seq, self._seq))
        return self._recv_exact(c, n)

    # -- operations
    def allgather_bytes(self, payload):
        """Every rank's byte string, in rank order, on every rank."""
        payload = bytes(payload)
        self._seq = (self._seq + 1) & 0xffffffff
        if self.world == 1:
            return [payload]
        if self.rank == 0:
            parts = [payload] + [self._recv_frame(self._peers[r]) for r in range(1, self.world)]
            blob = b"".join(struct.pack("<Q", len(p)) + p for p in parts)
            for r in range(1, self.world):
                self._send_frame(self._peers[r], blob)
            return parts
        self._send_frame(self._sock, payload)
        blob = self._recv_frame(self._sock)
        parts, pos = [], 0
        for _ in range(self.world):
            n = struct.unpack_from("<Q", blob, pos)[0]
            parts.append(blob[pos + 8:pos + 8 + n])
            pos += 8 + n
        return parts

    def barrier(self):
        self.allgather_bytes(b"")

    def broadcast_bytes(self, payload, src=0):
        """`payload` of rank `src` on every rank (None elsewhere is fine)."""
        return self.allgather_bytes(payload if self.rank == src and payload is not None else b"")[src]

    def all_min(self, value):
        return min(struct.unpack("<q", p)[0] for p in self.allgather_bytes(struct.pack("<q", int(value))))

    def close(self):
        for c in list(self._peers.values()) + ([self._sock] if self._sock is not None else []):
            try:
                c.close()
            except OSError:
                pass
        self._peers, self._sock = {}, None
        if self._file is not None:
            try:
                os.unlink(self._file)
            except OSError:
                pass
            self._file = None
