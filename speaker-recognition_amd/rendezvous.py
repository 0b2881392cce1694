"""Host-side meeting point of the N single-GPU processes of ONE node -- barrier, gather of small Python objects, max of a
float -- without torch, MPI or any GPU collective.

The data path of this package never crosses GPUs (utterances shard, models are replicated: SURVEY.md 8e); what the ranks
exchange is a barrier around a timed region, the slowest rank's time and 4 bytes per utterance -- what the reference's
``multiprocessing.Pool`` hands back to its parent (src/test/test-gmm.py:128-133).  That needs no tensor library: rank 0
listens on an abstract Unix-domain socket named after the job (MASTER_PORT and, under torch.distributed.run,
TORCHELASTIC_RUN_ID -- the launcher's own store keeps MASTER_PORT itself), the others connect, and every collective is one
length-prefixed pickle per rank to rank 0 and the gathered list back.

``init(backend="socket" | "gloo")``: "gloo" keeps the round-1..3 path (torch.distributed over gloo) for those who want it;
the default imports nothing but the standard library."""
from __future__ import annotations

import os
import pickle
import socket
import struct
import time


def rank_env():
    """(rank, local_rank, world_size) as torch.distributed.run / bench.py's own spawner export them."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")))


def _job_key() -> str:
    return "sr-rdzv-%s-%s-%d" % (os.environ.get("MASTER_PORT", "29512"), os.environ.get("TORCHELASTIC_RUN_ID", "none"), os.getuid())


def _send(sock, obj):
    data = pickle.dumps(obj, protocol=pickle.HIGHEST_PROTOCOL)
    sock.sendall(struct.pack("<Q", len(data)) + data)


def _recv_exact(sock, n):
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(n - len(buf))
        if not chunk:
            raise ConnectionError("rendezvous: a rank closed its connection (did it die?)")
        buf += chunk
    return bytes(buf)


def _recv(sock):
    (n,) = struct.unpack("<Q", _recv_exact(sock, 8))
    return pickle.loads(_recv_exact(sock, n))


class SocketGroup:
    """The N ranks of one node over an abstract Unix-domain socket; rank 0 is the hub."""

    def __init__(self, rank: int, world: int, key: str | None = None, timeout: float = 600.0):
        self.rank, self.world = rank, world
        self._peers = []          # rank 0: connections indexed by rank - 1
        self._hub = None
        name = "\0" + (key or _job_key())
        if world <= 1:
            return
        if rank == 0:
            srv = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
            srv.bind(name)
            srv.listen(world)
            srv.settimeout(timeout)
            conns = {}
            while len(conns) < world - 1:
                c, _ = srv.accept()
                c.settimeout(timeout)
                r = _recv(c)
                conns[int(r)] = c
            srv.close()
            self._peers = [conns[r] for r in range(1, world)]
        else:
            deadline = time.monotonic() + timeout
            while True:
                c = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
                try:
                    c.connect(name)
                    break
                except (ConnectionRefusedError, FileNotFoundError):
                    c.close()
                    if time.monotonic() > deadline:
                        raise TimeoutError("rendezvous: rank 0 never opened %r" % name[1:])
                    time.sleep(0.01)
            c.settimeout(timeout)
            _send(c, rank)
            self._hub = c

    # every collective is an all-gather of one object per rank
    def all_gather(self, obj):
        if self.world <= 1:
            return [obj]
        if self.rank == 0:
            out = [obj] + [_recv(c) for c in self._peers]
            for c in self._peers:
                _send(c, out)
            return out
        _send(self._hub, obj)
        return _recv(self._hub)

    def barrier(self):
        self.all_gather(None)

    def all_max(self, x: float) -> float:
        return max(float(v) for v in self.all_gather(float(x)))

    def close(self):
        for c in self._peers:
            c.close()
        if self._hub is not None:
            self._hub.close()
        self._peers, self._hub = [], None


class GlooGroup:
    """The same interface on torch.distributed (gloo, CPU tensors): rounds 1-3's path, kept as an option."""

    def __init__(self, rank: int, world: int):
        import torch.distributed as dist
        self.rank, self.world = rank, world
        self._dist = dist
        if world > 1 and not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29512")
            dist.init_process_group(backend="gloo")

    def all_gather(self, obj):
        if self.world <= 1:
            return [obj]
        out = [None] * self.world
        self._dist.all_gather_object(out, obj)
        return out

    def barrier(self):
        if self.world > 1:
            self._dist.barrier()

    def all_max(self, x: float) -> float:
        return max(float(v) for v in self.all_gather(float(x)))

    def close(self):
        pass


_group = None


def init(backend: str | None = None, key: str | None = None):
    """The process's group (created once).  backend: "socket" (default; SR_RENDEZVOUS overrides) or "gloo"."""
    global _group
    if _group is None:
        rank, _, world = rank_env()
        backend = backend or os.environ.get("SR_RENDEZVOUS", "socket")
        _group = GlooGroup(rank, world) if backend == "gloo" else SocketGroup(rank, world, key)
    return _group


def reset():
    global _group
    if _group is not None:
        _group.close()
    _group = None
