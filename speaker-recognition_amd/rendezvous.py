"""Host-side meeting point of the N single-GPU processes of ONE node -- barrier, gather of small Python objects, max of a
float -- without torch, MPI or any GPU collective.

The data path of this package never crosses GPUs (utterances shard, models are replicated: SURVEY.md 8e); what the ranks
exchange is a barrier around a timed region, the slowest rank's time and 4 bytes per utterance -- what the reference's
``multiprocessing.Pool`` hands back to its parent (src/test/test-gmm.py:128-133).  That needs no tensor library: rank 0
listens on an abstract Unix-domain socket named after the job (MASTER_PORT and, under torch.distributed.run,
TORCHELASTIC_RUN_ID or SR_RDZV_NONCE -- the launcher's own store keeps MASTER_PORT itself), the others connect, and every
collective is one length-prefixed message per rank to rank 0 and the gathered list back.  Messages are JSON + raw array bytes
(nothing is unpickled), both ends check the peer's uid (SO_PEERCRED), and rank 0 seats only ranks 1..N-1 of its own world size.

``init(backend="socket" | "gloo")``: "gloo" keeps the round-1..3 path (torch.distributed over gloo) for those who want it;
the default imports the standard library and numpy only."""
from __future__ import annotations

import json
import os
import socket
import struct
import time

import numpy as np

MAX_MESSAGE = 1 << 30          # bytes; a gather of per-utterance sums of a very large job stays far below


def rank_env():
    """(rank, local_rank, world_size) as torch.distributed.run / bench.py's own spawner export them."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")))


def _job_key() -> str:
    """One name per job and user: MASTER_PORT, the launcher's run id (torch.distributed.run) or this package's own nonce
    (SR_RDZV_NONCE: bench.py's spawner draws one per launch, so two jobs of one user on the default port do not meet), uid."""
    run = os.environ.get("SR_RDZV_NONCE") or os.environ.get("TORCHELASTIC_RUN_ID", "none")
    return "sr-rdzv-%s-%s-%d" % (os.environ.get("MASTER_PORT", "29512"), run, os.getuid())


# ---- wire format: no pickle.  A message is JSON (None / bool / int / float / str / list / dict with string keys) in which every
# numpy array is a placeholder {"__nd__": i, "dtype": ..., "shape": [...]} whose bytes follow the JSON text; tuples travel as lists.
_DTYPES = {"b", "i", "u", "f"}


def _encode(obj) -> bytes:
    blobs = []

    def walk(o):
        if isinstance(o, np.ndarray):
            if o.dtype.kind not in _DTYPES:
                raise TypeError("rendezvous: arrays of dtype %s do not travel" % o.dtype)
            blobs.append(np.ascontiguousarray(o).tobytes())
            return {"__nd__": len(blobs) - 1, "dtype": o.dtype.str, "shape": list(o.shape)}
        if isinstance(o, np.generic):
            return o.item()
        if isinstance(o, (list, tuple)):
            return [walk(v) for v in o]
        if isinstance(o, dict):
            return {str(k): walk(v) for k, v in o.items()}
        if o is None or isinstance(o, (bool, int, float, str)):
            return o
        raise TypeError("rendezvous: objects of type %s do not travel" % type(o).__name__)
    head = json.dumps({"v": walk(obj), "blobs": [len(b) for b in blobs]}).encode()
    return struct.pack("<I", len(head)) + head + b"".join(blobs)


def _decode(data: bytes):
    (hl,) = struct.unpack_from("<I", data, 0)
    head = json.loads(data[4:4 + hl].decode())
    pos, blobs = 4 + hl, []
    for n in head["blobs"]:
        blobs.append(data[pos:pos + n])
        pos += n

    def walk(o):
        if isinstance(o, dict):
            if "__nd__" in o:
                dt = np.dtype(o["dtype"])
                if dt.kind not in _DTYPES:
                    raise ValueError("rendezvous: refused dtype %s" % o["dtype"])
                return np.frombuffer(blobs[int(o["__nd__"])], dtype=dt).reshape(o["shape"]).copy()
            return {k: walk(v) for k, v in o.items()}
        if isinstance(o, list):
            return [walk(v) for v in o]
        return o
    return walk(head["v"])


def _send(sock, obj):
    data = _encode(obj)
    sock.sendall(struct.pack("<Q", len(data)) + data)


def _recv_exact(sock, n):
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(min(n - len(buf), 1 << 20))
        if not chunk:
            raise ConnectionError("rendezvous: a rank closed its connection (did it die?)")
        buf += chunk
    return bytes(buf)


def _recv(sock):
    (n,) = struct.unpack("<Q", _recv_exact(sock, 8))
    if n > MAX_MESSAGE:
        raise ConnectionError("rendezvous: a %d-byte message was announced (limit %d)" % (n, MAX_MESSAGE))
    return _decode(_recv_exact(sock, n))


def _peer_is_me(sock) -> bool:
    """An abstract socket name carries no permissions: anybody on the host may bind or connect to it.  The kernel tells who did."""
    pid, uid, gid = struct.unpack("3i", sock.getsockopt(socket.SOL_SOCKET, socket.SO_PEERCRED, struct.calcsize("3i")))
    return uid == os.getuid()


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
                try:
                    if not _peer_is_me(c):
                        raise ConnectionError("another user's process")
                    hello = _recv(c)
                    r = int(hello["rank"])
                    if hello.get("world") != world or not 1 <= r < world or r in conns:
                        raise ConnectionError("not a rank of this job: %r" % (hello,))
                except Exception:
                    c.close()                      # a stranger, a stray rank of another job, a duplicate: no slot for it
                    continue
                conns[r] = c
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
            if not _peer_is_me(c):
                c.close()
                raise ConnectionError("rendezvous: %r is held by another user's process" % name[1:])
            _send(c, {"rank": rank, "world": world})
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
