"""CPU: the N>1 path -- utterance partition + host-side gather (world_size 2 and 3), over the package's own torch-free
rendezvous (speaker-recognition_amd/rendezvous.py, with torch made unimportable in the workers) and over gloo.
The per-shard compute is stood in by the oracle here (the GPU step is covered by -m gpu)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_partition_covers_and_balances():
    from speaker_recognition_amd.shard import partition_utterances
    rng = np.random.default_rng(0)
    lengths = rng.integers(1, 3000, size=101)
    for n in (1, 2, 3, 8):
        parts = partition_utterances(lengths, n)
        allidx = np.sort(np.concatenate(parts))
        assert np.array_equal(allidx, np.arange(101))
        loads = [int(lengths[p].sum()) for p in parts]
        assert max(loads) - min(loads) <= int(lengths.max())
    assert [p.tolist() for p in partition_utterances([5, 5, 5, 5], 2)] == [[0, 2], [1, 3]]
    assert [len(p) for p in partition_utterances([], 4)] == [0, 0, 0, 0]


def _worker(rank, world, port, q, backend="socket"):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SR_RENDEZVOUS=backend)
    sys.path.insert(0, ROOT)
    if backend == "socket":
        sys.modules["torch"] = None          # `import torch` raises ImportError from here on: the socket path must not need it
    from oracle import gmm_oracle as go
    from speaker_recognition_amd import synth
    from speaker_recognition_amd.shard import predict_sharded
    models = [go.GMMParams(*synth.synth_gmm(4, 5, 40 + s)) for s in range(6)]
    rng = np.random.default_rng(3)
    lengths = rng.integers(5, 60, size=23)
    utts = [synth.draw_frames((models[u % 6].weights, models[u % 6].mean, models[u % 6].sigma),
                              int(lengths[u]), 100 + u) for u in range(23)]

    def compute(idx):
        sums = np.array([[go.score_all(m, utts[u]) for m in models] for u in idx]).reshape(len(idx), 6)
        return sums, np.argmax(sums, axis=1) if len(idx) else np.zeros(0, np.int32)

    arg, sums = predict_sharded(23, lengths, compute, 6, want_sums=True)
    full = np.array([[go.score_all(m, utts[u]) for m in models] for u in range(23)])
    ok = bool(np.array_equal(arg, np.argmax(full, axis=1)) and np.allclose(sums, full, rtol=0, atol=0))
    q.put((rank, ok, arg.tolist()))


def _run_ranks(world, backend):
    import multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, backend)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
    assert all(ok for _, ok, _ in res)
    assert all(r[2] == res[0][2] for r in res)          # every rank ends with the same full answer


def test_two_and_three_rank_gather_without_torch():
    _run_ranks(2, "socket")
    _run_ranks(3, "socket")


def test_two_rank_gather_over_gloo():
    _run_ranks(2, "gloo")


def _rdzv_worker(rank, world, key, q):
    sys.path.insert(0, ROOT)
    sys.modules["torch"] = None
    from speaker_recognition_amd.rendezvous import SocketGroup
    g = SocketGroup(rank, world, key=key, timeout=60)
    g.barrier()
    got = g.all_gather({"rank": rank, "payload": list(range(rank))})
    mx = g.all_max(10.0 + rank)
    g.barrier()
    g.close()
    q.put((rank, [d["rank"] for d in got], mx))


def test_socket_group_collectives():
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    key = "sr-test-%d" % os.getpid()
    procs = [ctx.Process(target=_rdzv_worker, args=(r, 4, key, q)) for r in range(4)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(30)
    assert [r[1] for r in res] == [[0, 1, 2, 3]] * 4 and all(r[2] == 13.0 for r in res)


def test_rendezvous_wire_format_carries_no_pickle():
    """What the ranks exchange -- None, numbers, dicts of them, (index, argmax, sums) array triples -- survives the JSON + raw-bytes
    encoding bit for bit; anything else is refused at the sender, and no byte of a message is ever unpickled."""
    import inspect
    from speaker_recognition_amd import rendezvous as rv
    assert "pickle" not in inspect.getsource(rv).replace("unpickled", "").replace("no pickle", "")
    rng = np.random.default_rng(0)
    msg = [None, 1.5, {"rate": 3.25e7, "device": 2, "ok": True, "numa_node": -1},
           (np.arange(5, dtype=np.int64), rng.integers(0, 9, 5).astype(np.int32), rng.standard_normal((5, 3))), np.float64(2.0)]
    back = rv._decode(rv._encode(msg))
    assert back[0] is None and back[1] == 1.5 and back[2] == msg[2] and back[4] == 2.0
    for a, b in zip(msg[3], back[3]):
        assert a.dtype == b.dtype and np.array_equal(a, b)
    with pytest.raises(TypeError):
        rv._encode({"f": lambda: 0})
    with pytest.raises(TypeError):
        rv._encode(np.array(["a"], dtype=object))


def _stray_then_real(key, q):
    """a process that is not a rank of the job (wrong world size, then a rank out of range) knocks first"""
    sys.path.insert(0, ROOT)
    import socket
    import time
    from speaker_recognition_amd import rendezvous as rv
    for hello in ({"rank": 1, "world": 7}, {"rank": 5, "world": 2}, "garbage"):
        while True:
            c = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
            try:
                c.connect("\0" + key)
                break
            except (ConnectionRefusedError, FileNotFoundError):
                c.close()
                time.sleep(0.01)
        rv._send(c, hello)
        c.close()
    g = rv.SocketGroup(1, 2, key=key, timeout=60)
    q.put(g.all_gather("one"))
    g.close()


def test_rank0_seats_only_ranks_of_its_own_job():
    import multiprocessing as mp
    from speaker_recognition_amd.rendezvous import SocketGroup
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    key = "sr-test-stray-%d" % os.getpid()
    p = ctx.Process(target=_stray_then_real, args=(key, q))
    p.start()
    g = SocketGroup(0, 2, key=key, timeout=60)
    assert g.all_gather("zero") == ["zero", "one"]
    assert q.get(timeout=60) == ["zero", "one"]
    g.close()
    p.join(30)
