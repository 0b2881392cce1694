"""CPU: the N>1 path -- utterance partition + host-side gather over gloo (world_size 2).
The per-shard compute is stood in by the oracle here (the GPU step is covered by -m gpu)."""
import os
import socket
import sys

import numpy as np

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


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
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


def test_two_rank_gather_matches_single_process():
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
    assert all(ok for _, ok, _ in res)
    assert res[0][2] == res[1][2]          # every rank ends with the same full answer
