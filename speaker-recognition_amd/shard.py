"""Multi-GPU by utterance sharding: models replicated, utterances partitioned,
no data-path collective.  One process per GPU (RANK / LOCAL_RANK / WORLD_SIZE from the
launcher); every rank scores its own utterances with the fused device step and only the
per-utterance results (argmax + optionally the S sums) are gathered on the host -- 4 bytes per
utterance, the host-side gather the reference's multiprocessing.Pool does
(src/test/test-gmm.py:129-133) -- over rendezvous.py's Unix-domain socket (no torch; ``backend="gloo"``
or SR_RENDEZVOUS=gloo keeps torch.distributed as the carrier).  xGMI / RCCL are not involved.
"""
from __future__ import annotations

import os

import numpy as np


def rank_env():
    """(rank, local_rank, world_size) from the launcher's environment (torch.distributed.run's names)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def partition_utterances(lengths, n_parts: int):
    """Greedy longest-first assignment of utterances to ``n_parts`` shards balanced by frame
    count.  Deterministic (ties -> lower utterance index, lower shard index).  Returns a list of
    ``n_parts`` sorted int64 index arrays covering every utterance exactly once."""
    lengths = np.asarray(lengths, dtype=np.int64)
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    load = [0] * n_parts
    parts = [[] for _ in range(n_parts)]
    for i in order:
        p = min(range(n_parts), key=lambda q: (load[q], q))
        parts[p].append(i)
        load[p] += int(lengths[i])
    return [np.array(sorted(p), dtype=np.int64) for p in parts]


def predict_sharded(n_utt: int, lengths, compute, n_models: int, want_sums: bool = False, backend: str | None = None):
    """Run ``compute(indices) -> (sums[len(indices), S] or None, argmax[len(indices)])`` on this
    rank's shard and gather the full result on every rank.

    Returns (argmax[n_utt] int32, sums[n_utt, S] float64 or None).  With WORLD_SIZE == 1 this is
    a plain call."""
    rank, _, world = rank_env()
    parts = partition_utterances(lengths, world)
    mine = parts[rank]
    sums, arg = compute(mine)
    arg = np.asarray(arg, dtype=np.int32)
    if world == 1:
        full = np.full(n_utt, -1, dtype=np.int32)
        full[mine] = arg
        fs = None
        if want_sums:
            fs = np.zeros((n_utt, n_models))
            fs[mine] = sums
        return full, fs
    from . import rendezvous
    grp = rendezvous.init(backend)
    # every rank contributes its (indices, argmax[, sums]); shards are disjoint and cover every utterance
    full = np.full(n_utt, -1, dtype=np.int32)
    fs = np.zeros((n_utt, n_models)) if want_sums else None
    for idx, a, sm in grp.all_gather((mine, arg, np.asarray(sums, dtype=np.float64) if want_sums else None)):
        full[idx] = a
        if want_sums and len(idx):
            fs[idx] = sm
    return full, fs
