"""Multi-GPU by utterance sharding: models replicated, utterances partitioned,
no data-path collective.  One process per GPU (RANK / LOCAL_RANK / WORLD_SIZE from the
launcher); every rank scores its own utterances with the fused device step and only the
per-utterance results (argmax + optionally the S sums) are gathered on the host -- 4 bytes per
utterance, the host-side gather the reference's multiprocessing.Pool does
(src/test/test-gmm.py:129-133), with `gloo` carrying it.  xGMI / RCCL are not involved.
"""
from __future__ import annotations

import os

import numpy as np


def rank_env():
    """(rank, local_rank, world_size) from the torch.distributed.run environment."""
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


def init_process_group():
    """gloo group for the host-side gather (rendezvous on 127.0.0.1 unless told otherwise)."""
    import torch.distributed as dist
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        dist.init_process_group(backend="gloo")
    return dist


def predict_sharded(n_utt: int, lengths, compute, n_models: int, want_sums: bool = False):
    """Run ``compute(indices) -> (sums[len(indices), S] or None, argmax[len(indices)])`` on this
    rank's shard and gather the full result on every rank.

    Returns (argmax[n_utt] int32, sums[n_utt, S] float64 or None).  With WORLD_SIZE == 1 this is
    a plain call (no torch import)."""
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
    import torch
    dist = init_process_group()
    # fixed-size exchange: every rank contributes a dense [n_utt] vector (-2 = not mine)
    buf = torch.full((n_utt,), -2, dtype=torch.int32)
    buf[torch.from_numpy(mine)] = torch.from_numpy(arg)
    dist.all_reduce(buf, op=dist.ReduceOp.MAX)     # shards are disjoint: MAX picks the owner's value
    full = buf.numpy().copy()
    fs = None
    if want_sums:
        sb = torch.zeros((n_utt, n_models), dtype=torch.float64)
        sb[torch.from_numpy(mine)] = torch.from_numpy(np.asarray(sums, dtype=np.float64))
        dist.all_reduce(sb, op=dist.ReduceOp.SUM)
        fs = sb.numpy().copy()
    return full, fs
