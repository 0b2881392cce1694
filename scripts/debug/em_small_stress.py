#!/usr/bin/env python3
"""P processes enrol at the same time on one device (whole-fit kernel: an ordinary launch of 24 workgroups per 2998-frame fit, a
grid-wide barrier per iteration): every fit must come back, and how many were handed to the iteration-per-launch path because the grid
gave up at its barrier (other processes holding the CUs its last workgroups needed).  `em_small_stress.py [P] [fits]`"""
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def worker(rank, fits, q):
    from speaker_recognition_amd import _lib
    from speaker_recognition_amd.pygmm import GMM
    rng = np.random.default_rng(rank)
    cent = rng.normal(0, 2, (16, 13))
    X = (cent[rng.integers(0, 16, 2998)] + rng.normal(0, 0.7, (2998, 13))).astype(np.float32)
    GMM(16, nr_iteration=2, seed=1).fit(X)
    q.put(("ready", rank))
    whole = changed = 0
    t0 = time.perf_counter()
    ref = None
    for i in range(fits):
        g = GMM(16, nr_iteration=200, threshold=0.0, seed=1)
        g.fit(X)
        whole += _lib.last_em_stats_engine() == 4
        p = g.params()[1]
        if _lib.last_em_stats_engine() == 4:
            changed += ref is not None and not np.array_equal(ref, p)
            ref = p if ref is None else ref
    q.put(("done", rank, whole, time.perf_counter() - t0, changed))


if __name__ == "__main__":
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    fits = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=worker, args=(r, fits, q)) for r in range(P)]
    for p in ps:
        p.start()
    res = [q.get(timeout=90) for _ in range(2 * P)]
    for p in ps:
        p.join(60)
    done = [r for r in res if r[0] == "done"]
    print("%d processes x %d fits of 200 iterations: whole fits %d of %d, fits whose bits differ from the process's first %d, slowest process %.2f s, %.1f ms per fit and process" % (
        P, fits, sum(r[2] for r in done), P * fits, sum(r[4] for r in done), max(r[3] for r in done), 1e3 * max(r[3] for r in done) / fits))
