#!/usr/bin/env python3
"""The whole-fit kernel (em_small.hip, em_stats_engine 0 on speaker-sized models) against (a) the iteration-at-a-time path
(em_stats_engine 3) from the same start: same number of iterations, parameters inside the training gates; (b) the float64 oracle
iterated N times with the stop rule off; and what a fit costs either way.  `em_small_check.py [seed]`"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import gmm_oracle as go  # noqa: E402
from speaker_recognition_amd import _lib  # noqa: E402
from speaker_recognition_amd.pygmm import GMM  # noqa: E402

go.build(ref=False)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 5)
fails = 0


def data(n, K, D, spread=2.0, shift=0.0):
    cent = shift + rng.normal(0, spread, (K, D))
    return (cent[rng.integers(0, K, n)] + rng.normal(0, 0.7, (n, D))).astype(np.float32), cent


def fit(eng, X, K, iters, threshold, km, seed, ubm=None, start=None):
    _lib.set_option("em_stats_engine", eng)
    if start is not None:
        g = GMM.from_arrays(*start)
        g.nr_iteration, g.init_with_kmeans, g.threshold = iters, -1, threshold
    else:
        g = GMM(K, nr_iteration=iters, threshold=threshold, init_with_kmeans=km, seed=seed)
    t0 = time.perf_counter()
    it = g.fit(X, ubm=ubm) if ubm is not None else g.fit(X)
    dt = time.perf_counter() - t0
    return it, g.params(), dt, _lib.last_em_stats_engine()


try:
    # (a) both paths from the same initialisation (random frames / k-means||, the reference's own random numbers), default stop rule
    for n, K, D, km in ((2998, 16, 13, 0), (2998, 16, 13, 1), (1250, 32, 34, 1), (5000, 8, 20, 0), (64, 4, 3, 0), (333, 5, 39, 0),
                        (1, 1, 1, 0)):
        if n < 2:
            continue
        X, _ = data(n, K, D)
        res = {}
        for eng in (3, 0, 3, 0):
            res[eng] = fit(eng, X, K, 200, 0.01, km, 7)
        (it3, p3, t3, e3), (it0, p0, t0, e0) = res[3], res[0]
        e_w = np.max(np.abs(p0[0] - p3[0]))
        e_mu = np.max(np.abs(p0[1] - p3[1]))
        e_sg = np.max(np.abs(p0[2] - p3[2]) / p3[2])
        ok = it0 == it3 and e0 == 4 and e3 != 4 and e_w < 2e-5 and e_mu < 2e-4 and e_sg < 1e-3
        fails += not ok
        print("n %5d K %2d D %2d km %d: iterations %3d / %3d, engine %d / %d, weights %.1e means %.1e sigmas %.1e; fit %.2f ms (iteration at a time %.2f) %s" % (
            n, K, D, km, it0, it3, e0, e3, e_w, e_mu, e_sg, t0 * 1e3, t3 * 1e3, "ok" if ok else "!!"))
    # (b) N iterations with the stop rule off against the oracle iterated N times (EM and MAP)
    for n, K, D, N in ((3000, 16, 13, 7), (900, 32, 39, 4), (150, 3, 40, 6), (4097, 9, 26, 5)):
        X, cent = data(n, K, D, shift=3.0)
        r6 = np.vectorize(lambda v: float("%g" % v))
        start = go.GMMParams(np.full(K, 1.0 / K), r6(cent + 0.2 * rng.standard_normal(cent.shape)), np.full((K, D), 0.9))
        want = start
        for _ in range(N):
            want = go.em_iteration(want, X.astype(np.float64))
        it, p, _, eng = fit(0, X, K, N, 0.0, 0, 1, start=(start.weights, start.mean, start.sigma))
        e = (np.max(np.abs(p[0] - want.weights)), np.max(np.abs(p[1] - want.mean)), np.max(np.abs(p[2] - want.sigma) / want.sigma))
        ok = it == N and eng == 4 and e[0] < 1e-9 and e[1] < 1e-8 and e[2] < 1e-8
        fails += not ok
        print("EM  n %5d K %2d D %2d, %d iterations vs the oracle: weights %.1e means %.1e sigmas %.1e engine %d %s" % (n, K, D, N, *e, eng, "ok" if ok else "!!"))
        m = min(n, 300)
        want = start
        for _ in range(N):
            want = go.em_iteration(want, X[:m].astype(np.float64), map_relevance=16.0, ubm=start)
        ubm = GMM.from_arrays(start.weights, start.mean, start.sigma)
        it, p, _, eng = fit(0, X[:m], K, N, 0.0, 0, 1, ubm=ubm)
        e_mu = np.max(np.abs(p[1] - want.mean))
        ok = it == N and eng == 4 and e_mu < 1e-8 and np.array_equal(p[0], start.weights) and np.array_equal(p[2], start.sigma)
        fails += not ok
        print("MAP n %5d K %2d D %2d, %d iterations vs the oracle: means %.1e engine %d %s" % (m, K, D, N, e_mu, eng, "ok" if ok else "!!"))
    # (c) frames far from every mixture carry no responsibility (gmm.cc:482-498): the kernel's own rule, against the other path
    X, cent = data(500, 8, 13)
    X[::50] += 1000.0
    start = (np.full(8, 1.0 / 8), cent, np.full((8, 13), 0.9))
    r = {eng: fit(eng, X, 8, 4, 0.0, 0, 1, start=start) for eng in (3, 0)}
    e = (np.max(np.abs(r[0][1][0] - r[3][1][0])), np.max(np.abs(r[0][1][1] - r[3][1][1])), np.max(np.abs(r[0][1][2] - r[3][1][2]) / r[3][1][2]))
    ok = r[0][3] == 4 and e[0] < 2e-5 and e[1] < 2e-4 and e[2] < 1e-3
    fails += not ok
    print("frames 1000 units away: engine %d, weights %.1e means %.1e sigmas %.1e against engine %d %s" % (r[0][3], *e, r[3][3], "ok" if ok else "!!"))
    # ... and frames in the band where the reference's partial-product flushes decide: the kernel hands the fit to the other path
    X, cent = data(500, 1, 13)
    X[::50] = (cent[0] + 9.1).astype(np.float32)
    start = (np.ones(1), cent, np.full((1, 13), 0.9))
    r = {eng: fit(eng, X, 1, 1, 0.0, 0, 1, start=start) for eng in (3, 0)}
    same = all(np.array_equal(a, b) for a, b in zip(r[0][1], r[3][1]))
    ok = r[0][3] != 4 and same
    fails += not ok
    print("frames in the flush band: engine %d, same bits as engine 3's run: %s %s" % (r[0][3], same, "ok" if ok else "!!"))
    # (d) bit-identical reruns
    X, _ = data(2998, 16, 13)
    a = fit(0, X, 16, 200, 0.01, 1, 3)
    b = fit(0, X, 16, 200, 0.01, 1, 3)
    ok = a[0] == b[0] and all(np.array_equal(x, y) for x, y in zip(a[1], b[1]))
    fails += not ok
    print("two runs, same bits: %s" % ok)
finally:
    _lib.set_option("em_stats_engine", 0)
print("findings:", fails)
