#!/usr/bin/env python3
"""The float64 iteration engine (em_f64.hip: short data, models of any size; em_stats_engine 0, last engine 5) against the
iteration-at-a-time path (em_stats_engine 3) and against the float64 oracle iterated; MAP enrolment from large UBMs timed both ways.
`em_f64_check.py [seed]`"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import gmm_oracle as go  # noqa: E402
from speaker_recognition_amd import _lib, synth  # noqa: E402
from speaker_recognition_amd.pygmm import GMM  # noqa: E402

go.build(ref=False)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 5)
fails = 0


def fit(eng, X, K, iters, threshold, ubm=None, start=None, km=0, seed=7):
    _lib.set_option("em_stats_engine", eng)
    if ubm is not None:
        g = GMM(K, nr_iteration=iters, threshold=threshold)
    elif start is not None:
        g = GMM.from_arrays(*start)
        g.nr_iteration, g.init_with_kmeans, g.threshold = iters, -1, threshold
    else:
        g = GMM(K, nr_iteration=iters, threshold=threshold, init_with_kmeans=km, seed=seed)
    t0 = time.perf_counter()
    it = g.fit(X, ubm=ubm) if ubm is not None else g.fit(X)
    return it, g.params(), time.perf_counter() - t0, _lib.last_em_stats_engine()


try:
    # MAP enrolment from a K-mixture UBM on 3000 frames, the drop-in defaults (200 iterations, threshold 0.01)
    for K in (64, 512, 2048):
        ubm_raw = synth.synth_gmm(K, 39, 99)
        ubm = GMM.from_arrays(*ubm_raw)
        X = synth.draw_frames(synth.synth_map_speaker(ubm_raw, 501), 3000, 11)
        r = {}
        for eng in (3, 0, 3, 0):
            r[eng] = fit(eng, X, K, 200, 0.01, ubm=ubm)
        e_mu = np.max(np.abs(r[0][1][1] - r[3][1][1]))
        ok = r[0][0] == r[3][0] and r[0][3] == 5 and r[3][3] != 5 and e_mu < 2e-4 and np.array_equal(r[0][1][0], r[3][1][0]) and np.array_equal(r[0][1][2], r[3][1][2])
        fails += not ok
        print("MAP K %4d: iterations %d / %d, engine %d / %d, means %.1e; %.2f ms (iteration at a time %.2f) %s" % (
            K, r[0][0], r[3][0], r[0][3], r[3][3], e_mu, r[0][2] * 1e3, r[3][2] * 1e3, "ok" if ok else "!!"))
    # EM, default stop rule, both paths from the same start
    for n, K, D, km in ((4000, 64, 13, 1), (8000, 40, 39, 0), (3000, 128, 20, 0), (500, 33, 5, 0), (2000, 8, 64, 0)):
        cent = rng.normal(0, 2, (K, D))
        X = (cent[rng.integers(0, K, n)] + rng.normal(0, 0.7, (n, D))).astype(np.float32)
        r = {eng: fit(eng, X, K, 200, 0.01, km=km) for eng in (3, 0)}
        e = (np.max(np.abs(r[0][1][0] - r[3][1][0])), np.max(np.abs(r[0][1][1] - r[3][1][1])), np.max(np.abs(r[0][1][2] - r[3][1][2]) / r[3][1][2]))
        ok = r[0][0] == r[3][0] and r[0][3] == 5 and e[0] < 2e-5 and e[1] < 2e-4 and e[2] < 1e-3
        fails += not ok
        print("EM n %5d K %3d D %2d km %d: iterations %3d / %3d, engine %d / %d, weights %.1e means %.1e sigmas %.1e; %.2f ms (%.2f) %s" % (
            n, K, D, km, r[0][0], r[3][0], r[0][3], r[3][3], *e, r[0][2] * 1e3, r[3][2] * 1e3, "ok" if ok else "!!"))
    # N iterations, stop rule off, against the oracle iterated
    for n, K, D, N in ((3000, 70, 39, 3), (900, 33, 13, 5), (130, 200, 20, 2), (8192, 64, 26, 2)):
        cent = 3.0 + rng.normal(0, 2, (K, D))
        X = (cent[rng.integers(0, K, n)] + rng.normal(0, 0.7, (n, D))).astype(np.float32)
        r6 = np.vectorize(lambda v: float("%g" % v))
        start = go.GMMParams(np.full(K, 1.0 / K), r6(cent + 0.2 * rng.standard_normal(cent.shape)), np.full((K, D), 0.9))
        want = start
        for _ in range(N):
            want = go.em_iteration(want, X.astype(np.float64))
        it, p, _, eng = fit(0, X, K, N, 0.0, start=(start.weights, start.mean, start.sigma))
        e = (np.max(np.abs(p[0] - want.weights)), np.max(np.abs(p[1] - want.mean)), np.max(np.abs(p[2] - want.sigma) / want.sigma))
        # (more mixtures than frames: sums of responsibilities ~1e-280, where the reference's partial-product flushes choose the survivors -- the training gates)
        tol = (1e-5, 1e-4, 1e-3) if K > n else (1e-9, 1e-8, 1e-8)
        ok = it == N and eng == 5 and e[0] < tol[0] and e[1] < tol[1] and e[2] < tol[2]
        fails += not ok
        print("EM  n %5d K %3d D %2d, %d iterations vs the oracle: weights %.1e means %.1e sigmas %.1e engine %d %s" % (n, K, D, N, *e, eng, "ok" if ok else "!!"))
        m = min(n, 300)
        want = start
        for _ in range(N):
            want = go.em_iteration(want, X[:m].astype(np.float64), map_relevance=16.0, ubm=start)
        ubm = GMM.from_arrays(start.weights, start.mean, start.sigma)
        it, p, _, eng = fit(0, X[:m], K, N, 0.0, ubm=ubm)
        e_mu = np.max(np.abs(p[1] - want.mean))
        ok = it == N and eng == 5 and e_mu < 1e-8 and np.array_equal(p[0], start.weights) and np.array_equal(p[2], start.sigma)
        fails += not ok
        print("MAP n %5d K %3d D %2d, %d iterations vs the oracle: means %.1e engine %d %s" % (m, K, D, N, e_mu, eng, "ok" if ok else "!!"))
    # frames far away (no responsibility) and frames in the flush band (handed over)
    K, D = 40, 13
    cent = rng.normal(0, 2, (K, D))
    X = (cent[rng.integers(0, K, 700)] + rng.normal(0, 0.7, (700, D))).astype(np.float32)
    X[::50] += 1000.0
    start = (np.full(K, 1.0 / K), cent, np.full((K, D), 0.9))
    r = {eng: fit(eng, X, K, 3, 0.0, start=start) for eng in (3, 0)}
    e = (np.max(np.abs(r[0][1][0] - r[3][1][0])), np.max(np.abs(r[0][1][1] - r[3][1][1])), np.max(np.abs(r[0][1][2] - r[3][1][2]) / r[3][1][2]))
    ok = r[0][3] == 5 and e[0] < 2e-5 and e[1] < 2e-4 and e[2] < 1e-3
    fails += not ok
    print("frames 1000 units away: engine %d, weights %.1e means %.1e sigmas %.1e %s" % (r[0][3], *e, "ok" if ok else "!!"))
    X = (cent[rng.integers(0, K, 700)] + rng.normal(0, 0.02, (700, D))).astype(np.float32)
    X[::50] = (cent[0] + 9.1).astype(np.float32)
    cent2 = np.tile(cent[0], (K, 1)) + rng.normal(0, 0.01, (K, D))
    start = (np.full(K, 1.0 / K), cent2, np.full((K, D), 0.9))
    X2 = (cent2[rng.integers(0, K, 700)] + rng.normal(0, 0.5, (700, D))).astype(np.float32)
    X2[::50] = (cent[0] + 9.1).astype(np.float32)
    r = {eng: fit(eng, X2, K, 1, 0.0, start=start) for eng in (3, 0)}
    same = all(np.array_equal(a, b) for a, b in zip(r[0][1], r[3][1]))
    ok = r[0][3] != 5 and same
    fails += not ok
    print("frames in the flush band: engine %d, same bits as engine 3's run: %s %s" % (r[0][3], same, "ok" if ok else "!!"))
    a = fit(0, X, K, 6, 0.0, start=(np.full(K, 1.0 / K), cent, np.full((K, D), 0.9)))
    b = fit(0, X, K, 6, 0.0, start=(np.full(K, 1.0 / K), cent, np.full((K, D), 0.9)))
    ok = all(np.array_equal(x, y) for x, y in zip(a[1], b[1]))
    fails += not ok
    print("two runs, same bits: %s" % ok)
finally:
    _lib.set_option("em_stats_engine", 0)
print("findings:", fails)
