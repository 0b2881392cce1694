"""ms per iteration of the whole-fit kernel (threshold 0: a fixed number of iterations): em_small_time.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from speaker_recognition_amd import _lib
from speaker_recognition_amd.pygmm import GMM
rng = np.random.default_rng(1)
for n, K, D in ((2998, 16, 13), (1250, 32, 34), (8192, 32, 40), (64, 4, 3)):
    cent = rng.normal(0, 2, (K, D))
    X = (cent[rng.integers(0, K, n)] + rng.normal(0, 0.7, (n, D))).astype(np.float32)
    out = []
    for iters in (2, 102):
        best = 1e9
        for rep in range(4):
            g = GMM(K, nr_iteration=iters, threshold=0.0, seed=1)
            t0 = time.perf_counter(); g.fit(X); best = min(best, time.perf_counter() - t0)
        out.append(best)
    print("n %5d K %2d D %2d: fit of 2 iterations %.3f ms, per further iteration %.1f us (engine %d)" % (n, K, D, out[0] * 1e3, (out[1] - out[0]) / 100 * 1e6, _lib.last_em_stats_engine()))
