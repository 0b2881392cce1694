#!/usr/bin/env python3
"""EM statistics engines side by side in one GPU-box call: `ab_em.py [K] [N] [D]` fits 1 and 5 iterations (random-frame
start, same seed) with em_stats_engine = 1 (vector ALU, per-mixture centring), = 2 (fp64 matrix cores, responsibilities on the vector ALU) and = 0
(automatic: fp64 matrix cores with the responsibilities on the 16-bit ones where the model allows), prints the
per-iteration time, the statistics kernel's HIP-event time and the largest relative parameter difference between the
two fits."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from speaker_recognition_amd import _lib  # noqa: E402
from speaker_recognition_amd.pygmm import GMM  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
N = int(sys.argv[2]) if len(sys.argv) > 2 else 400000
D = int(sys.argv[3]) if len(sys.argv) > 3 else 39
rng = np.random.default_rng(0)
cent = rng.normal(0, 3, (64, D)).astype(np.float32)
X = (cent[rng.integers(0, 64, N)] + rng.normal(0, 1, (N, D))).astype(np.float32)
_lib.profile_enable(True)
fits = {}
for eng in (2, 0, 1, 2, 0):
    _lib.set_option("em_stats_engine", eng)
    res, ev = [], []
    for it in (1, 5):
        g = GMM(nr_mixture=K, nr_iteration=it, init_with_kmeans=0, seed=5, threshold=0.0)
        _lib.profile_reset()
        t0 = time.perf_counter(); g.fit(X); _lib.synchronize(); res.append(time.perf_counter() - t0)
        ev.append(_lib.profile_get(_lib.T_ESTEP))
    fits[eng] = g.params()
    flops = 4.0 * K * N * D
    per = ev[1][0] / max(1, ev[1][1])
    print("K=%d N=%d D=%d em_stats_engine=%d: %.2f ms per iteration (wall); statistics kernel %.3f ms per launch = %.1f TFLOP/s of 4KND"
          % (K, N, D, eng, (res[1] - res[0]) / 4 * 1e3, per, flops / (per * 1e-3) / 1e12), flush=True)
if fits[0] is not None and fits[1] is not None:
    for name, a, b in zip(("weights", "means", "sigmas"), fits[0], fits[1]):
        print("max rel diff of %s between the two engines after 5 iterations: %.2e" % (name, np.max(np.abs(a - b) / np.maximum(1e-3, np.abs(b)))))
