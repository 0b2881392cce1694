#!/usr/bin/env python3
"""Enrolment at the large configs' sizes (rows f-1 / f-2): EM of a K-mixture UBM on N frames (random-frame and k-means||
starts, per-iteration time) and means-only MAP of speakers (per call).  `bench_train_scale.py [K] [N] [iters]`"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from speaker_recognition_amd import _lib
from speaker_recognition_amd.pygmm import GMM
K = int(sys.argv[1]) if len(sys.argv) > 1 else 512
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
IT = int(sys.argv[3]) if len(sys.argv) > 3 else 4
D = 39
rng = np.random.default_rng(0)
cent = rng.normal(0, 3, (64, D)).astype(np.float32)
X = (cent[rng.integers(0, 64, N)] + rng.normal(0, 1, (N, D))).astype(np.float32)
for init in ((0,) if os.environ.get("SKIP_KMEANS") else (0, 1)):
    res = []
    for it in (1, 1 + IT):
        g = GMM(nr_mixture=K, nr_iteration=it, init_with_kmeans=init, seed=5, threshold=0.0)
        t0 = time.perf_counter(); g.fit(X); _lib.synchronize(); res.append(time.perf_counter() - t0)
    print("EM K=%d N=%d init_with_kmeans=%d: 1 iteration %.3f s, %d iterations %.3f s -> %.1f ms per iteration, start %.3f s"
          % (K, N, init, res[0], 1 + IT, res[1], (res[1] - res[0]) / IT * 1e3, res[0] - (res[1] - res[0]) / IT))
ubm = g
for n in (3000, 30000):
    t = []
    for s in range(6):
        m = GMM(nr_mixture=K, nr_iteration=1)
        t0 = time.perf_counter(); m.fit(X[s * n:(s + 1) * n], ubm=ubm); t.append(time.perf_counter() - t0)
    print("MAP K=%d, %d frames, 1 iteration: %.2f ms per speaker (first %.2f)" % (K, n, np.median(t[1:]) * 1e3, t[0] * 1e3))
if os.environ.get("TRAIN_TRACE"):
    t_all = time.perf_counter()
    m = GMM(nr_mixture=K, nr_iteration=2, verbosity=2)
    t0 = time.perf_counter(); m.fit(X[:3000], ubm=ubm); print("traced MAP call: %.2f ms" % ((time.perf_counter() - t0) * 1e3))
    g = GMM(nr_mixture=K, nr_iteration=2, init_with_kmeans=0, seed=5, verbosity=2)
    g.fit(X)
