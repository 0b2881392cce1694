#!/usr/bin/env python3
"""A/B of the k-means|| start's nearest-centre search inside one GPU-box call: exact pass only (kmeans_assign_engine = 1) against the
fast full search with the exact pass behind it (0), K = 2048 x 39 dims on 1 M frames; reports seconds, how many searches went the fast
way, how many points they left to the exact pass, and whether the initial means are identical."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speaker_recognition_amd import synth, _lib
from speaker_recognition_amd.pygmm import GMM
import numpy as np
n, K, D = 1000000, 2048, 39
true = synth.synth_gmm(K, D, 5)
X = synth.draw_frames(true, n, 11)
GMM(8, nr_iteration=0, init_with_kmeans=1, seed=1, concurrency=8).fit(X[:4000])
res = {}
for eng in (1, 0, 1, 0):
    _lib.set_option("kmeans_assign_engine", eng)
    b = _lib.kmeans_fast_stats()
    g = GMM(K, nr_iteration=0, init_with_kmeans=1, seed=3, concurrency=256)
    t0 = time.perf_counter(); g.fit(X); t = time.perf_counter() - t0
    a = _lib.kmeans_fast_stats()
    p = g.params()
    print("engine", eng, "seconds %.3f" % t, "fast passes", a[0]-b[0], "points rechecked", a[1]-b[1], flush=True)
    res.setdefault(eng, p)
print("identical:", all(np.array_equal(x, y) for x, y in zip(res[0], res[1])))
