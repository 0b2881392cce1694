#!/usr/bin/env python3
"""Time of the reference-identical initialisers in front of EM (csrc/kmeans_init.hip): K random frames vs k-means||
(oversampling rounds + weighted k-means++ + Lloyd on the full data), at the reference's benchmark shape
(256 mixtures, 13 dims, 512 k frames) and a configs[2]-sized UBM (512 x 39, 2 M frames).  nr_iteration = 0: the
initialiser alone."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from speaker_recognition_amd import synth  # noqa: E402
from speaker_recognition_amd.pygmm import GMM  # noqa: E402

out = []
for n, K, D in ((512000, 256, 13), (2000000, 512, 39), (1000000, 2048, 39)):
    true = synth.synth_gmm(K, D, 5)
    X = synth.draw_frames(true, n, 11)
    GMM(8, nr_iteration=0, init_with_kmeans=1, seed=1, concurrency=8).fit(X[:4000])      # warm-up
    row = {"frames": n, "mixtures": K, "dims": D}
    for km in (0, 1):
        g = GMM(K, nr_iteration=0, init_with_kmeans=km, seed=3, concurrency=int(os.environ.get("KM_CONC", os.cpu_count() or 16)))
        t0 = time.perf_counter()
        g.fit(X)
        row["init_with_kmeans=%d_seconds" % km] = time.perf_counter() - t0
    row["concurrency (worker blocks whose order the sums keep)"] = int(os.environ.get("KM_CONC", os.cpu_count() or 16))
    out.append(row)
print(json.dumps(out))
