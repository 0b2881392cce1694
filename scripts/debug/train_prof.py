#!/usr/bin/env python3
"""Kernel mix of a means-only MAP call and an EM iteration at K = 2048 (run under rocprofv3 --kernel-trace --stats)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from speaker_recognition_amd import _lib
from speaker_recognition_amd.pygmm import GMM
K, D, N = 2048, 39, int(os.environ.get("N", 200000))
rng = np.random.default_rng(0)
cent = rng.normal(0, 3, (64, D)).astype(np.float32)
X = (cent[rng.integers(0, 64, N)] + rng.normal(0, 1, (N, D))).astype(np.float32)
g = GMM(nr_mixture=K, nr_iteration=int(os.environ.get("EM_IT", 1)), init_with_kmeans=0, seed=5)
g.fit(X)
for s in range(int(os.environ.get("MAPS", 5))):
    m = GMM(nr_mixture=K, nr_iteration=1)
    m.fit(X[s * 3000:(s + 1) * 3000], ubm=g)
