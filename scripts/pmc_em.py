#!/usr/bin/env python3
"""One EM pass of the configs[3]-sized UBM shape (K = 2048, 39 dims, 400 k frames) for PMC / kernel-trace passes on
em_stats_mfma_kernel: `rocprofv3 --kernel-trace --pmc ... -- python scripts/pmc_em.py`."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speaker_recognition_amd.pygmm import GMM
K, N, D = 2048, 400000, 39
rng = np.random.default_rng(0)
cent = rng.normal(0, 3, (64, D)).astype(np.float32)
X = (cent[rng.integers(0, 64, N)] + rng.normal(0, 1, (N, D))).astype(np.float32)
g = GMM(nr_mixture=K, nr_iteration=3, init_with_kmeans=0, seed=5, threshold=0.0)
g.fit(X)
print("done")
