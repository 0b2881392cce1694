"""Phase times of EM iterations (K = 2048, 400 k frames, 39 dims; verbosity 2 prints them per iteration) and the HIP-event
totals of the scoring and statistics kernels over the fit: em_trace.py [K] [N] [D]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from speaker_recognition_amd import _lib
from speaker_recognition_amd.pygmm import GMM
K = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
N = int(sys.argv[2]) if len(sys.argv) > 2 else 400000
D = int(sys.argv[3]) if len(sys.argv) > 3 else 39
rng = np.random.default_rng(0)
cent = rng.normal(0, 3, (64, D)).astype(np.float32)
X = (cent[rng.integers(0, 64, N)] + rng.normal(0, 1, (N, D))).astype(np.float32)
g = GMM(nr_mixture=K, nr_iteration=1, init_with_kmeans=0, seed=5, threshold=0.0); g.fit(X)
_lib.profile_enable(True)
_lib.profile_reset()
g = GMM(nr_mixture=K, nr_iteration=4, init_with_kmeans=0, seed=5, threshold=0.0, verbosity=2); g.fit(X)
ts, cs = _lib.profile_get(_lib.T_SCORE)
te, ce = _lib.profile_get(_lib.T_ESTEP)
print("scoring kernel: %d launches, %.3f ms each [%s]; statistics kernel: %d launches, %.3f ms each" % (cs, ts / max(1, cs), _lib.last_score_kernel()[:60], ce, te / max(1, ce)))
