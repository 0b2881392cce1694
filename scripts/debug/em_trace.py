import sys, numpy as np
sys.path.insert(0, '/root/repo')
from speaker_recognition_amd.pygmm import GMM
rng = np.random.default_rng(0); D=39; N=400000; K=2048
cent = rng.normal(0, 3, (64, D)).astype(np.float32)
X = (cent[rng.integers(0, 64, N)] + rng.normal(0, 1, (N, D))).astype(np.float32)
g = GMM(nr_mixture=K, nr_iteration=1, init_with_kmeans=0, seed=5, threshold=0.0); g.fit(X)
g = GMM(nr_mixture=K, nr_iteration=4, init_with_kmeans=0, seed=5, threshold=0.0, verbosity=2); g.fit(X)
