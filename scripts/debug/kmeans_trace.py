import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from speaker_recognition_amd import synth
from speaker_recognition_amd.pygmm import GMM
n, K, D = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
true = synth.synth_gmm(K, D, 5)
X = synth.draw_frames(true, n, 11)
GMM(8, nr_iteration=0, init_with_kmeans=1, seed=1, concurrency=8).fit(X[:4000])
g = GMM(K, nr_iteration=0, init_with_kmeans=1, seed=3, verbosity=2)
t0 = time.perf_counter(); g.fit(X); print("total %.3f s" % (time.perf_counter() - t0))
