import os, sys, time, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from speaker_recognition_amd import _lib, synth
from speaker_recognition_amd.core import Batch, ModelSet
from speaker_recognition_amd.pygmm import GMM
N, S, K = 200000, 16, 64
_lib.profile_enable(True)
for D in (65, 72, 80, 84, 96, 100, 112, 128):
    models = [synth.synth_gmm(K, D, 7 + s) for s in range(S)]
    ms = ModelSet([GMM.from_arrays(*m) for m in models])
    rng = np.random.default_rng(D)
    X = (models[0][1][rng.integers(0, K, N)] + 0.3 * rng.standard_normal((N, D))).astype(np.float32)
    feats = Batch.from_features([X[i:i + 1000] for i in range(0, N, 1000)])
    ts = []
    for r in range(4):
        _lib.profile_reset(); ms.score(feats, clamp_compat=False); t, c = _lib.profile_get(_lib.T_SCORE); ts.append(t)
    g = GMM.from_arrays(*models[0]); g.nr_iteration, g.init_with_kmeans = 1, -1
    Xe = X[:50000]; g.fit(Xe); t0 = time.perf_counter(); g.fit(Xe); te = (time.perf_counter() - t0) * 1e3
    print("D %3d: %.3f ms  EM %.2f ms  [%s]" % (D, min(ts[1:]), te, _lib.last_score_kernel().split(" (")[0]))
