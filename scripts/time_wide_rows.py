"""Scoring and one EM iteration on rows wider than a lane's registers (gmm_score_wide_kernel / em_stats_wide_kernel, round 6):
`time_wide_rows.py [FRAMES=200000]` prints, per dimension, the scoring kernel's HIP-event time, its rate against the fp32 vector
peak (2 FMAs per mixture and dimension = 4 flops, SURVEY 8d's count for the direct form) and the time of one EM iteration;
D = 128 (the widest register-resident kernel) beside them for scale."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speaker_recognition_amd import _lib, synth
from speaker_recognition_amd.core import Batch, ModelSet
from speaker_recognition_amd.pygmm import GMM
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
S, K = 16, 64
r6 = np.vectorize(lambda v: float("%g" % v))
_lib.profile_enable(True)
for D in (128, 192, 256, 512, 1024):
    models = [synth.synth_gmm(K, D, 7 + s) for s in range(S)]
    if D > 400:
        models = [(w, mu, r6(sg * 0.45)) for w, mu, sg in models]
    ms = ModelSet([GMM.from_arrays(*m) for m in models])
    rng = np.random.default_rng(D)
    X = (models[0][1][rng.integers(0, K, N)] + 0.3 * rng.standard_normal((N, D))).astype(np.float32)
    feats = Batch.from_features([X[i:i + 1000] for i in range(0, N, 1000)])
    ts = []
    for r in range(4):
        _lib.profile_reset(); ms.score(feats, clamp_compat=False); t, c = _lib.profile_get(_lib.T_SCORE); ts.append(t)
    t = min(ts[1:])
    tf = N * S * K * 4.0 * D / (t * 1e-3) / 1e12
    g = GMM.from_arrays(*models[0])
    g.nr_iteration, g.init_with_kmeans = 1, -1
    Xe = X[:50000]
    g.fit(Xe); t0 = time.perf_counter(); g.fit(Xe); te = (time.perf_counter() - t0) * 1e3
    print("D %4d: scoring %d frames x %d models x %d mixtures %.3f ms = %.1f TFLOP/s = %.2f of the fp32 vector peak (157.3), %.0f GB/s of "
          "feature reads; one EM iteration on 50 k frames %.2f ms  [%s]" % (D, N, S, K, t, tf, tf / 157.3, N * D * 4 / (t * 1e-3) / 1e9, te,
                                                                       _lib.last_score_kernel().split(" (")[0]))
