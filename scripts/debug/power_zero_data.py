import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from speaker_recognition_amd import _lib, synth
from speaker_recognition_amd.core import Batch, ModelSet
from speaker_recognition_amd.pygmm import GMM
S, K, D, U, T = (1, 256, 39, 2000, 1000) if (len(sys.argv) > 1 and sys.argv[1] == 'point256') else (100, 64, 39, 1000, 1000)
models = [synth.synth_gmm(K, D, 77 + s) for s in range(S)]
ms = ModelSet([GMM.from_arrays(*m) for m in models])
zm = [(np.full(K, 1.0 / K), np.zeros((K, D)), np.ones((K, D))) for s in range(S)]
ms0 = ModelSet([GMM.from_arrays(*m) for m in zm])
base = [synth.draw_frames(models[u % S], T, 100 + u) for u in range(min(U, 100))]
feats = Batch.from_features([base[u % len(base)] for u in range(U)])
zeros = Batch.from_features([np.zeros((T, D), dtype=np.float32) for u in range(U)])
_lib.set_option("score_engine", 5)
_lib.profile_enable(True)
for shape in (1, 8):
    _lib.set_option("score_split_shape", shape)
    for name, m_, f_ in (("random data, random models", ms, feats), ("ZERO frames, random models", ms, zeros), ("ZERO frames, ZERO-mean unit models", ms0, zeros), ("random data again", ms, feats)):
        ts = []
        for r in range(6):
            _lib.profile_reset()
            m_.score(f_)
            t, c = _lib.profile_get(_lib.T_SCORE)
            if r: ts.append(t)
        print("shape %d  %-36s %.3f ms  (%s)" % (shape, name, float(np.median(ts)), " ".join("%.3f" % x for x in ts)), flush=True)
