"""Scoring kernel under a forced engine: `time_score_engine.py ENGINE FT DEBUG [ROUNDS]` prints the HIP-event
time per launch (cfg-1 shape).  Used for PMC passes on one engine (scripts/pmc.sh with PMC_CMD)."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speaker_recognition_amd import _lib, synth
from speaker_recognition_amd.core import Batch, ModelSet
from speaker_recognition_amd.pygmm import GMM
eng, ft, dbg = (int(a) for a in sys.argv[1:4])
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 4
S, K, D, U, T = 100, 64, 39, 1000, 1000
models = [synth.synth_gmm(K, D, 7 + s) for s in range(S)]
ms = ModelSet([GMM.from_arrays(*m) for m in models])
base = [synth.draw_frames(models[s % S], T, 42 + s, outlier_frac=0.001) for s in range(50)]
feats = Batch.from_features([base[u % 50] for u in range(U)])
_lib.profile_enable(True)
_lib.set_option("score_engine", eng)
_lib.set_option("score_mfma_ft", ft)
ts = []
for r in range(rounds):
    _lib.profile_reset(); ms.score(feats); t, c = _lib.profile_get(_lib.T_SCORE); ts.append(t)
print("engine %d FT %d debug %d: %s ms  [%s]" % (eng, ft, dbg, " ".join("%.3f" % t for t in ts), _lib.last_score_kernel()))
