"""The scoring engines on the EM E-step's shape -- ONE model of K mixtures against N frames in one utterance, per-frame LL wanted:
em_score_shape.py [K=2048] [N=400000] [D=39]  ->  HIP-event time of the scoring kernel per engine (1 vector, 3 split-bf16, 5 split-fp16)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from speaker_recognition_amd import _lib, synth
from speaker_recognition_amd.core import Batch, ModelSet
from speaker_recognition_amd.pygmm import GMM
K = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
N = int(sys.argv[2]) if len(sys.argv) > 2 else 400000
D = int(sys.argv[3]) if len(sys.argv) > 3 else 39
m = synth.synth_gmm(K, D, 5)
ms = ModelSet([GMM.from_arrays(*m)])
X = synth.draw_frames(m, N, 6)
feats = Batch.from_features([X])
_lib.profile_enable(True)
for eng in (1, 3, 5, 0):
    _lib.set_option("score_engine", eng)
    ts = []
    for r in range(5):
        _lib.profile_reset()
        ms.score(feats, frame_ll=True)
        ts.append(_lib.profile_get(_lib.T_SCORE)[0])
    print("engine %d: %s ms  [%s]" % (eng, " ".join("%.3f" % t for t in ts[1:]), _lib.last_score_kernel()[:110]), flush=True)
