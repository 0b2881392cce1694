"""Scoring-kernel + offset-pre-pass time of U utterances x 300 frames against 201 x 512 x 39 under one forced shape of the
shared-sigma engine: [SR_GROUPS=n] h2s_small_one.py [SHAPE=0] [U ...]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from speaker_recognition_amd import _lib, synth
from speaker_recognition_amd.core import Batch, ModelSet
from speaker_recognition_amd.pygmm import GMM
shape = int(sys.argv[1]) if len(sys.argv) > 1 else 0
groups = int(os.environ.get("SR_GROUPS", "0"))          # score_model_groups (0 = the dispatcher's choice)
Us = [int(v) for v in sys.argv[2:]] or [1, 8, 64]
ubm = synth.synth_gmm(512, 39, 99)
ms = ModelSet([GMM.from_arrays(*m) for m in [ubm] + [synth.synth_map_speaker(ubm, 500 + s) for s in range(200)]])
_lib.profile_enable(True)
_lib.set_option("score_h2s_shape", shape)
_lib.set_option("score_model_groups", groups)
for U in Us:
    feats = Batch.from_features([synth.draw_frames(ubm, 300, 10 + u) for u in range(U)])
    ts, tr = [], []
    for r in range(9):
        _lib.profile_reset()
        ms.score(feats)
        if r > 1:
            ts.append(_lib.profile_get(_lib.T_SCORE)[0])
            tr.append(_lib.profile_get(_lib.T_SCORE_REF)[0])
    print("gm %s groups %d shape %d U = %3d: scoring %.4f ms  pre-pass %.4f ms  %s" % (os.environ.get("SR_GM", "-"), groups, shape, U, float(np.median(ts)), float(np.median(tr)), _lib.last_score_kernel().split(" ")[0]), flush=True)
