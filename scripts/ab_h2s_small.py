#!/usr/bin/env python3
"""The shared-sigma engine's workgroup shapes (score_h2s_shape 1 = 4 waves, 2 = 12 waves, 3 = 12 waves pipelined) on SMALL batches:
U utterances x 300 frames against 201 x 512 x 39 (the serving shape, gui/interface.py:85-94).  HIP-event time of the scoring kernel."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from speaker_recognition_amd import _lib, synth
from speaker_recognition_amd.core import Batch, ModelSet
from speaker_recognition_amd.pygmm import GMM
ubm = synth.synth_gmm(512, 39, 99)
ms = ModelSet([GMM.from_arrays(*m) for m in [ubm] + [synth.synth_map_speaker(ubm, 500 + s) for s in range(200)]])
_lib.profile_enable(True)
for U in (1, 2, 3, 4, 6, 8, 16, 32, 64):
    feats = Batch.from_features([synth.draw_frames(ubm, 300, 10 + u) for u in range(U)])
    out = []
    for shape in (1, 2, 3, 4):
        _lib.set_option("score_h2s_shape", shape)
        ts = []
        for r in range(7):
            _lib.profile_reset()
            sums, arg = ms.score(feats)
            t = _lib.profile_get(_lib.T_SCORE)[0] + _lib.profile_get(_lib.T_SCORE_REF)[0]
            if r > 1: ts.append(t)
        out.append("%d: %.4f" % (shape, float(np.median(ts))))
    _lib.set_option("score_h2s_shape", 0)
    ms.score(feats)
    print("U = %3d (%6d frames): ms by shape  %s   auto -> %s" % (U, U * 300, "   ".join(out), _lib.last_score_kernel().split(">")[0][-28:]), flush=True)
