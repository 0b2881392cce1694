#!/usr/bin/env python3
"""Wall time per call of the 256 x 39 point (1 model, 2 M resident frames in 2000 utterances) beside the kernel's HIP-event time:
what the host side of a scoring call costs on a large batch.  point256_wall.py [ROUNDS=40]"""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from speaker_recognition_amd import _lib, synth
from speaker_recognition_amd.core import Batch, ModelSet
from speaker_recognition_amd.pygmm import GMM
R = int(sys.argv[1]) if len(sys.argv) > 1 else 40
m = synth.synth_gmm(256, 39, 7)
ms = ModelSet([GMM.from_arrays(*m)])
X = synth.draw_frames(m, 100000, 3)
feats = Batch.from_features([X[(i * 1000) % 99000:(i * 1000) % 99000 + 1000] for i in range(2000)])
for prof in (False, True):
    _lib.profile_enable(prof)
    ts = []
    for r in range(R):
        t0 = time.perf_counter(); ms.score(feats); ts.append((time.perf_counter() - t0) * 1e3)
    print("event timers %d: wall per call p50 %.4f ms, min %.4f ms" % (prof, float(np.median(ts[5:])), min(ts[5:])))
