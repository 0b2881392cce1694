#!/usr/bin/env python3
"""MFCC kernel time (HIP events) on the cfg-1 audio, steady state: `time_mfcc.py [rounds]`."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from speaker_recognition_amd import _lib  # noqa: E402
from speaker_recognition_amd.core import Batch, MfccExtractor  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 16
clips, _ = bench.build_workload(0, 1000, 1000)
pcm = Batch.from_pcm(clips)
ex = MfccExtractor(bench.FS, **bench.MFCC_KW)
_lib.profile_enable(True)
ts = []
for r in range(rounds):
    _lib.profile_reset()
    ex.extract_batch(pcm, nd=2)
    t, n = _lib.profile_get(_lib.T_MFCC)
    ts.append(t)
print("mfcc kernel: first %.3f ms, steady median %.3f ms, min %.3f ms  [%s]" % (ts[0], float(np.median(ts[rounds // 2:])), min(ts), _lib.LIB_PATH))
