#!/usr/bin/env python3
"""MFCC kernel time per FFT size on the cfg-1 audio (1000 utterances x 1000 frames, 25/10 ms at 16 kHz, 50 filters, 13 ceps):
the register-resident kernels vs the generic LDS-pass one (mfcc_generic = 1)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from speaker_recognition_amd import _lib  # noqa: E402
from speaker_recognition_amd.core import Batch, MfccExtractor  # noqa: E402

clips, _ = bench.build_workload(0, 1000, 1000)
pcm = Batch.from_pcm(clips)
_lib.profile_enable(True)
for fft in (512, 1024, 2048, 4096):
    for generic in (0, 1):
        _lib.set_option("mfcc_generic", generic)
        kw = dict(bench.MFCC_KW, FFT_SIZE=fft)
        ex = MfccExtractor(bench.FS, **kw)
        ts = []
        for r in range(8):
            _lib.profile_reset()
            ex.extract_batch(pcm, nd=2)
            ts.append(_lib.profile_get(_lib.T_MFCC)[0])
        print("fft %4d  generic=%d  mfcc kernel median %.3f ms" % (fft, generic, float(np.median(ts[3:]))))
_lib.set_option("mfcc_generic", 0)
