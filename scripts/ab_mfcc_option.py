#!/usr/bin/env python3
"""A/B of an MFCC-side library option inside ONE GPU-box call: `ab_mfcc_option.py OPTION V0 [V1 ...] [--utts N]` times the MFCC kernel
(HIP events) on the configs[1] audio (N x 1000 frames, default 1000) under each value in turn, several rounds, and checks that the
features do not change."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from speaker_recognition_amd import _lib  # noqa: E402
from speaker_recognition_amd.core import Batch, MfccExtractor  # noqa: E402

args = sys.argv[1:]
utts = 1000
if "--utts" in args:
    i = args.index("--utts")
    utts = int(args[i + 1])
    del args[i:i + 2]
opt, vals = args[0], [int(v) for v in args[1:]]
clips, _ = bench.build_workload(0, utts, 1000)
pcm = Batch.from_pcm(clips)
ex = MfccExtractor(bench.FS, **bench.MFCC_KW)
_lib.profile_enable(True)
ts, out = {v: [] for v in vals}, {}
for r in range(7):
    for v in vals:
        _lib.set_option(opt, v)
        _lib.profile_reset()
        f = ex.extract_batch(pcm, nd=2)
        _lib.synchronize()
        if r > 1:
            ts[v].append(_lib.profile_get(_lib.T_MFCC)[0])
        if r == 0:
            out[v] = f.download()
for v in vals:
    print("%s=%d: %s ms (median %.3f) max |d| vs %s=%d: %.1e" % (opt, v, " ".join("%.3f" % t for t in ts[v]), float(np.median(ts[v])),
                                                              opt, vals[0], float(np.max(np.abs(out[v] - out[vals[0]])))), flush=True)
