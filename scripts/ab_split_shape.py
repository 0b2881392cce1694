#!/usr/bin/env python3
"""Workgroup shapes of the generic split-fp16 engine side by side inside ONE GPU-box call:
    ab_split_shape.py WORKLOAD [SHAPE ...]        (shapes: 1 = 4-wave kernel, 8 / 12 / 16 = waves of gmm_score_splitp_kernel; default all)
    ab_split_shape.py WORKLOAD --groups G ...     (force the number of model groups)
WORKLOAD as scripts/ab_option.py: point256 (1 x 256 x 39, 2 M frames) | cfg1 (100 x 64 x 39, 1 M frames) | small13 | ubm512 | serve (201 x 64 x 39, 8 x 300 frames) | stream1024 (20 x 256 x 13, 1024 x 61 frames)
Prints the scoring kernel's HIP-event time per shape (alternating, median of 5), G frames/s, TB/s of feature reads, the algorithmic
TFLOP/s (S K (4 D + 6) per frame) and the largest relative difference of the sums against the first shape."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from speaker_recognition_amd import _lib, synth  # noqa: E402
from speaker_recognition_amd.core import Batch, ModelSet  # noqa: E402
from speaker_recognition_amd.pygmm import GMM  # noqa: E402

SHAPES = {"point256": (1, 256, 39, 2000, 1000), "cfg1": (100, 64, 39, 1000, 1000), "small13": (10, 32, 13, 1000, 1000), "ubm512": (1, 512, 39, 1000, 1000),
          "serve": (201, 64, 39, 8, 300), "cfg1_k256": (100, 256, 39, 250, 1000),
          "stream1024": (20, 256, 13, 1024, 61), "cfg1_d26": (100, 64, 26, 1000, 1000), "cfg1_d20": (100, 64, 20, 1000, 1000), "cfg1_d13": (100, 64, 13, 1000, 1000)}     # configs[4]: 1024 one-second windows per tick


def main():
    args = sys.argv[1:]
    wl = args.pop(0)
    groups = 0
    if args and args[0] == "--groups":
        groups = int(args[1])
        args = args[2:]
    shapes = [int(v) for v in args] or [1, 8, 12, 16]
    S, K, D, U, T = SHAPES[wl]
    models = [synth.synth_gmm(K, D, 77 + s) for s in range(S)]
    ms = ModelSet([GMM.from_arrays(*m) for m in models])
    base = [synth.draw_frames(models[u % S], T, 100 + u, outlier_frac=0.001) for u in range(min(U, 100))]
    feats = Batch.from_features([base[u % len(base)] for u in range(U)])
    _lib.set_option("score_engine", 5)
    _lib.set_option("score_model_groups", groups)
    _lib.profile_enable(True)
    res, ts, names = {}, {v: [] for v in shapes}, {}
    for r in range(6):
        for v in shapes:
            _lib.set_option("score_split_shape", v)
            _lib.profile_reset()
            sums, arg = ms.score(feats)
            t, c = _lib.profile_get(_lib.T_SCORE)
            if r > 0:
                ts[v].append(t)
            res[v] = sums
            names[v] = _lib.last_score_kernel().split(" ")[0]
    n = U * T
    for v in shapes:
        t = float(np.median(ts[v]))
        d = float(np.max(np.abs(res[v] - res[shapes[0]]) / np.maximum(1.0, np.abs(res[shapes[0]]))))
        print("%s groups=%d shape=%2d: %8.4f ms  %.2f G frames/s  %.2f TB/s  %.0f TFLOP/s algorithmic  (%s)  max rel sum diff %.1e  %s" % (
            wl, groups, v, t, n / t / 1e6, n * 4 * D / t / 1e9, n * S * K * (4 * D + 6) / t / 1e9, " ".join("%.3f" % x for x in ts[v]), d, names[v]), flush=True)


if __name__ == "__main__":
    main()
