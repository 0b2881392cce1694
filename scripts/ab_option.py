#!/usr/bin/env python3
"""A/B of a library option inside ONE GPU-box call on a chosen workload:
    ab_option.py WORKLOAD OPTION V0 V1
WORKLOAD: point256 (1 x 256 x 39, 2 M frames) | cfg1 (100 x 64 x 39, 1 M frames) | small13 (10 x 32 x 13, 1 M frames)
Prints the scoring kernel's HIP-event times under both settings (alternating) and the largest relative difference of the sums."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from speaker_recognition_amd import _lib, synth  # noqa: E402
from speaker_recognition_amd.core import Batch, ModelSet  # noqa: E402
from speaker_recognition_amd.pygmm import GMM  # noqa: E402

SHAPES = {"point256": (1, 256, 39, 2000), "cfg1": (100, 64, 39, 1000), "small13": (10, 32, 13, 1000), "ubm512": (1, 512, 39, 1000)}


def main():
    wl, opt, v0, v1 = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    S, K, D, U = SHAPES[wl]
    models = [synth.synth_gmm(K, D, 77 + s) for s in range(S)]
    ms = ModelSet([GMM.from_arrays(*m) for m in models])
    base = [synth.draw_frames(models[u % S], 1000, 100 + u, outlier_frac=0.001) for u in range(100)]
    feats = Batch.from_features([base[u % 100] for u in range(U)])
    _lib.profile_enable(True)
    res, ts, names = {}, {v0: [], v1: []}, {}
    for r in range(6):
        for v in (v0, v1):
            _lib.set_option(opt, v)
            _lib.profile_reset()
            sums, arg = ms.score(feats)
            t, c = _lib.profile_get(_lib.T_SCORE)
            if r > 0:
                ts[v].append(t)
            res[v] = sums
            names[v] = _lib.last_score_kernel().split(" ")[0]
    d = float(np.max(np.abs(res[v0] - res[v1]) / np.maximum(1.0, np.abs(res[v0]))))
    n = U * 1000
    for v in (v0, v1):
        t = float(np.median(ts[v]))
        print("%s %s=%d: %s ms (median %.4f: %.2f G frames/s = %.2f TB/s of feature reads) %s" % (
            wl, opt, v, " ".join("%.4f" % x for x in ts[v]), t, n / t / 1e6, n * 4 * D / t / 1e9, names[v]))
    print("ratio %.3f, max rel sum diff %.2e" % (np.median(ts[v1]) / np.median(ts[v0]), d), flush=True)


if __name__ == "__main__":
    main()
