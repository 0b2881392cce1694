#!/usr/bin/env python3
"""A/B sweep of the scoring-kernel variants (frames per lane x packed FMA x model groups) on
the cfg-1 shape, interleaved rounds in one process; prints kernel ms and TFLOP/s per variant
from the library's HIP-event timers, plus the parity of each variant against variant 0."""
import itertools
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from speaker_recognition_amd import _lib, synth  # noqa: E402
from speaker_recognition_amd.core import Batch, ModelSet  # noqa: E402
from speaker_recognition_amd.pygmm import GMM  # noqa: E402


def main():
    S = int(os.environ.get("TUNE_S", 100))
    K = int(os.environ.get("TUNE_K", 64))
    D = int(os.environ.get("TUNE_D", 39))
    U = int(os.environ.get("TUNE_U", 1000))
    T = int(os.environ.get("TUNE_T", 1000))
    rounds = int(os.environ.get("TUNE_ROUNDS", 3))
    models = [synth.synth_gmm(K, D, 7 + s) for s in range(S)]
    ms = ModelSet([GMM.from_arrays(*m) for m in models])
    rng = np.random.default_rng(0)
    base = [synth.draw_frames(models[s % S], T, 42 + s, outlier_frac=0.001) for s in range(min(U, 50))]
    utts = [base[u % len(base)] for u in range(U)]
    feats = Batch.from_features(utts)
    n = feats.n_rows
    flops = float(n) * S * K * (4 * D + 6)
    # (frames per lane | -FT for the fp32 matrix-core engine | -(10+FT) for the split-bf16 one, packed, groups)
    variants = [(4, 1, 0), (4, -1, 0), (2, 1, 0), (-1, 0, 0), (-2, 0, 0), (-3, 0, 0), (-11, 0, 0), (-12, 0, 0)]
    if os.environ.get('TUNE_FEW'):
        variants = [(4, 1, 0), (-1, 0, 0), (-2, 0, 0), (-11, 0, 0), (-12, 0, 0)]
    if D > 40:
        variants = [v for v in variants if v[0] <= 2 and v[0] >= -3 or v[0] == -11]
    ref = None
    res = {v: [] for v in variants}
    _lib.profile_enable(True)
    for r in range(rounds + 1):
        for v in variants:
            F, pk, G = v
            _lib.set_option("score_engine", 3 if F <= -10 else 2 if F < 0 else 1)
            _lib.set_option("score_mfma_ft", -F - 10 if F <= -10 else -F if F < 0 else 0)
            _lib.set_option("score_frames_per_lane", F if F > 0 else 0)
            _lib.set_option("score_packed", pk)
            _lib.set_option("score_model_groups", G)
            _lib.profile_reset()
            sums, arg = ms.score(feats)
            t, cnt = _lib.profile_get(_lib.T_SCORE)
            if r > 0:
                res[v].append(t)
            if ref is None:
                ref = sums.copy()
            elif r == 0:
                d = np.max(np.abs(sums - ref) / np.maximum(1, np.abs(ref)))
                print("variant", v, "max rel diff of sums vs first variant: %.3e" % d, flush=True)
    out = []
    for v in variants:
        t = np.array(res[v])
        out.append({"F": v[0], "packed": v[1], "groups": v[2], "ms_min": float(t.min()), "ms_med": float(np.median(t)),
                    "tflops_best": flops / (t.min() * 1e-3) / 1e12})
        print("F=%d pk=%d G=%d  min %.3f ms  med %.3f ms  -> %.1f TFLOP/s (%.1f%% of 157.3)" % (
            v[0], v[1], v[2], t.min(), np.median(t), out[-1]["tflops_best"], out[-1]["tflops_best"] / 1.573), flush=True)
    print(json.dumps({"shape": dict(S=S, K=K, D=D, U=U, T=T), "variants": out}))


if __name__ == "__main__":
    main()
