#!/usr/bin/env python3
"""Randomised check of the shared-sigma engines (the three workgroup shapes of engine 6, its exception kernel, the hybrid
form) against the float64 oracle: random K, D, S, ragged utterances, outliers, clamp on / off.  `fuzz_shared.py [cases] [seed]`"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import gmm_oracle as go  # noqa: E402
from speaker_recognition_amd import _lib, synth  # noqa: E402
from speaker_recognition_amd.core import Batch, ModelSet  # noqa: E402
from speaker_recognition_amd.pygmm import GMM  # noqa: E402

go.build(ref=False)
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
worst = 0.0
for c in range(cases):
    K, D, S = int(rng.integers(20, 200)), int(rng.integers(5, 49)), int(rng.integers(12, 36))
    ubm = synth.synth_gmm(K, D, int(rng.integers(1 << 30)))
    if rng.random() < 0.3:                      # a mixture of weight 0, shared by every model of the set
        w, mu, sg = ubm
        w = w.copy()
        w[int(rng.integers(K))] = 0.0
        ubm = (w, mu, sg)
    spoil = rng.random() < 0.3
    if spoil:
        w, mu, sg = (a.copy() for a in ubm)
        for k in rng.choice(K, size=2, replace=False):
            sg[k] = 0.04
            mu[k] = np.round(mu.mean(0) + 2.0 * rng.choice([-1.0, 1.0], size=D), 4)
        ubm = (w, mu, sg)
    models = [ubm] + [synth.synth_map_speaker(ubm, int(rng.integers(1 << 30))) for _ in range(S - 1)]
    lens = [int(v) for v in rng.choice([0, 1, 5, 31, 32, 33, 100, 257, 700], size=int(rng.integers(1, 9)))]
    utts = [synth.draw_frames(models[int(rng.integers(S))], n, int(rng.integers(1 << 30)), outlier_frac=float(rng.choice([0.0, 0.02]))) for n in lens]
    if sum(lens) == 0:
        continue
    X = np.concatenate(utts).astype(np.float64)
    compat = bool(rng.integers(2))
    want = np.stack([go.score_batch(go.GMMParams(*m), X, go.MODE_LOGSUMEXP, clamp_compat=compat) for m in models])
    ms = ModelSet([GMM.from_arrays(*m) for m in models])
    bad = False
    for shape, force in ((1, 0), (2, 0), (3, 0), (4, 0), (0, 1), (2, 1), (3, 1), (4, 1)):
        _lib.set_option("score_h2s_shape", shape)
        _lib.set_option("score_h2s_force_exc", force)
        sums, arg, fll = ms.score(Batch.from_features(utts), frame_ll=True, clamp_compat=compat)
        err = float(np.max(np.abs(fll - want) / np.maximum(1.0, np.abs(want))))
        worst = max(worst, err)
        name = _lib.last_score_kernel()
        off = np.concatenate([[0], np.cumsum(lens)])
        for u, n in enumerate(lens):
            if n:
                w_ = want[:, off[u]:off[u + 1]].sum(axis=1)
                dev = float(np.max(np.abs(sums[u] - w_) / np.maximum(1.0, np.abs(w_))))
                if dev >= 3e-5:
                    j = int(np.argmax(np.abs(sums[u] - w_)))
                    fe = np.abs(fll[j, off[u]:off[u + 1]] - want[j, off[u]:off[u + 1]])
                    print("  !! case", c, "shape", shape, "force", force, "utt", u, "n", n, "model", j, "sum dev", dev, "got", sums[u][j], "want", w_[j],
                          "worst frame err", fe.max(), "at", int(fe.argmax()), "K D S", K, D, S, "compat", compat, name[:50])
                    bad = True
        if err >= 1e-4:
            print("  !! per-frame err", err, (c, K, D, S, lens, compat, shape, force, name[:50]))
    print("case %2d K %3d D %2d S %2d frames %4d %s clamp %d: ok  [%s]" % (c, K, D, S, sum(lens), "spoiled" if spoil else "       ", compat, name[:60]))
print("worst relative per-frame error %.2e" % worst)
