#!/usr/bin/env python3
"""BASELINE configs[3], one rank's shard at full size: 1e8 frames / 8 GPUs = 12.5 M frames x 39 dims
against a 2048-mixture UBM + 1000 MAP-adapted speaker models (2048 mixtures each), features
resident.  Prints one JSON line: seconds per pass, frames/s, TFLOP/s, and the size-independent
checks (own speaker wins among the speakers; repeated utterances bit-identical; margin > 0)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from speaker_recognition_amd import _lib, synth  # noqa: E402
from speaker_recognition_amd.core import Batch, ModelSet  # noqa: E402
from speaker_recognition_amd.pygmm import GMM  # noqa: E402


def main():
    S = int(os.environ.get("CFG3_S", 1000))
    K = int(os.environ.get("CFG3_K", 2048))
    U = int(os.environ.get("CFG3_U", 12500))
    D, T = 39, 1000
    _lib.set_option("score_engine", int(os.environ.get("CFG3_ENGINE", 0)))   # before the set is packed
    _lib.set_option("score_h2s_force_exc", int(os.environ.get("CFG3_FORCE_EXC", 0)))
    _lib.set_option("score_h2s_shape", int(os.environ.get("CFG3_SHAPE", 0)))
    t0 = time.time()
    ubm = synth.synth_gmm(K, D, 99)
    w, mean, sigma = ubm
    spoil = int(os.environ.get("CFG3_SPOIL", 0))      # that many UBM mixtures collapsed (sigma 0.04) and far from the centre: an ill-conditioned set
    if spoil:
        mean, sigma = mean.copy(), sigma.copy()
        rng = np.random.default_rng(1)
        for k in rng.choice(K, size=spoil, replace=False):
            sigma[k] = 0.04
            mean[k] = mean.mean(0) + 2.0 * rng.choice([-1.0, 1.0], size=D)
        ubm = (w, mean, sigma)
    nk = w * 40.0 * K
    alpha = (nk / (nk + 16.0))[:, None]
    spk = []
    for s in range(S):
        rng = np.random.default_rng(500 + s)
        spk.append((w, mean + alpha * 0.3 * rng.standard_normal(mean.shape), sigma))
    models = [ubm] + spk
    ms = ModelSet([GMM.from_arrays(*m) for m in models])
    base = [synth.draw_frames(spk[s], T, 9000 + s) for s in range(min(S, U))]
    utts = [base[u % len(base)] for u in range(U)]
    feats = Batch.from_features(utts)
    t_setup = time.time() - t0
    _lib.profile_enable(True)
    times = []
    for r in range(int(os.environ.get("CFG3_ROUNDS", 2))):
        _lib.profile_reset()
        t1 = time.time()
        sums, arg = ms.score(feats)
        wall = time.time() - t1
        kms, _ = _lib.profile_get(_lib.T_SCORE)
        kref, _ = _lib.profile_get(_lib.T_SCORE_REF)
        times.append((wall, kms + kref, kref))
    n = feats.n_rows
    flops = float(n) * (S + 1) * K * (4 * D + 6)
    best = np.argmax(sums[:, 1:], axis=1)
    nb = len(base)
    checks = {
        "own_speaker_wins": bool(np.array_equal(best, np.arange(U) % nb)),
        "repeats_bit_identical": bool(all(np.array_equal(sums[r * nb:(r + 1) * nb], sums[:nb]) for r in range(1, U // nb))),
        "margin_positive": bool(np.all(sums[:, 1:].max(axis=1) > sums[:, 0])),
        "finite": bool(np.all(np.isfinite(sums))),
    }
    kms = min(t[1] for t in times)
    print(json.dumps({"workload": "configs[3] per-rank shard: %d frames x %d dims, UBM %d mixtures + %d MAP speakers" % (n, D, K, S),
                      "kernel": _lib.last_score_kernel(), "score_kernel_s": kms * 1e-3, "wall_s": min(t[0] for t in times),
                      "frames_per_s": n / (kms * 1e-3), "algorithmic_tflops": flops / (kms * 1e-3) / 1e12,
                      "ref_prepass_s": min(t[2] for t in times) * 1e-3, "setup_s": t_setup, "checks": checks}))


if __name__ == "__main__":
    main()
