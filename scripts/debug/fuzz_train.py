#!/usr/bin/env python3
"""Randomised check of the training engine against the oracle: one EM iteration and one MAP iteration from a shared start,
random K / D (wide rows included) / frame counts, far-from-origin feature spaces, and the legacy double** scorers against
the fused call.  `fuzz_train.py [cases] [seed]`"""
import ctypes as C
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
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
fails = 0
for c in range(cases):
    K, D = int(rng.integers(1, 40)), int(rng.choice([1, 5, 13, 26, 39, 48, 64, 70, 84, 128]))
    if c % 3 == 2:          # (round 4) whole 32-mixture tiles: the models the E-step's matrix-core responsibilities take (em_stats_split_kernel)
        K, D = int(rng.choice([32, 61, 64, 96, 125, 160, 256])), int(rng.choice([5, 13, 20, 26, 33, 39]))
    n = int(rng.choice([50, 300, 2000, 9000]))
    shift, scale = float(rng.choice([0.0, 0.0, -20.0])), float(rng.choice([1.0, 1.0, 0.1]))
    w, mu, sg = synth.synth_gmm(K, D, int(rng.integers(1 << 30)))
    true = (w, mu * scale + shift, sg * scale)
    X = synth.draw_frames(true, n, int(rng.integers(1 << 30)))
    start = go.GMMParams(np.full(K, 1.0 / K), true[1] + 0.2 * scale * rng.standard_normal(true[1].shape), np.full_like(true[2], 0.9 * scale))
    want = go.em_iteration(start, X.astype(np.float64))
    g = GMM.from_arrays(start.weights, start.mean, start.sigma)
    g.nr_iteration, g.init_with_kmeans = 1, -1
    g.fit(X)
    w1, mu1, sg1 = g.params()
    e_w, e_mu, e_sg = np.max(np.abs(w1 - want.weights)), np.max(np.abs(mu1 - want.mean)) / scale, np.max(np.abs(sg1 - want.sigma) / want.sigma)
    ubm = GMM.from_arrays(start.weights, start.mean, start.sigma)
    m = min(n, 300)
    want_map = go.em_iteration(start, X[:m].astype(np.float64), map_relevance=16.0, ubm=start)
    spk = GMM(K, nr_iteration=1)
    spk.fit(X[:m], ubm=ubm)
    e_map = np.max(np.abs(spk.params()[1] - want_map.mean)) / scale
    # legacy double** scorers (pygmm.hh:36-37) vs the fused call
    Xd = np.ascontiguousarray(X[:m], dtype=np.float64)
    ll = g.score(Xd)
    fused = ModelSet([g]).score(Batch.from_features([X[:m]]), frame_ll=True)[2][0]
    e_leg = float(np.max(np.abs(ll - fused)))
    ok = e_w < 2e-5 and e_mu < 3e-4 and e_sg < 2e-3 and e_map < 3e-4 and e_leg == 0.0
    fails += not ok
    print("case %2d K %2d D %3d n %4d shift %5g scale %4g: weights %.1e means %.1e sigmas %.1e MAP means %.1e legacy-vs-fused %.1e engine %d %s" % (
        c, K, D, n, shift, scale, e_w, e_mu, e_sg, e_map, e_leg, _lib.last_em_stats_engine(), "ok" if ok else "!!"))
print("cases with findings:", fails)
