#!/usr/bin/env python3
"""EM training time on the shape of the reference's only published benchmark
(doc/Final-Report-Complete/result.tex:41-49, img/time-comp.pdf): 10 EM iterations, 256 mixtures,
13-dim MFCC, 512 k frames -- reference C++ ~475 s (1 thread) / ~125 s (8) / ~70 s (16); scikit-learn
~2400 s (hardware unstated).  Synthetic frames drawn from a 256-mixture model."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from speaker_recognition_amd import _lib, synth  # noqa: E402
from speaker_recognition_amd.pygmm import GMM  # noqa: E402


def main():
    n, K, D, iters = (int(os.environ.get("EM_N", 512000)), int(os.environ.get("EM_K", 256)),
                      int(os.environ.get("EM_D", 13)), int(os.environ.get("EM_ITERS", 10)))
    true = synth.synth_gmm(K, D, 5)
    X = synth.draw_frames(true, n, 11)
    g = GMM(K, nr_iteration=iters, threshold=-1.0, seed=3)      # threshold < 0: never stop early
    g.fit(X[:2000])                                             # warm-up (context, code objects)
    _lib.profile_enable(True)
    _lib.profile_reset()
    g = GMM(K, nr_iteration=iters, threshold=-1.0, seed=3)
    t0 = time.perf_counter()
    it = g.fit(X)
    dt = time.perf_counter() - t0
    ms_e, n_e = _lib.profile_get(_lib.T_ESTEP)
    ms_s, n_s = _lib.profile_get(_lib.T_SCORE)
    ll = g.score_all(X[:50000]) / 50000
    ll_true = GMM.from_arrays(*true).score_all(X[:50000]) / 50000
    print(json.dumps({"what": "EM training, %d iterations, %d mixtures, %d-dim, %d frames" % (it, K, D, n),
                      "seconds": dt, "seconds_per_iteration": dt / it,
                      "estep_stats_kernel_ms_total": ms_e, "score_kernel_ms_total": ms_s,
                      "mean_ll_after": ll, "mean_ll_generating_model": ll_true,
                      "reference_published_seconds": {"c++ 1 thread": 475, "c++ 8 threads": 125,
                                                      "c++ 16 threads": 70, "scikit-learn": 2400},
                      "vs_reference_16_threads": 70.0 / dt}))


if __name__ == "__main__":
    main()
