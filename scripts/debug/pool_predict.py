#!/usr/bin/env python3
"""The reference's own driver pattern (src/test/test-nperson.py:126-146): models built in the parent AFTER it used the GPU, then a
multiprocessing.Pool of forked workers calling gmmset.predict_one per utterance -- every call served by the fork helper
(csrc/fork_proxy.cpp).  The logged run's shape (80 speakers x 32 mixtures x 34 dims, 311-frame fragments), N utterances:
pool_predict.py [N=400] [workers=8]"""
import multiprocessing
import os
import sys
import time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from speaker_recognition_amd import synth
from speaker_recognition_amd.gmmset import GMMSet
from speaker_recognition_amd.pygmm import GMM
N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
W = int(sys.argv[2]) if len(sys.argv) > 2 else 8
S, K, D, T = 80, 32, 34, 311
raw = [synth.synth_gmm(K, D, 300 + s) for s in range(S)]
gs = GMMSet(gmm_order=K)
for s, m in enumerate(raw):
    gs._append(s, GMM.from_arrays(*m))
utts = [synth.draw_frames(raw[u % S], T, 900 + u).astype(np.float64) for u in range(N)]
truth = [u % S for u in range(N)]
t0 = time.perf_counter(); direct = [gs.predict_one(x) for x in utts[:200]]; t_direct = (time.perf_counter() - t0) / 200
def task(x):
    return gs.predict_one(x)
pool = multiprocessing.get_context("fork").Pool(W)
t0 = time.perf_counter()
res = [pool.apply_async(task, (x,)) for x in utts]
pred = [r.get() for r in res]
dt = time.perf_counter() - t0
pool.close(); pool.join()
print("in-process loop: %.3f ms per utterance; forked Pool(%d): %d utterances in %.3f s = %.3f ms per utterance, %d correct, agrees with the in-process loop: %s"
      % (t_direct * 1e3, W, N, dt, dt / N * 1e3, sum(int(p == t) for p, t in zip(pred, truth)), pred[:200] == direct))
