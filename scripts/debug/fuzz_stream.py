#!/usr/bin/env python3
"""Randomised check of the serving stream (sr_stream_*: double-buffered H2D, optional hipGraph replay) against the synchronous
fused call on the same windows: random rates, window counts, model sets (independent, UBM + MAP, with collapsed components
-> hybrid form), other API traffic between ticks.  `fuzz_stream.py [cases] [seed]`"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from speaker_recognition_amd import _lib, synth
from speaker_recognition_amd.core import Batch, MfccExtractor, ModelSet, ServingStream
from speaker_recognition_amd.pygmm import GMM
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
fails = 0
for c in range(cases):
    fs = int(rng.choice([8000, 16000]))
    nwin = int(rng.integers(1, 9))
    K, S = int(rng.choice([16, 32, 64, 256])), int(rng.integers(1, 25))
    kind = str(rng.choice(["indep", "shared", "spoiled"]))
    if kind == "indep":
        models = [synth.synth_gmm(K, 13, 60 + s) for s in range(S)]
    else:
        ubm = synth.synth_gmm(K, 13, 99)
        if kind == "spoiled" and K >= 32:
            w, mu, sg = (a.copy() for a in ubm)
            sg[3] = 0.04
            mu[3] = mu.mean(0) + 2.0
            ubm = (w, mu, sg)
        S = max(S, 13)
        models = [ubm] + [synth.synth_map_speaker(ubm, 70 + s) for s in range(S - 1)]
    ms = ModelSet([GMM.from_arrays(*m) for m in models])
    ex = MfccExtractor(fs)
    audio = synth.synth_speech(int(rng.integers(20)), 14.0, fs)
    hop = int(0.3 * fs)
    n_ticks = 5
    ticks = [np.stack([audio[(t * nwin + j) * hop:(t * nwin + j) * hop + fs] for j in range(nwin)]) for t in range(n_ticks)]
    want = [ex.predict_batch(ms, Batch.from_pcm(list(tk)), nd=0) for tk in ticks]
    msg = ""
    for graph in (False, True):
        st = ServingStream(ex, ms, nwin, fs, graph=graph)
        got = []
        st.submit(ticks[0])
        for t in range(1, n_ticks):
            st.submit(ticks[t])
            if t == 2:
                ex.predict_batch(ms, Batch.from_pcm([audio[:2 * fs]]), nd=0)       # other traffic: workspaces move
            got.append(st.collect())
        got.append(st.collect())
        for t in range(n_ticks):
            if not (np.array_equal(got[t][0], want[t][0]) and np.array_equal(got[t][1], want[t][1])):
                msg += " [graph=%d tick %d differs: max %.2e]" % (graph, t, float(np.max(np.abs(got[t][0] - want[t][0]))))
        del st
    fails += bool(msg)
    print("case %2d fs %5d windows %d K %3d S %2d %-7s [%s]: %s" % (c, fs, nwin, K, len(models), kind, _lib.last_score_kernel()[:30], msg or "ok"))
print("cases with findings:", fails)
