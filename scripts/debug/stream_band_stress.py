#!/usr/bin/env python3
"""Stress of the serving stream with frames in the partial-product band (tests/test_gpu_pipeline.py::
test_band_frames_through_fused_pipelined_and_streaming_paths, which failed once in ~10 full-suite runs): N rounds of
submit / submit / collect / collect, plain and graph-captured, against the fused step; prints every mismatch."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from speaker_recognition_amd import _lib, synth
from speaker_recognition_amd.core import Batch, MfccExtractor, ModelSet, ServingStream
from speaker_recognition_amd.pygmm import GMM
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
fs, n_win, win = 16000, 16, 32000
ex = MfccExtractor(fs)
pcm = np.stack([synth.synth_speech(3 + u, 2.0, fs)[:win] for u in range(n_win)])
D, K = 13, 32
rng = np.random.default_rng(2)
models = []
for s in range(3):
    mean = np.zeros((K, D)); mean[:, s] = 1.84 + 0.01 * rng.standard_normal(K)
    sigma = np.full((K, D), 3.0); sigma[:, s] = 0.05
    r6 = np.vectorize(lambda v: float("%g" % v))
    models.append((np.full(K, 1.0 / K), r6(mean), sigma))
ms = ModelSet([GMM.from_arrays(*m) for m in models])
batch = Batch.from_pcm(list(pcm))
s_fused, a_fused = ex.predict_batch(ms, batch, nd=0)
bad = 0
for r in range(rounds):
    s2, a2 = ex.predict_batch(ms, batch, nd=0)
    if not np.array_equal(s2, s_fused):
        bad += 1; print("round", r, "fused differs from itself: max abs", np.max(np.abs(s2 - s_fused)), flush=True)
    for graph in (False, True):
        st = ServingStream(ex, ms, n_win, win, nd=0, graph=graph)
        st.submit(pcm); st.submit(pcm)
        for i in range(2):
            s_st, a_st, _ = st.collect()
            if not (np.array_equal(s_st, s_fused) and np.array_equal(a_st, a_fused)):
                bad += 1
                d = np.abs(s_st - s_fused)
                u, m = np.unravel_index(np.argmax(d), d.shape)
                print("round", r, "graph", graph, "collect", i, "max abs diff", d.max(), "at utt", u, "model", m, "got", s_st[u, m], "want", s_fused[u, m],
                      "n differing", int((d > 0).sum()), flush=True)
        del st
print("rounds", rounds, "mismatches", bad, "flush calls", _lib.flush_stats())
