#!/usr/bin/env python3
"""Which engine do TRAINED models get?  Enrol S synthetic speakers (EM on the device, the reference's defaults) from `secs`
seconds of audio each and report the set's conditioning, the engine the dispatcher takes, and the per-frame error of every
engine forced on it against the float64 oracle."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import gmm_oracle as go
from speaker_recognition_amd import _lib, synth
from speaker_recognition_amd.core import Batch, MfccExtractor, ModelSet
from speaker_recognition_amd.pygmm import GMM
go.build(ref=False)
fs = 16000
for secs, K, nd in ((30, 16, 0), (30, 32, 0), (30, 32, 2), (10, 32, 2), (60, 64, 2)):
    S = 10
    ex = MfccExtractor(fs, win_length_ms=25, win_shift_ms=10)
    audio = [synth.synth_speech(s, float(secs), fs, seed=1000 + s) for s in range(S)]
    fb = ex.extract_batch(Batch.from_pcm(audio), nd=nd)
    X, off = fb.download(), fb.offsets()
    gm = []
    for s in range(S):
        g = GMM(K, nr_iteration=200, seed=5 + s)
        g.fit(X[off[s]:off[s + 1]])
        gm.append(GMM.loads(g.dumps()))
    ms = ModelSet(gm)
    info = ms.info()
    test = ex.extract_batch(Batch.from_pcm([synth.synth_speech(s, 3.0, fs, seed=2000 + s) for s in range(S)]), nd=nd)
    Xt = test.download().astype(np.float64)
    want = np.stack([go.score_batch(go.GMMParams(*g.params()), Xt, go.MODE_LOGSUMEXP) for g in gm])
    line = "secs %2d K %2d dims %2d: amp %.0f hybrid %d sigma_min %.3f |" % (secs, K, 13 * (nd + 1), info["amp"], info["hybrid_vector_mixtures"], min(g.params()[2].min() for g in gm))
    for eng in (0, 1, 3, 5):
        _lib.set_option("score_engine", eng)
        try:
            _, arg, fll = ms.score(test, frame_ll=True)
            err = float(np.max(np.abs(fll - want) / np.maximum(1.0, np.abs(want))))
            line += " eng%d %.1e%s" % (eng, err, (" [" + _lib.last_score_kernel()[:22] + "]") if eng == 0 else "")
        except _lib.SRError as e:
            line += " eng%d refused" % eng
    _lib.set_option("score_engine", 0)
    print(line)
