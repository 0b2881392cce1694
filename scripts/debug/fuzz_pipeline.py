#!/usr/bin/env python3
"""Randomised end-to-end check: PCM -> MFCC (+ deltas) -> every model -> per-utterance sums and argmax on the device (the fused
sr_predict_pcm_batch call, the multi-slot path and the serving stream) against the CPU restatements run end to end in float64
(oracle MFCC -> oracle GMM).  `fuzz_pipeline.py [cases] [seed]`"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import gmm_oracle as go, mfcc_oracle as mo  # noqa: E402
from speaker_recognition_amd import _lib, synth  # noqa: E402
from speaker_recognition_amd.core import Batch, MfccExtractor, ModelSet, MultiPredictor  # noqa: E402
from speaker_recognition_amd.pygmm import GMM  # noqa: E402

go.build(ref=False)
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 15
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
fails = 0
for c in range(cases):
    fs = int(rng.choice([8000, 16000]))
    kw = dict(win_length_ms=float(rng.choice([25, 32])), win_shift_ms=float(rng.choice([10, 16])), FFT_SIZE=int(rng.choice([512, 1024, 2048])))
    if int(kw["win_length_ms"] / 1000 * fs) > kw["FFT_SIZE"]:
        kw["FFT_SIZE"] = 2048
    nd = int(rng.integers(0, 3))
    D = 13 * (nd + 1)
    S, K = int(rng.integers(2, 30)), int(rng.choice([8, 16, 32, 64]))
    shared = S >= 13 and rng.random() < 0.5
    # models trained on the device from the speakers' own audio (so that the scores are not all hopeless)
    ex = MfccExtractor(fs, **kw)
    spk_audio = [synth.synth_speech(s, 3.0, fs, seed=100 + s) for s in range(S)]
    feats = ex.extract_batch(Batch.from_pcm(spk_audio), nd=nd)
    X, off = feats.download(), feats.offsets()
    if shared:
        ubm = GMM(K, nr_iteration=6, seed=3)
        ubm.fit(X[::3])
        gm = [ubm]
        for s in range(S - 1):
            g = GMM(K, nr_iteration=2)
            g.fit(X[off[s]:off[s + 1]], ubm=ubm)
            gm.append(g)
    else:
        gm = []
        for s in range(S):
            g = GMM(K, nr_iteration=6, seed=5 + s)
            g.fit(X[off[s]:off[s + 1]])
            gm.append(g)
    # through the 6-digit text format, as a model file would
    gm = [GMM.loads(g.dumps()) for g in gm]
    ms = ModelSet(gm)
    clips = [synth.synth_speech(int(rng.integers(S)), float(rng.choice([0.2, 0.8, 1.5, 2.5])), fs, seed=int(rng.integers(1 << 30))) for _ in range(int(rng.integers(1, 7)))]
    sums, arg = ex.predict_batch(ms, Batch.from_pcm(clips), nd=nd)
    # the CPU restatements end to end
    params = [go.GMMParams(*g.params()) for g in gm]
    want = np.zeros((len(clips), S))
    warg = np.full(len(clips), -1)
    for u, p in enumerate(clips):
        x = p.astype(np.float64)
        if len(x) <= 5 * ex.FRAME_LEN:
            continue
        f = mo.extract(fs, x, diff=nd > 0, nd=max(nd, 1), **kw) if nd else mo.extract(fs, x, **kw)
        want[u] = [go.score_all(q, f) for q in params]
        warg[u] = int(np.argmax(want[u]))
    msg = ""
    for u in range(len(clips)):
        if warg[u] < 0:
            if arg[u] != -1:
                msg += " [utt %d: too short, device says %d]" % (u, arg[u])
            continue
        rel = float(np.max(np.abs(sums[u] - want[u]) / np.maximum(1.0, np.abs(want[u]))))
        top2 = np.sort(want[u])[-2:] if S > 1 else [0, 1]
        margin = (top2[1] - top2[0]) / max(1.0, abs(top2[1]))
        if arg[u] != warg[u] and margin > 1e-4:
            msg += " [utt %d argmax %d vs %d, margin %.1e]" % (u, arg[u], warg[u], margin)
        if rel > 2e-3:
            msg += " [utt %d sums rel %.1e]" % (u, rel)
    # multi-slot path and the plain split call agree with the fused one bit for bit
    mp = MultiPredictor(gm, fs, n_slots=2, **kw)
    s2, a2 = mp.predict(clips, nd=nd)
    if not (np.array_equal(s2, sums) and np.array_equal(a2, arg)):
        msg += " [multi-slot differs]"
    fb = ex.extract_batch(Batch.from_pcm(clips), nd=nd)
    s3, a3 = ms.score(fb)
    if not (np.array_equal(s3, sums) and np.array_equal(a3, arg)):
        msg += " [two-call path differs]"
    fails += bool(msg)
    print("case %2d fs %5d %s nd %d S %2d K %2d %s clips %d [%s]: %s" % (c, fs, kw, nd, S, K, "UBM+MAP" if shared else "indep  ", len(clips), _lib.last_score_kernel()[:32], msg or "ok"))
print("cases with findings:", fails)
