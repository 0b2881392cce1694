#!/usr/bin/env python3
"""Randomised check of the two third-party-derived front ends against their numpy restatements (both parity unpinned, the
restatements are the bar): LPC-n columns (talkbox's algorithm) over random rates / windows / orders / silent stretches, and
the LTSD measure over random rates, orders and ragged signals.  `fuzz_frontends.py [cases] [seed]`"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import lpc_oracle as lpo, ltsd_oracle as lto
from speaker_recognition_amd import synth
from speaker_recognition_amd.feature import LPC
from speaker_recognition_amd.filters import ltsd as L
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 15
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
fails = 0
for c in range(cases):
    fs = int(rng.choice([8000, 16000, 22050]))
    kw = dict(win_length_ms=float(rng.choice([20, 25, 32])), win_shift_ms=float(rng.choice([10, 16])))
    pcm = synth.synth_speech(int(rng.integers(30)), float(rng.choice([0.4, 1.1, 2.0])), fs)
    if rng.random() < 0.5:
        a = int(rng.integers(0, len(pcm) // 2))
        pcm[a:a + int(rng.integers(100, 3000))] = 0
    msg = ""
    try:
        got = LPC.extract(fs, pcm, **kw)
        ref = lpo.extract(fs, pcm, **kw)
        if got.shape != ref.shape:
            msg += " [lpc shape %s vs %s]" % (got.shape, ref.shape)
        else:
            e = float(np.max(np.abs(got - ref) / np.maximum(1.0, np.abs(ref)))) if got.size else 0.0
            if e > 5e-6:
                msg += " [lpc err %.1e]" % e
    except Exception as ex:
        msg += " [lpc raised %s]" % str(ex)[:60]
    if fs != 22050:
        N = lto.window_size(fs)
        order = int(rng.choice([3, 5, 7]))
        noise = np.clip(rng.normal(0, 120, 2 * fs), -32768, 32767).astype(np.int16)
        na = lto.noise_spectrum(noise, N)
        sig = np.clip(pcm.astype(np.float64) + rng.normal(0, 120, len(pcm)), -32768, 32767).astype(np.int16)
        sigs = [sig, sig[:int(rng.integers(1, 3 * N))], sig[::-1].copy()]
        got = L.ltsd_values(sigs, na[:N // 2 + 1].astype(np.float32), N, order)
        for s_, g in zip(sigs, got):
            want = lto.ltsd(s_, na, N, order)
            if g.shape != want.shape:
                msg += " [ltsd shape %s vs %s]" % (g.shape, want.shape)
            elif len(want) and np.max(np.abs(g - want)) > 3e-3:
                msg += " [ltsd err %.1e dB]" % float(np.max(np.abs(g - want)))
    fails += bool(msg)
    print("case %2d fs %5d %s: %s" % (c, fs, kw, msg or "ok"))
print("cases with findings:", fails)
