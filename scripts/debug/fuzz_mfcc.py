#!/usr/bin/env python3
"""Randomised check of the MFCC chain against the float64 oracle: sampling rates, window / shift, FFT sizes, filter and
cepstrum counts, pre-emphasis, ragged batches with too-short utterances, int16 and float32 PCM, delta orders, LPC columns.
`fuzz_mfcc.py [cases] [seed]`"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import mfcc_oracle as mo  # noqa: E402
from speaker_recognition_amd import _lib, synth  # noqa: E402
from speaker_recognition_amd.core import Batch, MfccExtractor  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
prec = int(sys.argv[3]) if len(sys.argv) > 3 else 2            # mfcc_precision: 2 float64 spectrum (default), 0 fp32
_lib.set_option("mfcc_precision", prec)
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
worst_raw = worst_feat = 0.0
fails = 0
for c in range(cases):
    fs = int(rng.choice([8000, 11025, 16000, 22050, 44100]))
    win = float(rng.choice([10, 20, 25, 32, 40]))
    shift = float(rng.choice([5, 10, 16, 20]))
    L = int(win / 1000 * fs)
    ffts = [f for f in (256, 512, 1024, 2048, 4096) if f >= L] or [4096]
    fft = int(rng.choice(ffts))
    nf = int(rng.choice([20, 26, 40, 50, 64]))
    nc = int(rng.integers(5, min(nf, 17)))
    pre = float(rng.choice([0.0, 0.9, 0.95, 0.97]))
    kw = dict(win_length_ms=win, win_shift_ms=shift, FFT_SIZE=fft, n_filters=nf, n_ceps=nc, pre_emphasis_coef=pre)
    secs = [float(v) for v in rng.choice([0.05, 0.3, 0.7, 1.1, 2.0], size=int(rng.integers(1, 5)))]
    pcm = [synth.synth_speech(int(rng.integers(50)), s, fs) for s in secs]
    nd = int(rng.integers(0, 3))
    as_float = bool(rng.integers(2))
    generic = int(rng.integers(2))
    _lib.set_option("mfcc_generic", generic)
    try:
        ex = MfccExtractor(fs, **kw)
    except Exception as e:
        print("case %d refused: %s %s" % (c, kw, str(e)[:60]))
        continue
    sigs = [p.astype(np.float32) * 0.5 if as_float else p for p in pcm]
    out = ex.extract_batch(Batch.from_pcm(sigs), nd=nd)
    X, off = out.download(), out.offsets()
    oe = mo.get_mfcc_extractor(fs, **kw)
    msg = ""
    for i, sg in enumerate(sigs):
        x = np.asarray(sg, dtype=np.float64)
        if len(x) <= 5 * ex.FRAME_LEN:
            if off[i + 1] != off[i]:
                msg += " [utt %d should be empty]" % i
            continue
        raw_ref = oe.raw_cepstra(x)
        raw = ex.extract(sg, cmvn=False)
        if raw.shape != raw_ref.shape:
            msg += " [raw shape %s vs %s]" % (raw.shape, raw_ref.shape)
            continue
        e_raw = float(np.max(np.abs(raw - raw_ref)) / max(1.0, np.abs(raw_ref).max()))
        ref = mo.extract(fs, x, diff=nd > 0, nd=max(nd, 1), **kw) if nd else mo.extract(fs, x, **kw)
        got = X[off[i]:off[i + 1]]
        if got.shape != ref.shape:
            msg += " [feat shape %s vs %s]" % (got.shape, ref.shape)
            continue
        e_feat = float(np.max(np.abs(got - ref)))
        worst_raw, worst_feat = max(worst_raw, e_raw), max(worst_feat, e_feat)
        if e_raw > (3e-6 if prec == 2 else 3e-4) or e_feat > (1e-4 if prec == 2 else 6e-3):
            msg += " [utt %d raw %.1e feat %.1e]" % (i, e_raw, e_feat)
    if msg:
        fails += 1
    print("case %2d fs %5d win %g/%g fft %4d filters %2d ceps %2d pre %.2f nd %d %s generic %d: %s" % (
        c, fs, win, shift, fft, nf, nc, pre, nd, "f32" if as_float else "i16", generic, msg or "ok"))
_lib.set_option("mfcc_generic", 0)
print("worst raw (relative to the largest cepstrum) %.2e, worst normalised feature %.2e, cases with findings: %d" % (worst_raw, worst_feat, fails))
