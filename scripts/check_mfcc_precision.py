#!/usr/bin/env python3
"""MFCC precision modes against the float64 oracle on SURVEY.md 8d's voices, and their kernel times.
`check_mfcc_precision.py [speakers...]`: per speaker and mode (2 = float64 spectrum, 0 = fp32), max / mean |d| after CMVN
(13 statics) and over the 39 dims with both deltas; then the MFCC kernel's time on the configs[1] audio (1 M frames)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from oracle import mfcc_oracle as mo  # noqa: E402
from speaker_recognition_amd import _lib, synth  # noqa: E402
from speaker_recognition_amd.core import Batch, MfccExtractor  # noqa: E402

speakers = [int(a) for a in sys.argv[1:]] or [0, 7, 50, 99, 150]
fs = bench.FS
for generic in (0, 1):
    _lib.set_option("mfcc_generic", generic)
    for mode in (2, 0):
        _lib.set_option("mfcc_precision", mode)
        ex = MfccExtractor(fs, **bench.MFCC_KW)
        for s in speakers:
            pcm = synth.synth_speech(s % 200, 10.04, fs, seed=bench.AUDIO_SEED + s)
            ref = mo.extract(fs, pcm, diff=True, nd=2, **bench.MFCC_KW)
            got = ex.extract_batch(Batch.from_pcm([pcm]), nd=2).download()
            d = np.abs(got - ref)
            print("generic %d precision %d speaker %3d: statics max %.2e mean %.2e | 39 dims max %.2e mean %.2e | finite %s"
                  % (generic, mode, s, d[:, :13].max(), d[:, :13].mean(), d.max(), d.mean(), bool(np.all(np.isfinite(got)))), flush=True)
_lib.set_option("mfcc_generic", 0)
# other shapes through the float64 kernels (8 kHz preset, FFT 512 -> generic float64)
_lib.set_option("mfcc_precision", 2)
for fs2, kw in ((8000, {}), (16000, {}), (16000, dict(win_length_ms=25, win_shift_ms=10, FFT_SIZE=512)), (44100, dict(win_length_ms=25, win_shift_ms=10))):
    pcm = synth.synth_speech(50, 2.0, fs2)
    ref = mo.extract(fs2, pcm, diff=True, nd=2, **kw)
    got = MfccExtractor(fs2, **kw).extract_batch(Batch.from_pcm([pcm, pcm.astype(np.float32)]), nd=2).download()
    T = ref.shape[0]
    print("fs %d %s: int16 max %.2e, float32 PCM max %.2e" % (fs2, kw, np.abs(got[:T] - ref).max(), np.abs(got[T:] - ref).max()), flush=True)

clips, _ = bench.build_workload(0, 1000, 1000)
pcm = Batch.from_pcm(clips)
_lib.profile_enable(True)
for mode in (2, 0, 2):
    _lib.set_option("mfcc_precision", mode)
    ex = MfccExtractor(fs, **bench.MFCC_KW)
    ts = []
    for r in range(12):
        _lib.profile_reset()
        ex.extract_batch(pcm, nd=2)
        _lib.synchronize()
        ts.append(_lib.profile_get(_lib.T_MFCC)[0])
    print("precision %d: mfcc kernel on 1.002 M frames: first %.3f ms, median %.3f ms, min %.3f ms" % (mode, ts[0], float(np.median(ts[4:])), min(ts)), flush=True)
