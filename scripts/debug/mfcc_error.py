"""Measured error of the device MFCC chain against the float64 goldens of the reference's MFCC.py (tests/golden/mfcc_golden.npz)
and against the float64 restatement on the bench's audio: max and MEAN absolute difference after CMVN (SURVEY.md 8d's gate: max
<= 1e-3, mean <= 1e-5), per case and FFT kernel."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import mfcc_oracle as mo
from speaker_recognition_amd import _lib, synth
from speaker_recognition_amd.feature import MFCC
m = np.load(os.path.join(ROOT, "tests", "golden", "mfcc_golden.npz"))
for generic in (0, 1):
    _lib.set_option("mfcc_generic", generic)
    for c in m["cases"]:
        kw = eval(str(m[c + "_kw"]))
        fs, pcm = int(m[c + "_fs"]), m[c + "_pcm"]
        feat = MFCC.extract(fs, pcm, **kw)
        d = np.abs(feat - m[c + "_feat"])
        print("golden %-12s generic=%d: max %.2e mean %.2e  (%d frames, kw %s)" % (c, generic, d.max(), d.mean(), len(feat), kw))
_lib.set_option("mfcc_generic", 0)
kw = dict(win_length_ms=25, win_shift_ms=10)
for s in range(4):
    pcm = synth.synth_speech(s, 10.04, 16000, seed=2000 + s)
    feat = MFCC.extract(16000, pcm, diff=True, nd=2, **kw)
    ref = mo.extract(16000, pcm, diff=True, nd=2, **kw)
    d = np.abs(feat - ref)
    print("bench audio speaker %d (39 dims): max %.2e mean %.2e" % (s, d.max(), d.mean()))
