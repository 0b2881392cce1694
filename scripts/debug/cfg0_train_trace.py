"""Where configs[0]'s training time goes: one speaker's fit (2996 x 13 MFCC frames of a 30 s synthetic voice, 16 mixtures, the
interface's defaults) with the phase times of every EM iteration (verbosity 2), then the wall time of the fit with the trace off, and of
the Python around it (list of rows -> array) as ModelInterface.train does it: cfg0_train_trace.py [K]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from speaker_recognition_amd import synth
from speaker_recognition_amd.feature import mix_feature
from speaker_recognition_amd.pygmm import GMM
K = int(sys.argv[1]) if len(sys.argv) > 1 else 16
kw = dict(win_length_ms=25, win_shift_ms=10)
feats = [mix_feature((16000, synth.synth_speech(sp, 30.0, 16000, seed=1000 + sp)), lpc=False, **kw) for sp in (0, 3, 6)]
print("features", feats[0].shape, feats[0].dtype)
g = GMM(K, seed=1); g.fit(feats[0])                       # warm: code objects, workspaces
g = GMM(K, seed=1, verbosity=2)
t0 = time.perf_counter(); g.fit(feats[1]); t1 = time.perf_counter()
print("traced fit: %.2f ms" % ((t1 - t0) * 1e3))
for rep in range(3):
    g = GMM(K, seed=1)
    t0 = time.perf_counter(); g.fit(feats[2]); t1 = time.perf_counter()
    print("fit: %.2f ms" % ((t1 - t0) * 1e3))
rows = list(feats[2])
t0 = time.perf_counter(); a = np.asarray(rows); t1 = time.perf_counter()
print("np.asarray(list of %d rows): %.2f ms" % (len(rows), (t1 - t0) * 1e3))
