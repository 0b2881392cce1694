#!/usr/bin/env python3
"""One short utterance per call against the 201 x 512 x 39 set (bench.py's serving_small_batch, utterances_1), N calls --
for `rocprofv3 --kernel-trace`: what the 0.24 ms of a decision are made of.  serving_one.py [N=60] [U=1]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from speaker_recognition_amd import _lib, synth
from speaker_recognition_amd.core import Batch, MfccExtractor, ModelSet
from speaker_recognition_amd.pygmm import GMM
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
U = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ex = MfccExtractor(bench.FS, **bench.MFCC_KW)
T = 300
n_samples = (T + bench.ND - 1) * ex.FRAME_SHIFT + ex.FRAME_LEN
ubm = synth.synth_gmm(512, 39, 99)
ms = ModelSet([GMM.from_arrays(*m) for m in [ubm] + [synth.synth_map_speaker(ubm, 500 + s) for s in range(200)]])
clips = [synth.synth_speech(u, 3.1, bench.FS, seed=7 + u)[:n_samples] for u in range(U)]
cat = np.concatenate(clips)
batch = Batch.from_pcm(clips)
lat = []
for i in range(N):
    t0 = time.perf_counter()
    batch.update_pcm(cat)
    sums, arg = ex.predict_batch(ms, batch, nd=bench.ND)
    lat.append((time.perf_counter() - t0) * 1e3)
print("p50 %.4f ms, min %.4f ms over the last %d calls" % (float(np.median(lat[N // 2:])), float(np.min(lat[N // 2:])), N - N // 2))
