#!/usr/bin/env python3
"""A serving loop whose utterances DIFFER in length from call to call (2.5 .. 3.5 s, one per call) against the 201 x 512 x 39 set:
Batch.reset_pcm + predict_batch per decision, beside the fixed-length loop of serving_one.py.  serving_varlen.py [N=400]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from speaker_recognition_amd import synth
from speaker_recognition_amd.core import Batch, MfccExtractor, ModelSet
from speaker_recognition_amd.pygmm import GMM
N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
ex = MfccExtractor(bench.FS, **bench.MFCC_KW)
ubm = synth.synth_gmm(512, 39, 99)
ms = ModelSet([GMM.from_arrays(*m) for m in [ubm] + [synth.synth_map_speaker(ubm, 500 + s) for s in range(200)]])
clip = synth.synth_speech(3, 4.0, bench.FS, seed=7)
rng = np.random.default_rng(1)
lens = rng.integers(int(2.5 * bench.FS), int(3.5 * bench.FS), N)
batch = Batch.from_pcm([clip[:lens[0]]])
for mode in ("fixed length (update_pcm)", "varying length (reset_pcm)"):
    lat = []
    for i in range(N):
        x = clip[:lens[0]] if mode.startswith("fixed") else clip[:lens[i]]
        t0 = time.perf_counter()
        if mode.startswith("fixed"):
            batch.update_pcm(x)
        else:
            batch.reset_pcm([x])
        sums, arg = ex.predict_batch(ms, batch, nd=bench.ND)
        lat.append((time.perf_counter() - t0) * 1e3)
    print("%s: p50 %.4f ms, p99 %.4f ms over the last %d calls" % (mode, float(np.median(lat[N // 2:])), float(np.percentile(lat[N // 2:], 99)), N - N // 2))
# and the same decisions one by one from scratch must agree with the loop's last one
ref = ex.predict_batch(ms, Batch.from_pcm([clip[:lens[N - 1]]]), nd=bench.ND)
print("last decision equals a fresh batch's:", bool(np.array_equal(ref[0], sums) and np.array_equal(ref[1], arg)))
