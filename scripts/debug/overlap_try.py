import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from speaker_recognition_amd import _lib, synth
from speaker_recognition_amd.core import Batch, MfccExtractor, ModelSet
from speaker_recognition_amd.pygmm import GMM
ex = MfccExtractor(bench.FS, **bench.MFCC_KW)
n_samples = (1000 + 1) * ex.FRAME_SHIFT + ex.FRAME_LEN
base = bench.base_clips(100, n_samples)
cat, off = bench.make_pcm(base, 1000, 0)
pcm = Batch.from_pcm((cat, off))
ms = ModelSet([GMM.from_arrays(*synth.synth_gmm(64, 39, 7 + s)) for s in range(100)])
step = lambda: ex.predict_batch(ms, pcm, nd=2)
for wpb in (12, 4):
    for chunks in (1, 2, 4, 8):
        _lib.set_option("mfcc_waves_per_block", wpb); _lib.set_option("predict_chunks", chunks)
        step(); step(); _lib.synchronize()
        t0 = time.perf_counter()
        for _ in range(20): step()
        _lib.synchronize()
        print("wpb", wpb, "chunks", chunks, "ms/step %.3f" % ((time.perf_counter() - t0) / 20 * 1e3))
