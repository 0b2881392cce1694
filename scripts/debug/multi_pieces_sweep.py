#!/usr/bin/env python3
"""sr_multi_predict_pcm on configs[1] from page-locked host PCM, one slot: ms per call by multi_pieces (0 = the shaped default), beside the
resident-PCM step and the bare host -> device copy of the same PCM."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from speaker_recognition_amd import _lib, synth
from speaker_recognition_amd.core import Batch, MfccExtractor, ModelSet, MultiPredictor
from speaker_recognition_amd.pygmm import GMM
ex = MfccExtractor(bench.FS, **bench.MFCC_KW)
n_samples = (bench.FRAMES_PER_UTT + bench.ND - 1) * ex.FRAME_SHIFT + ex.FRAME_LEN
base_c = bench.base_clips(bench.CFG1_MODELS, n_samples)
gm = [GMM.from_arrays(*synth.synth_gmm(bench.CFG1_MIX, bench.DIM, bench.MODEL_SEED + s)) for s in range(bench.CFG1_MODELS)]
cat, off = bench.make_pcm(base_c, bench.CFG1_UTTS, 0)
ms = ModelSet(gm)
pcm = Batch.from_pcm((cat, off))
def timed(f, n=10):
    f(); f()
    _lib.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    _lib.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
print("resident step      %.3f ms" % timed(lambda: ex.predict_batch(ms, pcm, nd=2)))
mp0 = MultiPredictor(gm, bench.FS, n_slots=1, **bench.MFCC_KW)
print("pageable, 1 slot   %.3f ms" % timed(lambda: mp0.predict_concat(cat, off, nd=2)))
mp2 = MultiPredictor(gm, bench.FS, n_slots=2, **bench.MFCC_KW)
print("pageable, 2 slots  %.3f ms (one queue)" % timed(lambda: mp2.predict_concat(cat, off, nd=2)))
_lib.set_option("multi_merge_same_device", 0)
print("pageable, 2 slots  %.3f ms (a thread each)" % timed(lambda: mp2.predict_concat(cat, off, nd=2)))
_lib.set_option("multi_merge_same_device", 1)
del mp0, mp2
_lib.host_register(cat)
print("bare H2D update    %.3f ms (%.0f MB)" % (timed(lambda: pcm.update_pcm(cat)), cat.nbytes / 1e6))
mp_ = MultiPredictor(gm, bench.FS, n_slots=1, **bench.MFCC_KW)
for pieces in (0, 1, 2, 3, 4, 6, 8, 0):
    _lib.set_option("multi_pieces", pieces)
    print("multi_pieces %d     %.3f ms" % (pieces, timed(lambda: mp_.predict_concat(cat, off, nd=2))), flush=True)
