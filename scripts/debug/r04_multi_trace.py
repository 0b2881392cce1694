#!/usr/bin/env python3
"""resident predict step vs sr_multi_predict_pcm (1 slot, N pieces) on configs[1]: run under rocprofv3 --kernel-trace to compare launches."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from speaker_recognition_amd import _lib, synth
from speaker_recognition_amd.core import Batch, MfccExtractor, ModelSet, MultiPredictor
from speaker_recognition_amd.pygmm import GMM
pieces = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ex = MfccExtractor(bench.FS, **bench.MFCC_KW)
n_samples = (bench.FRAMES_PER_UTT + bench.ND - 1) * ex.FRAME_SHIFT + ex.FRAME_LEN
base_c = bench.base_clips(bench.CFG1_MODELS, n_samples)
gm = [GMM.from_arrays(*synth.synth_gmm(bench.CFG1_MIX, bench.DIM, bench.MODEL_SEED + s)) for s in range(bench.CFG1_MODELS)]
cat, off = bench.make_pcm(base_c, bench.CFG1_UTTS, 0)
ms = ModelSet(gm)
pcm = Batch.from_pcm((cat, off))
for _ in range(4):
    ex.predict_batch(ms, pcm, nd=2)
_lib.synchronize()
_lib.host_register(cat)
_lib.set_option("multi_pieces", pieces)
mp_ = MultiPredictor(gm, bench.FS, n_slots=1, **bench.MFCC_KW)
for _ in range(4):
    mp_.predict_concat(cat, off, nd=2)
print("done", _lib.last_score_kernel()[:50])
