import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import bench
from speaker_recognition_amd import _lib, synth
from speaker_recognition_amd.core import Batch, MfccExtractor, ModelSet
from speaker_recognition_amd.pygmm import GMM
utts = 3000
ex = MfccExtractor(bench.FS, **bench.MFCC_KW)
n_samples = (bench.FRAMES_PER_UTT + bench.ND - 1) * ex.FRAME_SHIFT + ex.FRAME_LEN
base = bench.base_clips(bench.CFG2_SPEAKERS, n_samples)
ubm = synth.synth_gmm(bench.CFG2_MIX, bench.DIM, 99)
ms = ModelSet([GMM.from_arrays(*m) for m in [ubm] + [synth.synth_map_speaker(ubm, 500 + s) for s in range(bench.CFG2_SPEAKERS)]])
cat, off = bench.make_pcm(base, utts, 0)
feats = ex.extract_batch(Batch.from_pcm((cat, off)), nd=bench.ND)
_lib.profile_enable(True)
res = {}
for r in range(3):
    for v in (1, 0):
        _lib.set_option("score_h2s_exact_offset", v)
        _lib.profile_reset(); sums, arg = ms.score(feats)
        t, _ = _lib.profile_get(_lib.T_SCORE); tr, _ = _lib.profile_get(_lib.T_SCORE_REF)
        res[v] = (sums, arg)
        if r: print("exact_offset=%d: pre-pass %.3f ms, main %.2f ms" % (v, tr, t), flush=True)
d = np.abs(res[0][0] - res[1][0]) / np.maximum(1, np.abs(res[1][0]))
print("max rel sum diff", d.max(), "argmax equal", np.array_equal(res[0][1], res[1][1]), "exception stats", _lib.flush_stats())
