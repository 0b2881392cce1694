#!/usr/bin/env python3
"""One bounded experiment: does the ORDER of the part-product slots inside the shared-sigma engine's flat
contraction move its time?  The kernel is bound by the socket's power cap (zero operands run 1.46x faster), and the low-part slots
toggle like noise; `h2s_slot_order` 0 keeps them in runs [lo x hi | hi x lo | hi x hi], 1 interleaves them per dimension.
Two ModelSets of the bench headline's 201 x 512 set, one per order, scored alternately on the same features (HIP-event kernel time)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from speaker_recognition_amd import _lib, synth  # noqa: E402
from speaker_recognition_amd.core import Batch, MfccExtractor, ModelSet  # noqa: E402
from speaker_recognition_amd.pygmm import GMM  # noqa: E402

utts = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
ex = MfccExtractor(bench.FS, **bench.MFCC_KW)
n_samples = (bench.FRAMES_PER_UTT + bench.ND - 1) * ex.FRAME_SHIFT + ex.FRAME_LEN
base = bench.base_clips(bench.CFG2_SPEAKERS, n_samples)
ubm = synth.synth_gmm(bench.CFG2_MIX, bench.DIM, 99)
raw = [ubm] + [synth.synth_map_speaker(ubm, 500 + s) for s in range(bench.CFG2_SPEAKERS)]
sets = {}
for order in (0, 1):
    _lib.set_option("h2s_slot_order", order)
    sets[order] = ModelSet([GMM.from_arrays(*m) for m in raw])
_lib.set_option("h2s_slot_order", 0)
cat, off = bench.make_pcm(base, utts, 0)
feats = ex.extract_batch(Batch.from_pcm((cat, off)), nd=bench.ND)
_lib.profile_enable(True)
ts, res = {0: [], 1: []}, {}
for r in range(5):
    for order in (0, 1):
        _lib.profile_reset()
        sums, arg = sets[order].score(feats)
        t, c = _lib.profile_get(_lib.T_SCORE)
        if r > 0:
            ts[order].append(t)
        res[order] = sums
d = float(np.max(np.abs(res[0] - res[1]) / np.maximum(1.0, np.abs(res[0]))))
print("h2s_slot_order A/B, %d utterances x 1000 frames, 201 x 512 x 39 (%s)" % (utts, _lib.last_score_kernel()[:48]))
print("  order 0 (runs):        %s ms" % " ".join("%.2f" % t for t in ts[0]))
print("  order 1 (interleaved): %s ms" % " ".join("%.2f" % t for t in ts[1]))
print("  ratio interleaved / runs = %.4f   max rel utterance-sum difference %.2e   argmax equal %s"
      % (np.median(ts[1]) / np.median(ts[0]), d, bool(np.array_equal(np.argmax(res[0], 1), np.argmax(res[1], 1)))))
