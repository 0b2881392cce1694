#!/usr/bin/env python3
"""A/B of an option of the split-fp16 shared-sigma engine inside ONE GPU-box call (box-to-box variance is 3-5 %):
`ab_h2s.py OPTION V0 V1 [UTTS]` times the scoring kernel (HIP events) under OPTION = V0 and = V1, alternating, on
 (a) the bench headline's workload -- MFCC features of the synthetic audio against SURVEY 8d's 201 x 512 set -- and
 (b) a configs[3]-shaped one -- frames drawn from the models of a 2048-mixture UBM + 1000 MAP speakers (250 k frames),
and reports the largest relative difference of the utterance sums between the two settings."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from speaker_recognition_amd import _lib, synth  # noqa: E402
from speaker_recognition_amd.core import Batch, MfccExtractor, ModelSet  # noqa: E402
from speaker_recognition_amd.pygmm import GMM  # noqa: E402


def time_both(name, ms, feats, opt, vals, rounds=3):
    _lib.profile_enable(True)
    res, ts = {}, {v: [] for v in vals}
    for r in range(rounds + 1):
        for v in vals:
            _lib.set_option(opt, v)
            _lib.profile_reset()
            sums, arg = ms.score(feats)
            t, c = _lib.profile_get(_lib.T_SCORE)
            if r > 0:
                ts[v].append(t)
            res[v] = sums
    d = float(np.max(np.abs(res[vals[0]] - res[vals[1]]) / np.maximum(1.0, np.abs(res[vals[0]]))))
    print("%s: %s=%d: %s ms | %s=%d: %s ms | ratio %.4f | max rel sum diff %.2e | %s" % (
        name, opt, vals[0], " ".join("%.2f" % t for t in ts[vals[0]]), opt, vals[1], " ".join("%.2f" % t for t in ts[vals[1]]),
        np.median(ts[vals[1]]) / np.median(ts[vals[0]]), d, _lib.last_score_kernel()[:60]), flush=True)


def main():
    opt, v0, v1 = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    utts = int(sys.argv[4]) if len(sys.argv) > 4 else 3000
    ex = MfccExtractor(bench.FS, **bench.MFCC_KW)
    n_samples = (bench.FRAMES_PER_UTT + bench.ND - 1) * ex.FRAME_SHIFT + ex.FRAME_LEN
    base = bench.base_clips(bench.CFG2_SPEAKERS, n_samples)
    ubm = synth.synth_gmm(bench.CFG2_MIX, bench.DIM, 99)
    ms = ModelSet([GMM.from_arrays(*m) for m in [ubm] + [synth.synth_map_speaker(ubm, 500 + s) for s in range(bench.CFG2_SPEAKERS)]])
    cat, off = bench.make_pcm(base, utts, 0)
    feats = ex.extract_batch(Batch.from_pcm((cat, off)), nd=bench.ND)
    time_both("headline-shaped (%d utts, MFCC features)" % utts, ms, feats, opt, (v0, v1))
    if os.environ.get("AB_QUICK"):
        return
    # the same set against frames DRAWN FROM ITS MODELS (sparse posteriors: what trained models see)
    drawn = Batch.from_features([synth.draw_frames(ubm, 1000, 70 + u, outlier_frac=0.001) for u in range(min(utts, 1000))])
    time_both("201 x 512, frames drawn from the UBM", ms, drawn, opt, (v0, v1))
    del ms, feats, drawn
    ubm3, spk3 = bench.cfg3_models()
    ms3 = ModelSet([GMM.from_arrays(*m) for m in [ubm3] + spk3])
    f3 = Batch.from_features([synth.draw_frames(spk3[u % 1000], 1000, 9000 + u, outlier_frac=0.001) for u in range(250)])
    time_both("configs[3]-shaped (250 k frames drawn from the models)", ms3, f3, opt, (v0, v1), rounds=2)


if __name__ == "__main__":
    main()
