#!/usr/bin/env python3
"""Randomised check of the generic engines (vector ALU, split-bf16, split-fp16, automatic incl. the hybrid
form) against the float64 oracle: random dims 1..64 (one case in four: 65..330, the wide rows of the vector engine and its D-sliced
kernels, round 6), K 1..200, 1..6 models of different sizes, weights with zeros,
shifted / scaled feature spaces, ragged utterances, far outliers, clamp on / off.  `fuzz_generic.py [cases] [seed]`"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import gmm_oracle as go  # noqa: E402
from speaker_recognition_amd import _lib, synth  # noqa: E402
from speaker_recognition_amd.core import Batch, ModelSet  # noqa: E402
from speaker_recognition_amd.pygmm import GMM  # noqa: E402

go.build(ref=False)
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
worst = {}
fails = 0
for c in range(cases):
    D, S = int(rng.integers(1, 65)), int(rng.integers(1, 7))
    if rng.random() < 0.25:
        D = int(rng.integers(65, 331))
    shift = float(rng.choice([0.0, 0.0, 5.0, -40.0]))          # feature spaces away from the origin
    scale = float(rng.choice([1.0, 1.0, 0.05, 30.0]))
    models = []
    for s in range(S):
        K = int(rng.integers(1, 201 if D <= 64 else 41))
        w, mu, sg = synth.synth_gmm(K, D, int(rng.integers(1 << 30)))
        w = w.copy()
        if K > 3 and rng.random() < 0.3:
            w[int(rng.integers(K))] = 0.0
        if rng.random() < 0.25 and K > 4:                       # a couple of collapsed, far components
            for k in rng.choice(K, size=2, replace=False):
                sg = sg.copy(); mu = mu.copy()
                sg[k] = 0.04
                mu[k] = np.round(mu.mean(0) + 2.0 * rng.choice([-1.0, 1.0], size=D), 4)
        f = lambda a: np.vectorize(lambda v: float("%g" % v))(a)   # what the 6-digit text format keeps
        models.append((w, f(mu * scale + shift), f(sg * scale)))
    lens = [int(v) for v in rng.choice([0, 1, 5, 127, 128, 129, 300, 700], size=int(rng.integers(1, 7)))]
    if sum(lens) == 0:
        continue
    utts = [synth.draw_frames(models[int(rng.integers(S))], n, int(rng.integers(1 << 30)), outlier_frac=float(rng.choice([0.0, 0.03]))) for n in lens]
    X = np.concatenate(utts).astype(np.float64)
    compat = bool(rng.integers(2))
    want = np.stack([go.score_batch(go.GMMParams(*m), X, go.MODE_LOGSUMEXP, clamp_compat=compat) for m in models])
    ms = ModelSet([GMM.from_arrays(*m) for m in models])
    for eng in ((0, 1, 3, 5) if D <= 64 else (0, 1)):
        _lib.set_option("score_engine", eng)
        try:
            sums, arg, fll = ms.score(Batch.from_features(utts), frame_ll=True, clamp_compat=compat)
        except _lib.SRError as e:
            print("  case %d engine %d refused: %s" % (c, eng, str(e)[:80]))
            continue
        rel = np.abs(fll - want) / np.maximum(1.0, np.abs(want))
        if compat:
            # at the reference's underflow boundary the outcome flips with the last bits of a term (float64 oracle vs fp32
            # device; the reference's own polynomial exp has 1e-5 relative error there): frames within ln K + 1 nats of it
            # are left out of the comparison, as in tests/golden/make_clamp_golden.py
            kmax = max(len(m[0]) for m in models)
            unc = np.stack([go.score_batch(go.GMMParams(*m), X, go.MODE_LOGSUMEXP, clamp_compat=False) for m in models])
            band = np.abs(unc + 708.396) < np.log(kmax) + 1.0
            rel = np.where(band, 0.0, rel)
        err = float(np.max(rel))
        if err >= 1e-4:
            j = np.unravel_index(int(np.argmax(rel)), rel.shape)
            print("    worst frame: model %d frame %d want %.6f got %.6f" % (j[0], j[1], want[j], fll[j]))
        name = _lib.last_score_kernel()
        worst[eng] = max(worst.get(eng, 0.0), err)
        # forced matrix-core engines are allowed to be inexact on sets the dispatcher would not give them (ill conditioned)
        limit = 1e-4
        if err >= limit:
            amp = ms.info()["amp"]
            flag = "(forced engine on an ill-conditioned set, amp %.0f)" % amp if eng in (3, 5) and amp > 1000 else "!!"
            # fp32 sums of D squares against a log-likelihood that happens to cancel to |LL| < 1: sqrt(D) half-ulps of a sum of
            # ~0.72 D (log2 units) are 1e-4 absolute from D ~ 120 on -- the gate is relative to max(1, |LL|), the noise to the terms
            jw = np.unravel_index(int(np.argmax(rel)), rel.shape)
            if flag == "!!" and D > 64 and err < 3e-4 and abs(want[jw]) < 2.0:
                flag = "(wide row, |LL| < 2: fp32 summation noise of %d terms)" % D
            if flag == "!!":
                fails += 1
            print("  case %d D %d S %d shift %g scale %g clamp %d engine %d: err %.2e %s [%s]" % (c, D, S, shift, scale, compat, eng, err, flag, name[:50]))
    print("case %2d D %2d S %d shift %5g scale %5g frames %4d clamp %d  auto -> %s" % (c, D, S, shift, scale, sum(lens), compat, name[:40] if eng == 5 else ""))
_lib.set_option("score_engine", 0)
print("worst per engine:", {k: "%.2e" % v for k, v in worst.items()}, "unexplained failures:", fails)
