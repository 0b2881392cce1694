#!/usr/bin/env python3
"""Re-run one case of fuzz_generic.py (same seed stream) and take its worst frame apart: the oracle's largest mixture terms
against the device's score of the same mixtures as one-mixture models."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import gmm_oracle as go
from speaker_recognition_amd import _lib, synth
from speaker_recognition_amd.core import Batch, ModelSet
from speaker_recognition_amd.pygmm import GMM
go.build(ref=False)
target, seed = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
for c in range(target + 1):
    D, S = int(rng.integers(1, 65)), int(rng.integers(1, 7))
    shift = float(rng.choice([0.0, 0.0, 5.0, -40.0])); scale = float(rng.choice([1.0, 1.0, 0.05, 30.0]))
    models = []
    for s in range(S):
        K = int(rng.integers(1, 201))
        w, mu, sg = synth.synth_gmm(K, D, int(rng.integers(1 << 30)))
        w = w.copy()
        if K > 3 and rng.random() < 0.3:
            w[int(rng.integers(K))] = 0.0
        if rng.random() < 0.25 and K > 4:
            for k in rng.choice(K, size=2, replace=False):
                sg = sg.copy(); mu = mu.copy()
                sg[k] = 0.04
                mu[k] = np.round(mu.mean(0) + 2.0 * rng.choice([-1.0, 1.0], size=D), 4)
        f = lambda a: np.vectorize(lambda v: float("%g" % v))(a)
        models.append((w, f(mu * scale + shift), f(sg * scale)))
    lens = [int(v) for v in rng.choice([0, 1, 5, 127, 128, 129, 300, 700], size=int(rng.integers(1, 7)))]
    if sum(lens) == 0:
        continue
    utts = [synth.draw_frames(models[int(rng.integers(S))], n, int(rng.integers(1 << 30)), outlier_frac=float(rng.choice([0.0, 0.03]))) for n in lens]
    compat = bool(rng.integers(2))
X = np.concatenate(utts).astype(np.float64)
want = np.stack([go.score_batch(go.GMMParams(*m), X, go.MODE_LOGSUMEXP, clamp_compat=compat) for m in models])
ms = ModelSet([GMM.from_arrays(*m) for m in models])
_lib.set_option("score_engine", 1)
sums, arg, fll = ms.score(Batch.from_features(utts), frame_ll=True, clamp_compat=compat)
rel = np.abs(fll - want) / np.maximum(1.0, np.abs(want))
j = np.unravel_index(int(np.argmax(rel)), rel.shape)
print("case", target, "D", D, "S", S, "K", [len(m[0]) for m in models], "worst: model", j[0], "frame", j[1], "want", want[j], "got", fll[j])
w, mu, sg = models[j[0]]
x = X[j[1]]
terms = np.log(np.maximum(w, 1e-300)) - np.sum(np.log(np.sqrt(2 * np.pi) * sg), axis=1) - np.sum((x - mu) ** 2 / (2 * sg ** 2), axis=1)
order = np.argsort(-terms)[:4]
print("numpy LL", np.log(np.sum(np.exp(terms - terms.max()))) + terms.max())
for k in order:
    one = GMM.from_arrays(np.array([w[k]]), mu[k:k + 1], sg[k:k + 1])
    dev = one.score(x[None, :].astype(np.float32))[0]
    print("  mixture %3d: w %.4g sigma[min %.4g max %.4g] |mu|max %.3g  term %.6f  device(one-mixture model) %.6f  max|x-mu|/sigma %.1f" % (
        k, w[k], sg[k].min(), sg[k].max(), np.abs(mu[k]).max(), terms[k], dev, np.max(np.abs(x - mu[k]) / sg[k])))
print("x dtype check: max |x| %.3f, float32-representable %s" % (np.abs(x).max(), bool(np.all(x == x.astype(np.float32)))))
