#!/usr/bin/env python3
"""Known answers from the REFERENCE DSO around its underflow clamp (safe_log, gmm.cc:34-38, reached
from GMM::log_probability_of_fast_exp, gmm.cc:237-244, under the FTZ arithmetic of its -ffast-math
build).  Run in the build container only:

    make -C oracle ref && python tests/golden/make_clamp_golden.py

The reference returns ln(1e-15) when its linear-domain sum is exactly 0, i.e. when every term
w_k p_k(x) flushed to zero -- the largest term below DBL_MIN = exp(-708.396) -- and the true
log-likelihood otherwise.  The band where "the log of the SUM is below -708.396" and "the largest
TERM is below -708.396" disagree is ln K wide; the frames here walk the largest term through
[-708.4 - ln K - 2, -708.4 + 2] for models whose mixtures contribute equally ("flat": the whole band
is populated) and for random models.

All sigmas are >= 0.5, so every per-dimension factor exp(.)/(sqrt(2 pi) sigma) is < 1: partial
products then decrease monotonically and the reference's flush of INTERMEDIATE products in dimension
order (which no log-domain formulation reproduces) cannot fire before the final product does; the
offset is spread over all dimensions, so the per-dimension exponent floor of fastexp.cc:104-131 is
never reached either.  Frames within 0.05 of the boundary are not generated (fp32 noise).

Output: tests/golden/clamp_golden.npz."""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import gmm_oracle as go  # noqa: E402

MINLOG = -7.08396418532264106224e2


def term_max(p, x):
    """largest ln(w_k p_k(x)) over mixtures, float64"""
    d = (x[None, :] - p.mean) / p.sigma
    v = np.log(p.weights) - np.sum(np.log(np.sqrt(2 * np.pi) * p.sigma), axis=1) - 0.5 * np.sum(d * d, axis=1)
    return float(v.max())


def r6(a):
    return np.vectorize(lambda v: float("%g" % v))(a)


def make_model(kind, K, D, seed):
    rng = np.random.default_rng(seed)
    sigma = rng.uniform(0.5, 1.5, (K, D))
    if kind == "flat":
        mean = np.tile(rng.standard_normal((1, D)), (K, 1)) + 1e-3 * rng.standard_normal((K, D))
        sigma = np.tile(sigma[:1], (K, 1))
        w = np.full(K, 1.0 / K)
    else:
        mean = rng.standard_normal((K, D))
        w = rng.dirichlet(np.ones(K))
    return go.GMMParams(r6(w), r6(mean), r6(sigma))


def frames_on_band(p, targets, seed):
    """For each target T: a frame whose largest term is T (to ~1e-6), found by bisection on the radius
    along a random direction from a random mixture mean; rounded to fp32 (both sides see the same input)."""
    rng = np.random.default_rng(seed)
    out = []
    for T in targets:
        k = int(rng.integers(p.K))
        u = rng.standard_normal(p.D)
        u = np.sign(u) * (0.7 + 0.3 * np.abs(u) / np.max(np.abs(u)))     # every dimension takes a share
        lo, hi = 0.0, 200.0
        for _ in range(80):
            mid = 0.5 * (lo + hi)
            if term_max(p, p.mean[k] + mid * u * p.sigma[k]) > T:
                lo = mid
            else:
                hi = mid
        out.append((p.mean[k] + lo * u * p.sigma[k]).astype(np.float32).astype(np.float64))
    return np.array(out)


def main():
    ref = go.RefLib()
    tmp = tempfile.mkdtemp()
    out = {}
    names = []
    for name, kind, K, D, seed in (("flat64", "flat", 64, 13, 1), ("flat256", "flat", 256, 39, 2),
                                   ("rand64", "rand", 64, 39, 3), ("rand256", "rand", 256, 20, 4)):
        p = make_model(kind, K, D, seed)
        path = os.path.join(tmp, name + ".model")
        with open(path, "w") as f:
            f.write(go.format_model_text(p))
        p = go.parse_model_text(open(path).read())
        lnk = np.log(K)
        targets = np.concatenate([np.linspace(MINLOG - lnk - 2.0, MINLOG - 0.06, 40),
                                  np.linspace(MINLOG + 0.06, MINLOG + 2.0, 16)])
        X = frames_on_band(p, targets, 50 + seed)
        tm = np.array([term_max(p, x) for x in X])
        keep = np.abs(tm - MINLOG) > 0.05
        X, tm = X[keep], tm[keep]
        h = ref.load(path)
        ll = ref.score_batch(h, X, 1)
        clamped = ll == np.log(1e-15)
        # the rule this file pins: clamp <=> largest term below DBL_MIN
        assert np.array_equal(clamped, tm < MINLOG), (name, tm[clamped != (tm < MINLOG)])
        # and where the sum-rule of round 1 differed from it
        lse = np.array([go.score_batch(p, x[None, :], go.MODE_LOGSUMEXP, clamp_compat=False)[0] for x in X])
        n_band = int(np.sum((lse >= MINLOG) & (tm < MINLOG)))
        print("%s: %d frames, %d clamped, %d in the band where log(sum) >= -708.396 > largest term" %
              (name, len(X), int(clamped.sum()), n_band))
        out[name + "_w"], out[name + "_mean"], out[name + "_sigma"] = p.weights, p.mean, p.sigma
        out[name + "_X"], out[name + "_ll"], out[name + "_termmax"] = X, ll, tm
        names.append(name)
    out["cases"] = np.array(names)
    np.savez_compressed(os.path.join(ROOT, "tests/golden/clamp_golden.npz"), **out)


if __name__ == "__main__":
    main()
