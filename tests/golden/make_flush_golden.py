#!/usr/bin/env python3
"""Known answers from the REFERENCE DSO for frames on which its flush-to-zero arithmetic decides the
result BEFORE the mixture's full product exists.  Build
container only:

    make -C oracle ref && python tests/golden/make_flush_golden.py

`Gaussian::probability_of_fast_exp` (gmm.cc:176-202) multiplies D per-dimension factors
exp(-d^2/2s^2)/(sqrt(2 pi) s) in the LINEAR domain; the DSO's -ffast-math start-up code sets FTZ/DAZ,
so any intermediate below DBL_MIN = exp(-708.396) is exactly 0 and stays 0 -- even when later factors
> 1 (s < 0.399: the rule for delta features) would have lifted the full product back above DBL_MIN.
Which intermediates exist is the compiler's choice under -ffast-math; for the reference's flags and
g++ 11 (oracle/Makefile; read off the disassembly, restated in oracle/gmm_oracle.c order 2) they are,
per mixture:

  (1) e_i = remez5(max(b_i, -708.396)), b_i = -d_i^2/(2 s_i^2): at the floor the polynomial is < 1, so
      e_i = 0 (fastexp.cc:104-105,128-131,195-206);
  (2) e_i * 0.39894 (the folded 1/sqrt(2 pi)) -- flushes when b_i < -708.396 + 0.919;
  (3) ( .. ) / s_i;
  (4) two running products: even dimensions, odd dimensions (pairs; the last dimension of an odd D apart);
  (5) even * odd; (6) * last dimension (odd D); (7) * w_k (gmm.cc:241).

`flush_rule` below is that list in the log domain (numpy, float64); it is used to (a) construct frames
that sit on a chosen side of a chosen rule with a margin of >= 0.03 nats on EVERY decision of every
mixture that could matter (fp32 inputs and parameters move a decision quantity by ~1e-4), and (b)
assert that the DSO's answer is the log-sum of exactly the surviving mixtures.

Output: tests/golden/flush_golden.npz -- per case the 6-digit model, fp32-representable frames, the
DSO's per-frame LL, what the full-product rule of round 2 (lse.hpp / oracle mode 2) would give, and
which rule kills the best mixture."""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import gmm_oracle as go  # noqa: E402

T = -7.08396418532264106224e2            # ln DBL_MIN = fastexp.cc's minlog
LN_SQRT_2PI = 0.9189385332046727
MARGIN = 0.03


def flush_rule(p, x, order=2):
    """Log-domain restatement of the DSO's flush decisions.
    Returns (term[K] = ln(w_k p_k) of the full product, alive[K], margin[K] = distance of the closest
    decision quantity from the boundary, rule[K] = first rule that killed the mixture or 0)."""
    K, D = p.mean.shape
    b = -0.5 * ((x[None, :] - p.mean) / p.sigma) ** 2
    lns = np.log(p.sigma)
    q = []                                        # decision quantities, each compared with T, in evaluation order
    rule = []
    if order == 2:
        u = b - LN_SQRT_2PI
        l = u - lns
        paired = D & ~1
        for i in range(D):
            q += [b[:, i], u[:, i], l[:, i]]
            rule += [1, 2, 3]
        for lane in (0, 1):
            acc = np.zeros(K)
            for i in range(lane, paired, 2):
                acc = acc + l[:, i]
                q.append(acc.copy())
                rule.append(4)
        tot = l[:, :paired].sum(axis=1)
        q.append(tot.copy()); rule.append(5)
        if D & 1:
            tot = tot + l[:, D - 1]
            q.append(tot.copy()); rule.append(6)
    else:
        l = b - LN_SQRT_2PI - lns
        acc = np.zeros(K)
        for i in range(D):
            q += [b[:, i], l[:, i]]
            rule += [1, 3]
            acc = acc + l[:, i]
            q.append(acc.copy()); rule.append(4)
        tot = acc
    with np.errstate(divide="ignore"):
        term = tot + np.log(p.weights)
    q.append(term.copy()); rule.append(7)
    q = np.array(q)                               # [n_decisions][K]
    dead = q < T
    alive = ~dead.any(axis=0)
    first = np.where(dead.any(axis=0), np.array(rule)[np.argmax(dead, axis=0)], 0)
    margin = np.min(np.abs(q - T), axis=0)
    return term, alive, margin, first


def full_product_rule(term):
    """what round 2 implemented: a term lives iff its FULL product is above DBL_MIN"""
    live = term >= T
    if not live.any():
        return go.LN_1E_15
    m = term[live].max()
    return float(m + np.log(np.sum(np.exp(term[live] - m))))


def ll_from(term, alive):
    if not alive.any():
        return go.LN_1E_15
    m = term[alive].max()
    return float(m + np.log(np.sum(np.exp(term[alive] - m))))


def r6(a):
    return np.vectorize(lambda v: float("%g" % v))(a)


def make_model(K, D, seed, small_frac, flat, conditioned):
    rng = np.random.default_rng(seed)
    sigma = np.where(rng.random((K, D)) < small_frac, rng.uniform(0.04, 0.35, (K, D)), rng.uniform(0.45, 1.5, (K, D)))
    if conditioned:
        # per dimension one sigma scale for all mixtures (x0.7 .. x1.4) and means within ~1 sigma of the origin: the
        # expanded quadratic form of the matrix-core engines stays well conditioned (amp ~ D), as for trained models
        sigma = np.where(rng.random((1, D)) < small_frac, rng.uniform(0.05, 0.3, (1, D)), rng.uniform(0.5, 1.2, (1, D))) * \
            rng.uniform(0.7, 1.4, (K, D))
        mean = 0.8 * sigma * rng.standard_normal((K, D))
    else:
        mean = rng.standard_normal((K, D))
    w = rng.dirichlet(np.ones(K))
    if flat:                                       # near-identical mixtures: several terms sit in the band together
        mean = np.tile(mean[:1], (K, 1)) + 0.02 * rng.standard_normal((K, D))
        sigma = np.tile(sigma[:1], (K, 1)) * np.exp(0.01 * rng.standard_normal((K, D)))
        w = np.full(K, 1.0 / K)
    return go.GMMParams(r6(w), r6(mean), r6(sigma))


def adapted(p, seed):
    """a MAP-style speaker of the UBM `p`: means moved, sigma and weights shared (gmmubm.cc:40-81)"""
    rng = np.random.default_rng(seed)
    return go.GMMParams(p.weights, r6(p.mean + 0.1 * p.sigma * rng.standard_normal(p.mean.shape)), p.sigma)


def dip_level(p, x):
    """max over mixtures of the smallest decision quantity: the mixture that comes closest to surviving, minus T"""
    K, D = p.mean.shape
    b = -0.5 * ((x[None, :] - p.mean) / p.sigma) ** 2
    l = b - LN_SQRT_2PI - np.log(p.sigma)
    paired = D & ~1
    q = [b.min(axis=1) - LN_SQRT_2PI, l.min(axis=1)]
    for lane in (0, 1):
        q.append(np.cumsum(l[:, lane:paired:2], axis=1).min(axis=1))
    tot = l.sum(axis=1)
    q.append(l[:, :paired].sum(axis=1))
    with np.errstate(divide="ignore"):
        q.append(tot + np.log(p.weights))
    return float(np.max(np.min(np.array(q), axis=0)) - T)


def draw_frames(p, n, seed, also=()):
    """Frames with the offset concentrated on 1..4 dimensions -- leading ones half of the time -- and scaled (bisection)
    so that the mixture that comes CLOSEST to surviving has its tightest partial product V nats below DBL_MIN, V in
    (-4, 8): it dips below and -- where the other dimensions' factors are > 1 (sigma < 0.399) -- its full product recovers;
    kept only when every decision of every mixture within 40 nats of the best one is >= MARGIN from its boundary."""
    rng = np.random.default_rng(seed)
    K, D = p.mean.shape
    out, info = [], []
    tries = 0
    while len(out) < n and tries < 200 * n:
        tries += 1
        k = int(rng.integers(K))
        nd = int(rng.integers(1, 5))
        dims = rng.permutation(min(D, 6))[:nd] if rng.random() < 0.5 else rng.permutation(D)[:nd]
        share = rng.dirichlet(np.ones(nd))
        base = p.mean[k] + 0.15 * p.sigma[k] * rng.standard_normal(D)
        off = np.zeros(D)
        for d, sh in zip(dims, share):
            off[d] = np.sign(rng.standard_normal()) * p.sigma[k, d] * np.sqrt(2.0 * 700.0 * sh)
        V = rng.uniform(-4.0, 8.0)
        lo, hi = 0.0, 3.0
        for _ in range(60):
            mid = 0.5 * (lo + hi)
            if dip_level(p, base + mid * off) > -V:
                lo = mid
            else:
                hi = mid
        x = (base + lo * off).astype(np.float32).astype(np.float64)
        term, alive, margin, first = flush_rule(p, x)
        near = term >= term.max() - 40.0
        if margin[near].min() < MARGIN:
            continue
        ok = True
        for q in also:                                # the speakers of a shared-sigma set see the same frames
            t2, _, m2, _ = flush_rule(q, x)
            ok &= m2[t2 >= t2.max() - 40.0].min() >= MARGIN
        if not ok:
            continue
        out.append(x)
        info.append(int(first[np.argmax(term)]))
    return np.array(out), np.array(info)


def main():
    # name, K, D, seed, fraction of sigmas < 0.399, flat, conditioned for the matrix-core engines, speakers sharing sigma
    cases = (("d13_k1", 1, 13, 11, 0.9, False, False, 0), ("d39_k8", 8, 39, 12, 0.8, False, False, 0),
             ("d39_flat32", 32, 39, 13, 0.7, True, True, 0), ("d20_k64", 64, 20, 14, 0.9, False, True, 0),
             ("d34_k16", 16, 34, 15, 0.7, False, False, 0), ("d39_flat256", 256, 39, 16, 0.6, True, True, 0),
             ("d26_k4", 4, 26, 17, 1.0, False, False, 0), ("d39_k32c", 32, 39, 18, 0.8, False, True, 0),
             ("d39_ubm64", 64, 39, 19, 0.8, False, True, 13))
    built = []
    for name, K, D, seed, small, flat, cond, n_spk in cases:
        p = make_model(K, D, seed, small, flat, cond)
        p = go.parse_model_text(go.format_model_text(p))
        spk = [go.parse_model_text(go.format_model_text(adapted(p, 1000 * seed + i))) for i in range(n_spk)]
        X, first = draw_frames(p, 160, 100 + seed, spk)
        models = [p] + spk
        rule_ll = np.array([[ll_from(*flush_rule(q, x)[:2]) for x in X] for q in models])
        src_ll = np.array([[ll_from(*flush_rule(q, x, order=1)[:2]) for x in X] for q in models])
        full_ll = np.array([[full_product_rule(flush_rule(q, x)[0]) for x in X] for q in models])
        built.append((name, models, X, first, rule_ll, src_ll, full_ll))
    # the DSO switches the process to FTZ/DAZ when it loads: everything numpy had to do is done
    ref = go.RefLib()
    tmp = tempfile.mkdtemp()
    out, names = {}, []
    for name, models, X, first, rule_ll, src_ll, full_ll in built:
        ll = []
        for i, q in enumerate(models):
            path = os.path.join(tmp, "%s_%d.model" % (name, i))
            with open(path, "w") as f:
                f.write(go.format_model_text(q))
            ll.append(ref.score_batch(ref.load(path), X, 1))
        ll = np.array(ll)
        err = np.abs(ll - rule_ll)
        assert err.max() < 3e-4, (name, err.max(), np.argmax(err))          # remez5's 1e-5 per factor, D factors
        orc = np.array([go.score_batch(q, X, go.MODE_FASTEXP) for q in models])
        assert np.max(np.abs(orc - ll)) < 1e-9, (name, np.max(np.abs(orc - ll)))
        n_diff = int(np.sum(np.abs(full_ll - ll) > 1e-3))
        n_src = int(np.sum(np.abs(src_ll - ll) > 1e-3))
        print("%-12s %2d model(s) x %3d frames: DSO = as-compiled rule (max |d| %.1e), oracle mode 0 = DSO (%.1e); the "
              "full-product rule differs on %d values, the source-order rule on %d; best mixture killed by rule %s" %
              (name, len(models), len(X), err.max(), np.max(np.abs(orc - ll)), n_diff, n_src,
               {int(k): int(v) for k, v in zip(*np.unique(first, return_counts=True))}))
        p = models[0]
        out[name + "_w"], out[name + "_mean"], out[name + "_sigma"] = p.weights, p.mean, p.sigma
        if len(models) > 1:
            out[name + "_spk_mean"] = np.array([q.mean for q in models[1:]])
        out[name + "_X"], out[name + "_ll"], out[name + "_full_rule_ll"], out[name + "_src_order_ll"] = X, ll, full_ll, src_ll
        out[name + "_first_rule"] = first
        names.append(name)
    out["cases"] = np.array(names)
    np.savez_compressed(os.path.join(ROOT, "tests/golden/flush_golden.npz"), **out)


if __name__ == "__main__":
    main()
