"""CPU: pins the oracle (oracle/) against the known-answer vectors generated from the
reference itself (tests/golden/make_golden.py)."""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from oracle import mfcc_oracle as mo


def _params(go, g, c):
    return go.GMMParams(g[c + "_w"], g[c + "_mean"], g[c + "_sigma"])


def test_gmm_oracle_matches_reference_dso(oracle_built, gmm_golden):
    go, g = oracle_built, gmm_golden
    for c in g["cases"]:
        p = _params(go, g, c)
        ll = go.score_batch(p, g[c + "_X"], go.MODE_FASTEXP)
        # scalar restatement of the SSE2 polynomial vs the -ffast-math DSO: rounding-order only
        assert np.max(np.abs(ll - g[c + "_ll"])) < 1e-12, c
        assert abs(np.sum(ll) - float(g[c + "_sum"])) < 1e-9 * max(1, abs(float(g[c + "_sum"])))


def test_gmm_oracle_modes_agree_and_clamp(oracle_built, gmm_golden):
    go, g = oracle_built, gmm_golden
    for c in g["cases"]:
        p = _params(go, g, c)
        ref = g[c + "_ll"]
        lse = go.score_batch(p, g[c + "_X"], go.MODE_LOGSUMEXP)
        libm = go.score_batch(p, g[c + "_X"], go.MODE_LIBM)
        # remez5 error of the reference's fast exp: <= ~2e-6 absolute
        assert np.max(np.abs(lse - ref)) < 3e-6, c
        assert np.max(np.abs(libm - ref)) < 3e-6, c
        # the two outlier frames underflow -> safe_log floor ln(1e-15) (gmm.cc:34-38)
        assert np.all(ref[-2:] == go.LN_1E_15) and np.all(lse[-2:] == go.LN_1E_15)
        true_ll = go.score_batch(p, g[c + "_X"], go.MODE_LOGSUMEXP, clamp_compat=False)
        assert np.all(true_ll[-2:] < go.MINLOG)


def test_gmm_oracle_clamp_band_matches_reference(oracle_built, clamp_golden):
    """Frames that walk the largest term w_k p_k through DBL_MIN: the reference DSO returns
    ln(1e-15) exactly when every term flushed (gmm.cc:34-38, :237-244 under FTZ).  Mode 0 (the
    restated linear-domain arithmetic) reproduces the DSO to rounding order; mode 2 (log-sum-exp,
    the kernels' formulation) reproduces the clamp pattern exactly and the values to the remez5
    error -- including the ln K wide band where the log of the SUM is still above -708.396."""
    go, g = oracle_built, clamp_golden
    in_band = 0
    for c in g["cases"]:
        p = _params(go, g, c)
        X, ref, tm = g[c + "_X"], g[c + "_ll"], g[c + "_termmax"]
        clamped = ref == go.LN_1E_15
        assert np.array_equal(clamped, tm < go.MINLOG), c
        fast = go.score_batch(p, X, go.MODE_FASTEXP)
        assert np.array_equal(fast == go.LN_1E_15, clamped), c
        # just above DBL_MIN the reference's partial sums lose bits to FTZ: compare loosely there
        assert np.max(np.abs(fast - ref)) < 1e-9, c
        lse = go.score_batch(p, X, go.MODE_LOGSUMEXP)
        assert np.array_equal(lse == go.LN_1E_15, clamped), c
        assert np.max(np.abs(lse - ref) / np.maximum(1, np.abs(ref))) < 1e-7, c
        unclamped = go.score_batch(p, X, go.MODE_LOGSUMEXP, clamp_compat=False)
        in_band += int(np.sum(clamped & (unclamped >= go.MINLOG)))
    assert in_band >= 40          # the band the sum-rule would have got wrong is populated


def test_gmm_oracle_partial_product_flushes_match_reference(oracle_built, flush_golden):
    """SURVEY 8a-12, the part round 2 left open: the reference's FTZ arithmetic zeroes a mixture as soon as ANY
    intermediate of its density product dips below DBL_MIN (gmm.cc:192-195) or one dimension's exponent reaches the
    floor of fastexp.cc:104-131 -- also when the full product would be representable.  Goldens from the reference
    DSO (tests/golden/make_flush_golden.py).  Mode 0 of the oracle (the as-compiled order of the partial products)
    equals the DSO; the full-product rule (mode 2, what the engines compute) does not on a sixth of the values; every
    value on which it differs lies inside the band the engines hand to the exact path (csrc/lse.hpp, gmm_flush.hip)."""
    from conftest import flush_models
    go, g = oracle_built, flush_golden
    T = go.MINLOG
    n_diff = n_src = n_all = 0
    for c in g["cases"]:
        X, ref = g[c + "_X"], g[c + "_ll"]
        models = [go.GMMParams(*m) for m in flush_models(g, c)]
        band = max(float(np.max(np.sum(np.maximum(0.0, -np.log(p.sigma)), axis=1))) for p in models) + np.log(models[0].K) + 17.5
        for i, p in enumerate(models):
            assert np.max(np.abs(go.score_batch(p, X, go.MODE_FASTEXP) - ref[i])) < 1e-9, c
            full = go.score_batch(p, X, go.MODE_LOGSUMEXP, clamp_compat=True)
            assert np.max(np.abs(full - g[c + "_full_rule_ll"][i])) < 1e-6, c
            differs = np.abs(full - ref[i]) > 1e-3
            # where the engines' rule is wrong its value sits in [ln DBL_MIN, band_hi): noted, re-evaluated exactly
            assert np.all(full[differs] >= T) and np.all(full[differs] < T + band - 1.0), c
            n_diff += int(differs.sum())
            n_all += differs.size
            go.set_flush_order(1)
            try:
                src = go.score_batch(p, X, go.MODE_FASTEXP)
            finally:
                go.set_flush_order(2)
            assert np.max(np.abs(src - g[c + "_src_order_ll"][i])) < 3e-4, c      # (numpy rule vs remez5 arithmetic)
            n_src += int(np.sum(np.abs(src - ref[i]) > 1e-3))
    assert n_diff >= 150 and n_src >= 150, (n_diff, n_src, n_all)


def test_gmm_oracle_fast_mode_equals_the_reference(oracle_built, gmm_golden, clamp_golden, flush_golden):
    """Mode 3 (the bulk checker of bench.py: per-mixture constants, vectorised across mixtures, the reference's own
    arithmetic only in the band where its partial-product flushes decide) against the reference DSO's goldens: remez5's
    polynomial error (<= 1.2e-6 absolute) is all that separates them, clamp decisions are identical."""
    from conftest import flush_models
    go = oracle_built
    for g in (gmm_golden, clamp_golden):
        for c in g["cases"]:
            ref = g[c + "_ll"]
            got = go.score_batch(go.GMMParams(g[c + "_w"], g[c + "_mean"], g[c + "_sigma"]), g[c + "_X"], go.MODE_FAST)
            assert np.array_equal(got == go.LN_1E_15, ref == go.LN_1E_15), c
            assert np.max(np.abs(got - ref)) < 2e-6, c
    g = flush_golden
    for c in g["cases"]:
        for i, m in enumerate(flush_models(g, c)):
            ref = g[c + "_ll"][i]
            got = go.score_batch(go.GMMParams(*m), g[c + "_X"], go.MODE_FAST)
            assert np.array_equal(got == go.LN_1E_15, ref == go.LN_1E_15), c
            assert np.max(np.abs(got - ref)) < 2e-6, c


def test_model_text_roundtrip(oracle_built, gmm_golden):
    go, g = oracle_built, gmm_golden
    p = _params(go, g, "syn16x13")
    q = go.parse_model_text(go.format_model_text(p))
    assert q.K == 16 and q.D == 13
    # synthetic models were generated through the 6-significant-digit format already
    assert np.array_equal(q.weights, p.weights) and np.array_equal(q.mean, p.mean)
    assert np.array_equal(q.sigma, p.sigma)


def test_em_iteration_increases_likelihood(oracle_built, gmm_golden):
    go, g = oracle_built, gmm_golden
    p = _params(go, g, "syn5x3")
    rng = np.random.default_rng(0)
    k = rng.choice(p.K, 400, p=p.weights / p.weights.sum())
    X = p.mean[k] + p.sigma[k] * rng.standard_normal((400, p.D))
    start = go.GMMParams(np.full(p.K, 1.0 / p.K), p.mean + 0.3, np.ones_like(p.sigma))
    l0 = go.score_all(start, X, go.MODE_LOGSUMEXP)
    nxt = go.em_iteration(start, X)
    l1 = go.score_all(nxt, X, go.MODE_LOGSUMEXP)
    assert l1 > l0
    assert abs(nxt.weights.sum() - 1) < 1e-12 and np.all(nxt.sigma >= np.sqrt(1e-3))
    m = go.em_iteration(start, X, map_relevance=16.0, ubm=start)
    assert np.array_equal(m.weights, start.weights) and np.array_equal(m.sigma, start.sigma)
    assert not np.array_equal(m.mean, start.mean)


def test_mfcc_oracle_matches_reference(mfcc_golden):
    m = mfcc_golden
    assert np.allclose(mo.hamming(4), m["hamming4"], rtol=0, atol=1e-15)
    assert np.allclose(mo.hamming(4), [0.2147, 0.8653, 0.8653, 0.2147], atol=5e-5)
    for c in m["cases"]:
        kw = eval(str(m[c + "_kw"]))
        fs, pcm = int(m[c + "_fs"]), m[c + "_pcm"]
        ex = mo.get_mfcc_extractor(fs, **kw)
        assert np.max(np.abs(ex.raw_cepstra(pcm.astype(float)) - m[c + "_raw"])) < 1e-9, c
        assert np.max(np.abs(mo.extract(fs, pcm, **kw) - m[c + "_feat"])) < 1e-9, c
        assert np.max(np.abs(mo.extract(fs, pcm, diff=True, **kw) - m[c + "_d1"])) < 1e-9, c
        assert np.max(np.abs(mo.extract((fs, pcm), diff=True, nd=2, **kw) - m[c + "_d2"])) < 1e-9, c
        T = m[c + "_feat"].shape[0]
        assert m[c + "_d1"].shape == (T - 1, 2 * m[c + "_feat"].shape[1])
        assert m[c + "_d2"].shape == (T - 2, 3 * m[c + "_feat"].shape[1])


def test_mfcc_constants(mfcc_golden):
    m = mfcc_golden
    ex = mo.get_mfcc_extractor(16000)
    assert ex.FRAME_LEN == 512 and ex.FRAME_SHIFT == 256
    assert np.array_equal(ex.window, m["default16k_window"])
    assert np.allclose(ex.D, m["default16k_D"], rtol=0, atol=1e-15)
    r, c = np.nonzero(ex.M)
    assert np.array_equal(r, m["default16k_M_nnz_rows"]) and np.array_equal(c, m["default16k_M_nnz_cols"])
    assert len(r) == 1989 and c.min() == 1 and c.max() == 1023
    assert np.allclose(ex.M[r, c], m["default16k_M_vals"], rtol=0, atol=1e-13)


def test_signal_too_short_asserts():
    with pytest.raises(AssertionError):
        mo.extract(16000, np.zeros(5 * 512, dtype=np.int16))          # MFCC.py:56


def test_map_adaptation_matches_reference_dso(oracle_built, gmm_golden):
    """train_model_from_ubm is deterministic in the reference (it starts from the UBM copy); the
    oracle's MAP iteration reproduces its dumped models to the 6 digits the text format keeps."""
    go, g = oracle_built, gmm_golden
    ubm = _params(go, g, "syn16x13")
    X = g["map_X"]
    q = ubm
    for it in range(1, 5):
        q = go.em_iteration(q, X, map_relevance=16.0, ubm=ubm)
        if it in (1, 4):
            ref = g["map%d_mean" % it]
            assert np.max(np.abs(q.mean - ref) / np.maximum(1.0, np.abs(ref))) < 1e-5, it
            assert np.array_equal(g["map%d_w" % it], ubm.weights)        # gmmubm.cc:40-42
            assert np.array_equal(g["map%d_sigma" % it], ubm.sigma)      # gmmubm.cc:76-78


def _em_start(g):
    X = g["em_X"]
    sig = np.sqrt(((X - X.mean(0)) ** 2).sum(0) / (len(X) - 1))        # gmm.cc:309-325
    return X, np.full(8, 1.0 / 8), X[g["em_init_rows"]].copy(), np.tile(sig, (8, 1))


def test_em_matches_reference_trainer(oracle_built, gmm_golden):
    """Full EM against the reference's own train_model: same start (its rand()-seeded draw repeats
    across fresh processes; see tests/golden/make_golden.py), models after 1, 2 and 6 iterations equal
    to the 6 digits its dump keeps -- E-step, weights, means, variances and the sigma floor."""
    go, g = oracle_built, gmm_golden
    X, w, mu, sg = _em_start(g)
    assert np.max(np.abs(sg[0] - g["em_init_sigma_dumped"][0]) / sg[0]) < 1e-5
    p = go.GMMParams(w, mu, sg)
    for it in range(1, 7):
        p = go.em_iteration(p, X)
        if it in (1, 2, 6):
            assert np.max(np.abs(p.weights - g["em%d_w" % it])) < 2e-6, it
            assert np.max(np.abs(p.mean - g["em%d_mean" % it])) < 2e-5, it
            assert np.max(np.abs(p.sigma - g["em%d_sigma" % it]) / g["em%d_sigma" % it]) < 2e-5, it


def test_lpc_oracle_solves_the_normal_equations():
    """LPC (talkbox, third-party, absent -> parity unpinned): the restated Levinson-Durbin recursion
    solves the Toeplitz normal equations of the restated biased autocorrelation (independent solver),
    and the autocorrelation equals the direct time-domain sum."""
    from scipy.linalg import solve_toeplitz
    from oracle import lpc_oracle as lo
    from speaker_recognition_amd import synth
    ex = lo.LPCExtractor(16000)
    pcm = synth.synth_speech(2, 0.6, 16000).astype(float)
    feat = ex.extract(pcm)
    assert feat.shape == ((len(pcm) - 512) // 256 + 1, 15)
    for f in (0, 7, 20):
        frame = pcm[f * 256:f * 256 + 512] * ex.window
        frame[1:] -= frame[:-1] * 0.95
        r = lo.acorr_lpc(frame)
        direct = np.array([np.dot(frame[:512 - k], frame[k:]) for k in range(16)]) / 512
        assert np.max(np.abs(direct - r[:16])) < 1e-9 * abs(r[0])
        a = solve_toeplitz(r[:15], -r[1:16])
        assert np.max(np.abs(a - feat[f])) < 1e-8
    silent = ex.extract(np.zeros(4000))
    assert silent.shape[1] == 15 and np.all(silent == 0)          # NaN -> 0, LPC.py:56


def test_ltsd_oracle_properties():
    """The LTSD restatement (parity unpinned -- third-party algorithm): window sizes the reference
    uses, the edge rule, scale invariance of the measure when signal and noise scale together, and
    the decision rule of the product's host side (pure Python, no device)."""
    from oracle import ltsd_oracle as lo
    assert lo.window_size(16000) == 743 and lo.window_size(8000) == 371
    assert lo.num_windows(743 + 371, 743) == 2 and lo.num_windows(100, 743) == 0
    rng = np.random.default_rng(0)
    noise = rng.normal(0, 100, 16000)
    sig = rng.normal(0, 100, 16000)
    t = np.arange(4000) / 16000.0
    sig[6000:10000] += 4000 * np.sin(2 * np.pi * 440 * t)
    na, lam0, lam1 = lo.thresholds(noise, 743)
    l = lo.ltsd(sig, na, 743)
    assert np.all(l[:5] == 0) and np.all(l[-5:] == 0)
    assert l[(6000 // 371) + 2] > lam1 and l[7] < lam0
    l2 = lo.ltsd(3.0 * sig, 3.0 * na, 743)
    assert np.allclose(l, l2, atol=1e-9)
    from speaker_recognition_amd.filters.ltsd import voiced_runs
    assert voiced_runs(np.array([0, 3, 3, 9, 3, 0, 3, 3, 0, 9.0]), 2.0, 5.0) == [(1, 4), (9, 9)]
    assert voiced_runs(np.array([]), 1.0, 2.0) == []


def test_glibc_rand_restatement_matches_libc():
    """oracle/init_oracle.py carries glibc's rand() (the reference seeds every draw of its trainer from it,
    random.hh:22-25): equal to the C library's own, from the default seed of a fresh process."""
    import subprocess
    import sys
    from oracle import init_oracle as io
    code = "import ctypes as C; l = C.CDLL('libc.so.6'); print(' '.join(str(l.rand()) for _ in range(3000)))"
    want = [int(v) for v in subprocess.check_output([sys.executable, "-c", code]).split()]
    r = io.GlibcRand()
    assert [r() for _ in range(3000)] == want


def test_init_oracle_matches_reference_trainer(oracle_built):
    """Training FROM SCRATCH: the numpy restatement of the reference's initialisers (K random frames; k-means|| +
    weighted k-means++ + Lloyd) with its random streams, then the oracle's EM iterations under the reference's stop
    rule, against the models its compiled trainer produced in fresh processes (tests/golden/make_init_golden.py) --
    also after a load() whose 32 Gaussians each consumed a draw."""
    import hashlib
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from make_init_golden import CASES, case_data
    from oracle import init_oracle as io
    go = oracle_built
    g = np.load(os.path.join(ROOT, "tests", "golden", "init_golden.npz"))
    for name, K, D, n, iters, km, conc, preload in CASES:
        X = case_data(K, D, n, 300 + K + D)
        assert hashlib.sha256(X.astype(np.float32).tobytes()).hexdigest() == str(g[name + "_X_sha256"])
        rand = io.GlibcRand()
        if preload:
            for _ in range(32):
                rand()                                 # GMM::load: one Random per Gaussian (gmm.cc:671-676, gmm.hh:44)
        w, mu, sg = io.init_gaussians(X, K, km, conc, rand)
        p = go.GMMParams(w, mu, sg)
        last = -np.finfo(np.float64).max
        for it in range(iters):                        # GMMTrainerBaseline::train, gmm.cc:619-650
            p = go.em_iteration(p, X)
            if it % 2 == 0:
                continue
            ll = go.score_all(p, X)
            if abs(ll - last) / abs(ll) < 0.01 and ll - last < 0.01:
                break
            last = ll
        assert np.max(np.abs(p.weights - g[name + "_w"])) < 2e-6, name
        assert np.max(np.abs(p.mean - g[name + "_mean"])) < 2e-5, (name, np.max(np.abs(p.mean - g[name + "_mean"])))
        assert np.max(np.abs(p.sigma - g[name + "_sigma"]) / g[name + "_sigma"]) < 2e-5, name


def test_oracle_zero_weight_mixture_adds_nothing(oracle_built):
    """A mixture of weight 0 contributes exactly 0 in the reference (w_k p_k in the linear domain, gmm.cc:237-244): the
    log-sum-exp mode of the oracle agrees with its reference-faithful modes also when the frame sits ON that mixture
    (it used safe_log's ln 1e-15 as the weight; found by scripts/debug/fuzz_generic.py)."""
    go = oracle_built
    rng = np.random.default_rng(3)
    K, D = 6, 5
    w = np.full(K, 1.0 / (K - 1))
    w[2] = 0.0
    mu = rng.normal(0, 3, (K, D))
    sg = rng.uniform(0.5, 1.0, (K, D))
    X = np.vstack([mu[2] + 0.01, mu[0] + 0.3, rng.normal(0, 1, D)])
    p = go.GMMParams(w, mu, sg)
    a = go.score_batch(p, X, go.MODE_LOGSUMEXP)
    b = go.score_batch(p, X, go.MODE_FASTEXP)
    c = go.score_batch(p, X, 1)
    assert np.max(np.abs(a - c)) < 1e-9 and np.max(np.abs(a - b) / np.maximum(1, np.abs(b))) < 1e-4
