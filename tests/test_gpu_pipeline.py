"""GPU end-to-end: BASELINE configs[0] (enroll + predict through the CLI surface on synthetic
16 kHz WAVs), EM / MAP single-iteration parity against the oracle, the fused serving step."""
import os

import numpy as np
import pytest
from scipy.io import wavfile

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("D", [13, 84, 200, 300])
def test_em_and_map_iteration_vs_oracle(built_lib, oracle_built, D):
    """One EM iteration and one MAP iteration from identical starting parameters (13 dims; the wide rows of
    MFCC + LPC with both deltas: 84 dims, vector-ALU engine and the looped statistics roles; 200 and 300 dims: rows
    wider than a lane's registers, the D-sliced kernels -- the reference has no limit, gmm.cc:40-51)."""
    from speaker_recognition_amd import synth
    from speaker_recognition_amd.pygmm import GMM
    go = oracle_built
    true = synth.synth_gmm(8, D, 3)
    X = synth.draw_frames(true, 4000, 9)
    rng = np.random.default_rng(1)
    start = go.GMMParams(np.full(8, 1 / 8), true[1] + 0.2 * rng.standard_normal(true[1].shape),
                         np.full_like(true[2], 0.9))
    want = go.em_iteration(start, X.astype(np.float64))
    g = GMM.from_arrays(start.weights, start.mean, start.sigma)
    g.nr_iteration, g.init_with_kmeans = 1, -1            # -1: warm start (extension)
    assert g.fit(X) == 1
    w, mu, sg = g.params()
    assert np.max(np.abs(w - want.weights)) < 1e-5
    assert np.max(np.abs(mu - want.mean)) < 1e-4
    assert np.max(np.abs(sg - want.sigma) / want.sigma) < 1e-3
    # MAP: means only, relevance 16, weights/sigmas copied (gmmubm.cc:29-81)
    ubm = GMM.from_arrays(start.weights, start.mean, start.sigma)
    want_map = go.em_iteration(start, X.astype(np.float64), map_relevance=16.0, ubm=start)
    spk = GMM(8, nr_iteration=1)
    assert spk.fit(X[:300], ubm=ubm) == 1
    want_map = go.em_iteration(start, X[:300].astype(np.float64), map_relevance=16.0, ubm=start)
    w2, mu2, sg2 = spk.params()
    assert np.array_equal(w2, start.weights) and np.array_equal(sg2, start.sigma)
    assert np.max(np.abs(mu2 - want_map.mean)) < 1e-4


def test_em_mixtures_without_support_follow_the_reference(built_lib, oracle_built):
    """More mixtures than the data supports (23 mixtures, 128 dims, 50 frames): responsibilities fall below fp32's range
    for most (frame, mixture) pairs.  The reference's float64 keeps them down to DBL_MIN, so such a mixture's mean moves
    to the weighted mean of the frames; the device's sums are 0 there and are redone on the host in float64 (em.hip).
    One EM iteration against the oracle (found by scripts/debug/fuzz_train.py)."""
    from speaker_recognition_amd import synth
    from speaker_recognition_amd.pygmm import GMM
    go = oracle_built
    rng = np.random.default_rng(10)
    K, D, n = 23, 128, 50
    true = synth.synth_gmm(K, D, 11)
    X = synth.draw_frames(true, n, 12)
    start = go.GMMParams(np.full(K, 1.0 / K), true[1] + 0.2 * rng.standard_normal(true[1].shape), np.full_like(true[2], 0.9))
    want = go.em_iteration(start, X.astype(np.float64))
    g = GMM.from_arrays(start.weights, start.mean, start.sigma)
    g.nr_iteration, g.init_with_kmeans = 1, -1
    assert g.fit(X) == 1
    w, mu, sg = g.params()
    assert np.max(np.abs(w - want.weights)) < 2e-5
    assert np.max(np.abs(mu - want.mean)) < 3e-4, np.max(np.abs(mu - want.mean))
    assert np.max(np.abs(sg - want.sigma) / want.sigma) < 2e-3


def test_em_converges_and_improves(built_lib):
    from speaker_recognition_amd import synth
    from speaker_recognition_amd.pygmm import GMM
    true = synth.synth_gmm(6, 10, 21)
    X = synth.draw_frames(true, 6000, 4)
    ll_true = GMM.from_arrays(*true).score_all(X)
    for km in (0, 1):
        g = GMM(6, nr_iteration=40, init_with_kmeans=km, seed=5)
        n_it = g.fit(X)
        assert 2 <= n_it <= 40
        assert g.get_dim() == 10 and g.get_nr_mixtures() == 6
        ll = g.score_all(X)
        assert ll > ll_true - 0.05 * abs(ll_true), (km, ll, ll_true)   # reaches the generating model's fit
        w, mu, sg = g.params()
        assert abs(w.sum() - 1) < 1e-9 and np.all(sg >= np.sqrt(1e-3) - 1e-12)


def test_cfg0_cli_enroll_predict(built_lib, tmp_path, capsys):
    """BASELINE configs[0], shortened to 4 speakers x 6 s so it stays a unit test: WAVs on disk ->
    speaker-recognition.py enroll -> model file -> predict; every clip recognised, and the device
    decision equals the CPU restatement (oracle MFCC + oracle scoring of the SAME trained models)."""
    from oracle import gmm_oracle as go, mfcc_oracle as mo
    from speaker_recognition_amd import cli, synth
    from speaker_recognition_amd.interface import ModelInterface
    fs = 16000
    for s in (0, 7, 14, 21):      # synthetic voices far enough apart to be separable by 13 MFCCs
        d = tmp_path / ("spk%d" % s)
        d.mkdir()
        wavfile.write(str(d / "enroll.wav"), fs, synth.synth_speech(s, 6.0, fs, seed=1000 + s))
        wavfile.write(str(tmp_path / ("test_spk%d.wav" % s)), fs, synth.synth_speech(s, 3.0, fs, seed=2000 + s))
    model = str(tmp_path / "model.out")
    cli.main(["-t", "enroll", "-i", str(tmp_path / "spk*"), "-m", model, "--mixtures", "16",
              "--win-length-ms", "25", "--win-shift-ms", "10", "--seed", "3", "--no-lpc"])
    assert os.path.getsize(model) > 1000
    args = cli.get_args(["-t", "predict", "-i", str(tmp_path / "test_*.wav"), "-m", model])
    res = cli.task_predict(args.input, args.model)
    assert len(res) == 4
    for f, label in res:
        assert os.path.basename(f).replace("test_", "").replace(".wav", "") == label
    # CPU restatement with the same models
    m = ModelInterface.load(model)
    kw = dict(win_length_ms=25, win_shift_ms=10)
    params = [go.GMMParams(*g.params()) for g in m.gmmset.gmms]
    for f, label in res:
        _, sig = wavfile.read(f)
        feat = mo.extract(fs, sig, **kw)
        scores = [go.score_all(p, feat) / len(feat) for p in params]
        assert m.gmmset.y[int(np.argmax(scores))] == label
        dev = np.array(m.gmmset.predict_one_scores(m._features(fs, sig))) / len(feat)
        assert np.max(np.abs(dev - np.array(scores)) / np.abs(scores)) < 2e-3


def test_fused_serving_step_equals_staged(built_lib):
    from speaker_recognition_amd import synth
    from speaker_recognition_amd.core import Batch, MfccExtractor, ModelSet
    from speaker_recognition_amd.pygmm import GMM
    fs = 16000
    kw = dict(win_length_ms=25, win_shift_ms=10)
    pcm = Batch.from_pcm([synth.synth_speech(s, 1.5 + 0.3 * s, fs) for s in range(5)])
    ex = MfccExtractor(fs, **kw)
    ms = ModelSet([GMM.from_arrays(*synth.synth_gmm(32, 39, 7 + s)) for s in range(9)])
    sums, arg = ex.predict_batch(ms, pcm, nd=2)
    feats = ex.extract_batch(pcm, nd=2)
    sums2, arg2 = ms.score(feats)
    assert np.array_equal(sums, sums2) and np.array_equal(arg, arg2)
    sums3, arg3 = ex.predict_batch(ms, pcm, nd=2)                     # workspace reuse is stable
    assert np.array_equal(sums, sums3)


def test_gmmset_rejection_and_pickle(built_lib, tmp_path):
    from speaker_recognition_amd import synth
    from speaker_recognition_amd.gmmset import GMMSetPyGMM
    from speaker_recognition_amd.pygmm import GMM
    ubm_p = synth.synth_gmm(8, 13, 77)
    ubm = GMM.from_arrays(*ubm_p)
    gs = GMMSetPyGMM(ubm=ubm, nr_iteration=3, seed=1)
    assert gs.gmm_order == 8
    spk = [synth.synth_map_speaker(ubm_p, 300 + s) for s in range(3)]
    for s, p in enumerate(spk):
        gs.fit_new(synth.draw_frames(p, 1500, 600 + s), "s%d" % s)
    tests = [synth.draw_frames(p, 400, 800 + s) for s, p in enumerate(spk)]
    assert gs.predict(tests) == ["s0", "s1", "s2"]
    assert [gs.predict_one(t) for t in tests] == ["s0", "s1", "s2"]
    gs.reject_threshold = 1e9
    assert gs.predict_one_with_rejection(tests[0]) is None
    gs.reject_threshold = -1e9
    assert gs.predict_one_with_rejection(tests[0]) == "s0"
    import pickle
    gs.before_pickle()
    blob = pickle.dumps(gs)
    gs.after_pickle()
    gs2 = pickle.loads(blob)
    gs2.after_pickle()
    assert gs2.predict(tests) == ["s0", "s1", "s2"]


def test_per_utterance_loop_reuses_its_batch_and_equals_the_batched_pass(built_lib):
    """The reference's drivers score one utterance per call (gmmset.py:62-64, :95-99).  Here such a loop refills ONE device batch
    (sr_batch_reset_features; a stale tile table is rebuilt in its own buffers): utterances of different lengths -- growing, shrinking,
    a tile-boundary length, a single frame, repeated lengths -- and then a different dimension through the same handle must give the
    bits of the one-pass batch and of a fresh batch per utterance."""
    from speaker_recognition_amd import synth
    from speaker_recognition_amd.core import Batch, ModelSet
    from speaker_recognition_amd.gmmset import GMMSet
    from speaker_recognition_amd.pygmm import GMM
    rng = np.random.default_rng(5)
    raw = [synth.synth_gmm(16, 13, 40 + s) for s in range(5)]
    gs = GMMSet(gmm_order=16)
    for s, m in enumerate(raw):
        gs._append("s%d" % s, GMM.from_arrays(*m))
    lens = [300, 77, 1000, 32, 33, 1, 300, 300, 512, 5]
    utts = [synth.draw_frames(raw[i % 5], n, 70 + i).astype(np.float64) for i, n in enumerate(lens)]
    ms = ModelSet(gs.gmms)
    want_sums, want_arg = ms.score(Batch.from_features(utts))
    for rep in range(2):                                  # the second sweep starts from the last (short) shape
        for i, x in enumerate(utts):
            got = np.asarray(gs.predict_one_scores(x))
            fresh, _ = ms.score(Batch.from_features([x]))
            assert np.array_equal(got, fresh[0]) and np.array_equal(got, want_sums[i]), (rep, i)
            assert gs.predict_one(x) == "s%d" % want_arg[i]
    assert len(gs._scratch) == 1
    # the same handle with another dimension and layout
    b = gs._scratch[next(iter(gs._scratch))]
    raw39 = [synth.synth_gmm(8, 39, 90 + s) for s in range(3)]
    ms39 = ModelSet([GMM.from_arrays(*m) for m in raw39])
    x39 = synth.draw_frames(raw39[1], 450, 3)
    b.reset_features(x39)
    a, _ = ms39.score(b)
    c, _ = ms39.score(Batch.from_features([x39]))
    assert b.dim == 39 and b.n_rows == 450 and np.array_equal(a, c)
    with pytest.raises(Exception):
        Batch.from_pcm([np.zeros(4000, np.int16)]).reset_features(x39)       # a PCM batch is not refilled with features
    # several utterances per refill, through the C ABI itself: ragged, an empty one in the middle, then fewer and longer ones, then a
    # refill too large for the page-locked path (> 4 MB: upload + wait) -- each against a fresh batch of the same layout
    from speaker_recognition_amd import _lib
    L = _lib.lib()
    for lens in ([40, 0, 333, 64], [700, 5], [40, 0, 333, 64], [30000]):
        mats = [synth.draw_frames(raw39[i % 3], n, 200 + i + len(lens)) for i, n in enumerate(lens)]
        X = np.ascontiguousarray(np.concatenate(mats))
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        assert L.sr_batch_reset_features(b._h, _lib.as_fp(X), X.shape[0], 39, _lib.as_i64p(off), len(lens)) == 0, _lib.last_error()
        a, aa = ms39.score(b)
        c, ca = ms39.score(Batch.from_features(mats))
        assert b.n_utt == len(lens) and np.array_equal(a, c) and np.array_equal(aa, ca), lens
    bad = np.array([0, 10, 5], dtype=np.int64)
    assert L.sr_batch_reset_features(b._h, _lib.as_fp(X), 5, 39, _lib.as_i64p(bad), 2) != 0 and "non-decreasing" in _lib.last_error()


def test_score_models_entry_point_equals_the_packed_set(built_lib):
    """sr_score_models_f32 in-process (what a fork helper runs for a forked worker's predict_one): GMM handles in, the sums of one fused
    pass out -- the bits of ModelSet.score on the same models; the packed set is kept across calls and follows a model whose
    parameters change behind the same handle, a shorter list, and a list that names a handle twice."""
    import ctypes as C
    from speaker_recognition_amd import _lib, synth
    from speaker_recognition_amd.core import Batch, ModelSet
    from speaker_recognition_amd.pygmm import GMM
    L = _lib.lib()
    raw = [synth.synth_gmm(16, 20, 610 + s) for s in range(9)]
    gm = [GMM.from_arrays(*m) for m in raw]

    def entry(models, x):
        X = _lib.f32_matrix(x)
        h = (C.c_void_p * len(models))(*[g.gmm for g in models])
        out = np.zeros(len(models))
        assert L.sr_score_models_f32(h, len(models), _lib.as_fp(X), X.shape[0], X.shape[1], _lib.as_dp(out), _lib.SR_CLAMP_COMPAT) == 0, _lib.last_error()
        return out

    xs = [synth.draw_frames(raw[i % 9], n, 77 + i, outlier_frac=0.01) for i, n in enumerate([500, 31, 500, 1200])]
    for models in (gm, gm[:4], [gm[2], gm[5], gm[2]], gm):
        ms = ModelSet(models)
        for x in xs:
            want, _ = ms.score(Batch.from_features([x]))
            assert np.array_equal(entry(models, x), want[0])
    gm[1].fit(xs[3], None)                                     # new parameters behind the same handle
    want, _ = ModelSet(gm).score(Batch.from_features([xs[0]]))
    assert np.array_equal(entry(gm, xs[0]), want[0])
    wrong = synth.draw_frames(synth.synth_gmm(4, 13, 1), 50, 2)
    X = _lib.f32_matrix(wrong)
    h = (C.c_void_p * 2)(gm[0].gmm, gm[1].gmm)
    assert L.sr_score_models_f32(h, 2, _lib.as_fp(X), 50, 13, _lib.as_dp(np.zeros(2)), 0) != 0 and "dim" in _lib.last_error()


def test_serving_loop_with_changing_layouts_equals_fresh_batches(built_lib):
    """A serving loop whose batch changes its layout from call to call (Batch.reset_pcm: 1-4 utterances of 0.4-1.2 s): samples,
    offsets, the feature stage's frame offsets and the rebuilt tile tables all travel through page-locked copies left in flight
    (common.hpp: StagedUpload) -- every decision must carry the bits of a fresh batch, including right after a larger and a
    smaller layout, an unchanged layout with other samples, and a tile-boundary length."""
    from speaker_recognition_amd import synth
    from speaker_recognition_amd.core import Batch, MfccExtractor, ModelSet
    from speaker_recognition_amd.pygmm import GMM
    fs = 16000
    ex = MfccExtractor(fs, win_length_ms=25, win_shift_ms=10)
    ubm = synth.synth_gmm(64, 39, 99)
    ms = ModelSet([GMM.from_arrays(*m) for m in [ubm] + [synth.synth_map_speaker(ubm, 500 + s) for s in range(14)]])     # shared sigma
    ms2 = ModelSet([GMM.from_arrays(*synth.synth_gmm(32, 39, 7 + s)) for s in range(5)])                                   # generic engine
    clips = [synth.synth_speech(u, 1.3, fs, seed=40 + u) for u in range(4)]
    rng = np.random.default_rng(3)
    n34 = (34 - 1) * ex.FRAME_SHIFT + ex.FRAME_LEN            # 34 raw frames -> 32 after the deltas: exactly one tile
    layouts = [[6400], [19000, 7000], [n34], [12000, 12000, 12000, 12000], [6400], [6400], [n34, n34], [20000]]
    batch = Batch.from_pcm([clips[0][:8000]])
    for rep in range(2):
        for li, lens in enumerate(layouts):
            sigs = [clips[(u + li + rep) % 4][:n] for u, n in enumerate(lens)]
            batch.reset_pcm(sigs)
            for models in (ms, ms2):
                got = ex.predict_batch(models, batch, nd=2)
                want = ex.predict_batch(models, Batch.from_pcm(sigs), nd=2)
                assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), (rep, li)
    assert batch.n_utt == 1 and batch.n_rows == 20000


def test_map_training_vs_reference_dso_golden(built_lib, gmm_golden):
    """train_model_from_ubm on the GPU (legacy symbol, double** rows) against the models the
    reference's own compiled trainer produced for the same UBM and frames (1 and 4 iterations)."""
    import ctypes as C
    from speaker_recognition_amd._lib import Parameter
    from speaker_recognition_amd.pygmm import GMM
    L, g = built_lib, gmm_golden
    ubm = GMM.from_arrays(g["syn16x13_w"], g["syn16x13_mean"], g["syn16x13_sigma"])
    X = np.ascontiguousarray(g["map_X"])
    n, d = X.shape
    rows = (C.POINTER(C.c_double) * n)(*[C.cast(X[i].ctypes.data, C.POINTER(C.c_double)) for i in range(n)])
    for iters in (1, 4):
        spk = GMM(16)
        p = Parameter(nr_instance=n, nr_dim=d, nr_mixture=16, min_covar=1e-3, threshold=0.01,
                      nr_iteration=iters, init_with_kmeans=0, concurrency=4, verbosity=0)
        L.train_model_from_ubm(spk.gmm, ubm.gmm, rows, C.byref(p))
        w, mu, sg = spk.params()
        ref = g["map%d_mean" % iters]
        assert np.max(np.abs(mu - ref) / np.maximum(1.0, np.abs(ref))) < 5e-5, (iters, np.max(np.abs(mu - ref)))
        assert np.array_equal(w, g["syn16x13_w"]) and np.array_equal(sg, g["syn16x13_sigma"])


def test_em_training_vs_reference_trainer_golden(built_lib, gmm_golden):
    """train_model's engine on the GPU, warm-started at the reference trainer's own initial state,
    against the models the reference produced after 1, 2 and 6 iterations (stop rule included)."""
    from speaker_recognition_amd.pygmm import GMM
    g = gmm_golden
    X = g["em_X"]
    sig = np.sqrt(((X - X.mean(0)) ** 2).sum(0) / (len(X) - 1))
    for iters in (1, 2, 6):
        m = GMM.from_arrays(np.full(8, 1.0 / 8), X[g["em_init_rows"]], np.tile(sig, (8, 1)))
        m.nr_iteration, m.init_with_kmeans = iters, -1          # -1: start from the handle's parameters
        assert m.fit(X) == iters
        w, mu, sg = m.params()
        assert np.max(np.abs(w - g["em%d_w" % iters])) < 1e-5, iters
        assert np.max(np.abs(mu - g["em%d_mean" % iters])) < 1e-4, (iters, np.max(np.abs(mu - g["em%d_mean" % iters])))
        assert np.max(np.abs(sg - g["em%d_sigma" % iters]) / g["em%d_sigma" % iters]) < 5e-4, iters


def test_train_from_scratch_vs_reference_trainer_golden(built_lib, tmp_path):
    """train_model FROM SCRATCH through the legacy symbol, in a fresh process as the goldens were made with the
    reference's compiled library (tests/golden/make_init_golden.py): both of the reference's initialisers -- K random
    frames and k-means|| (oversampling rounds, weighted k-means++, weighted Lloyd, Lloyd on the full data) -- draw
    the reference's own random numbers (libc rand() and the engines it seeds), also after a load() whose Gaussians
    consumed 32 draws; the models after EM agree to the digits the text format keeps."""
    import hashlib
    import subprocess
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from make_init_golden import case_data
    from oracle import gmm_oracle as go
    g = np.load(os.path.join(ROOT, "tests", "golden", "init_golden.npz"))
    pre = tmp_path / "preload.model"
    pre.write_text(str(g["preload_text"]))
    helper = os.path.join(ROOT, "tests", "golden", "_train_proc.py")
    for name in g["cases"]:
        K, iters, km, conc, preload = (int(v) for v in g[name + "_args"])
        want_mean = g[name + "_mean"]
        D = want_mean.shape[1]
        n = {"rand8x13": 3000, "km8x13": 3000, "km16x20_after_load": 5000, "rand5x39_after_load": 2000, "km32x39": 20000}[str(name)]
        X = case_data(K, D, n, 300 + K + D)
        assert hashlib.sha256(X.astype(np.float32).tobytes()).hexdigest() == str(g[name + "_X_sha256"]), "input regenerated differently"
        xp, out = str(tmp_path / "X.npy"), str(tmp_path / ("%s.model" % name))
        np.save(xp, X)
        cmd = [sys.executable, helper, "--lib", "hip", xp, str(K), str(iters), str(km), str(conc), out]
        if preload:
            cmd.append(str(pre))
        subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL)
        got = go.parse_model_text(open(out).read())
        assert np.max(np.abs(got.weights - g[name + "_w"])) < 2e-5, name
        assert np.max(np.abs(got.mean - want_mean)) < 2e-4, (name, np.max(np.abs(got.mean - want_mean)))
        assert np.max(np.abs(got.sigma - g[name + "_sigma"]) / g[name + "_sigma"]) < 1e-3, name


def test_kmeans_initialiser_on_the_device_equals_its_restatement(built_lib):
    """k-means|| + weighted k-means++ + Lloyd with the nearest-centre pass and (round 3) the cluster sums of every Lloyd
    step on the device: a stable sort of the points by (worker block, cluster), sequential segment sums, block-ordered
    totals -- the additions the reference makes (kmeans.cc:72-107, :189-212).  Against the independent numpy
    restatement of the same initialiser and random streams (oracle/init_oracle.py, pinned on the reference trainer's
    goldens) with MANY worker blocks of odd sizes and clusters that are empty in most of them; float64 on both sides,
    same order: the centres agree to rounding."""
    from oracle import init_oracle as io
    from speaker_recognition_amd.pygmm import GMM
    rng = np.random.default_rng(3)
    for n, K, D, conc, seed in ((30011, 96, 13, 37, 11), (9001, 40, 39, 64, 5), (5000, 8, 20, 3, 2), (3001, 20, 200, 5, 4)):
        cent = rng.normal(0, 3, (K, D))
        X = (cent[rng.integers(0, K, n)] + rng.normal(0, 1, (n, D))).astype(np.float32)
        g = GMM(nr_mixture=K, nr_iteration=0, init_with_kmeans=1, seed=seed, concurrency=conc)
        g.fit(X)
        w, mu, sg = g.params()
        w0, mu0, sg0 = io.init_gaussians(X.astype(np.float64), K, 1, conc, io.GlibcRand(seed + 1))
        assert np.max(np.abs(mu - mu0)) < 1e-9 * max(1.0, np.max(np.abs(mu0))), (n, K, D, np.max(np.abs(mu - mu0)))
        assert np.max(np.abs(sg - sg0) / sg0) < 1e-12 and np.allclose(w, w0, atol=0)


def test_em_statistics_engines_vs_oracle(built_lib, oracle_built):
    """One EM iteration from identical starting parameters through both statistics engines -- the fp64 matrix cores
    (em_stats_mfma_kernel: sums about the origin from fp32-exact operands, re-centred in the float64 M-step; dims <= 40)
    and the vector ALU (em_stats_kernel: per-mixture centring, fp32 slabs; every dim) -- against the float64 oracle, on
    data FAR FROM THE ORIGIN (where sums about the origin cancel the most): both inside the gate, the float64 accumulation
    at least as close as the fp32 one."""
    from speaker_recognition_amd import _lib
    from speaker_recognition_amd.pygmm import GMM
    go = oracle_built
    rng = np.random.default_rng(8)
    try:
        for n, K, D in ((20000, 40, 13), (12000, 70, 39), (6000, 17, 26)):
            cent = 30.0 + rng.normal(0, 2, (K, D))
            X = (cent[rng.integers(0, K, n)] + rng.normal(0, 0.7, (n, D))).astype(np.float32)
            r6 = np.vectorize(lambda v: float("%g" % v))
            start = go.GMMParams(np.full(K, 1.0 / K), r6(cent + 0.2 * rng.standard_normal(cent.shape)), np.full((K, D), 0.9))
            want = go.em_iteration(start, X.astype(np.float64))
            err = {}
            for eng in (1, 3):             # 3: automatic among the iteration-at-a-time engines (0 would take the 17 x 26 fit whole, em_small.hip)
                _lib.set_option("em_stats_engine", eng)
                g = GMM.from_arrays(start.weights, start.mean, start.sigma)
                g.nr_iteration, g.init_with_kmeans = 1, -1            # -1: warm start (extension)
                assert g.fit(X) == 1
                w, mu, sg = g.params()
                err[eng] = (np.max(np.abs(w - want.weights)), np.max(np.abs(mu - want.mean)), np.max(np.abs(sg - want.sigma) / want.sigma))
                assert err[eng][0] < 1e-5 and err[eng][1] < 1e-4 and err[eng][2] < 1e-3, (n, K, D, eng, err[eng])
            assert err[3][2] <= err[1][2] + 1e-6 and err[3][1] <= err[1][1] + 1e-6, (n, K, D, err)
    finally:
        _lib.set_option("em_stats_engine", 0)


def test_em_responsibilities_on_the_matrix_cores_vs_oracle(built_lib, oracle_built):
    """Round 4: em_stats_split_kernel -- log2 densities of the E-step from the scoring engine's split-bf16 contraction
    (32-mixture tiles, 128 mixtures per workgroup), responsibilities through LDS into the fp64 statistics -- taken when the
    model is inside that layout's range.  One EM iteration against the float64 oracle and against round 3's kernel
    (responsibilities on the vector ALU) from the same start: models of 1, 2, 4 and 5 tiles (a last workgroup with one live
    tile; padding mixtures inside a tile), dims with and without padding, a frame count that is no multiple of the tile."""
    from speaker_recognition_amd import _lib
    from speaker_recognition_amd.pygmm import GMM
    go = oracle_built
    rng = np.random.default_rng(18)
    try:
        for n, K, D in ((9001, 160, 39), (5000, 56, 13), (7013, 128, 20), (3000, 32, 26)):
            cent = 3.0 + rng.normal(0, 2, (K, D))
            X = (cent[rng.integers(0, K, n)] + rng.normal(0, 0.7, (n, D))).astype(np.float32)
            r6 = np.vectorize(lambda v: float("%g" % v))
            start = go.GMMParams(np.full(K, 1.0 / K), r6(cent + 0.2 * rng.standard_normal(cent.shape)), np.full((K, D), 0.9))
            want = go.em_iteration(start, X.astype(np.float64))
            got = {}
            for eng, ran in ((2, 2), (3, 3)):
                _lib.set_option("em_stats_engine", eng)
                g = GMM.from_arrays(start.weights, start.mean, start.sigma)
                g.nr_iteration, g.init_with_kmeans = 1, -1            # -1: warm start (extension)
                assert g.fit(X) == 1
                assert _lib.last_em_stats_engine() == ran, (n, K, D, eng, _lib.last_em_stats_engine())
                w, mu, sg = got[eng] = g.params()
                err = (np.max(np.abs(w - want.weights)), np.max(np.abs(mu - want.mean)), np.max(np.abs(sg - want.sigma) / want.sigma))
                assert err[0] < 1e-5 and err[1] < 1e-4 and err[2] < 1e-3, (n, K, D, eng, err)
            # the two kernels differ in the rounding of the log densities only
            assert np.max(np.abs(got[3][1] - got[2][1])) < 2e-5 and np.max(np.abs(got[3][2] - got[2][2]) / got[2][2]) < 2e-5, (n, K, D)
        # frames whose likelihood underflows carry no responsibility (gmm.cc:482-498), and fewer frames than one 128-frame tile:
        # a means-only MAP enrolment (gmmubm.cc:29-81) of 70 frames, five of them 1000 units away, from a 64-mixture UBM
        K, D = 64, 39
        cent = rng.normal(0, 2, (K, D))
        start = go.GMMParams(np.full(K, 1.0 / K), np.vectorize(lambda v: float("%g" % v))(cent), np.full((K, D), 0.9))
        X = (cent[rng.integers(0, K, 70)] + rng.normal(0, 0.7, (70, D))).astype(np.float32)
        X[::14] += 1000.0
        want = go.em_iteration(start, X.astype(np.float64), map_relevance=16.0, ubm=start)
        ubm = GMM.from_arrays(start.weights, start.mean, start.sigma)
        for eng, ran in ((2, 2), (3, 3)):
            _lib.set_option("em_stats_engine", eng)
            spk = GMM(K, nr_iteration=1)
            assert spk.fit(X, ubm=ubm) == 1 and _lib.last_em_stats_engine() == ran
            w, mu, sg = spk.params()
            assert np.array_equal(w, start.weights) and np.array_equal(sg, start.sigma)
            assert np.max(np.abs(mu - want.mean)) < 1e-4, (eng, np.max(np.abs(mu - want.mean)))
    finally:
        _lib.set_option("em_stats_engine", 0)


def test_band_frames_through_fused_pipelined_and_streaming_paths(built_lib, oracle_built):
    """Frames whose log-likelihood sits in the band where the reference's partial-product flushes decide on
    every path that scores from PCM: the fused serving step, its chunk-pipelined form (the band of ANY chunk sends the batch
    back through one pass), the double-buffered serving stream (resolved at collect) and the one-process multi-slot
    predictor -- all equal to the reference's arithmetic (oracle mode 0) on the device's own features.  The models sit
    37 sigma from the features in ONE tight dimension, so that a few per cent of the frames land in the band, others beyond
    it (clamped) and the rest before it."""
    from speaker_recognition_amd import _lib, synth
    from speaker_recognition_amd.core import Batch, MfccExtractor, ModelSet, MultiPredictor, ServingStream
    from speaker_recognition_amd.pygmm import GMM
    go = oracle_built
    fs, n_win, win = 16000, 16, 32000
    ex = MfccExtractor(fs)
    pcm = np.stack([synth.synth_speech(3 + u, 2.0, fs)[:win] for u in range(n_win)])
    feats = ex.extract_batch(Batch.from_pcm(list(pcm)), nd=0)
    X, off = feats.download().astype(np.float64), feats.offsets()
    D, K = X.shape[1], 32
    rng = np.random.default_rng(2)
    models = []
    for s in range(3):
        mean = np.zeros((K, D))
        mean[:, s] = 1.84 + 0.01 * rng.standard_normal(K)       # 36.8 sigma from a feature value of 0; the tight dimension differs per model
        sigma = np.full((K, D), 3.0)
        sigma[:, s] = 0.05
        r6 = np.vectorize(lambda v: float("%g" % v))
        models.append((np.full(K, 1.0 / K), r6(mean), sigma))
    want_ll = np.stack([go.score_batch(go.GMMParams(*m), X, go.MODE_FASTEXP) for m in models])
    band = (want_ll < -600.0) & (want_ll > -709.0)
    assert band.sum() > 20 and (want_ll == go.LN_1E_15).sum() > 20, (band.sum(), (want_ll == go.LN_1E_15).sum())
    want = np.array([[want_ll[s][off[u]:off[u + 1]].sum() for s in range(3)] for u in range(n_win)])
    gm = [GMM.from_arrays(*m) for m in models]
    ms = ModelSet(gm)
    ok = lambda sums: np.max(np.abs(sums - want) / np.maximum(1.0, np.abs(want))) < 1e-4
    calls0 = _lib.flush_stats()[0]
    batch = Batch.from_pcm(list(pcm))
    s_fused, a_fused = ex.predict_batch(ms, batch, nd=0)
    assert ok(s_fused) and np.array_equal(a_fused, np.argmax(want, axis=1))
    s_again, a_again = ex.predict_batch(ms, batch, nd=0)
    assert np.array_equal(s_again, s_fused) and np.array_equal(a_again, a_fused)
    # the band list overflowing (capacity 1) on the paths that complete HOST copies of the results: the fused step (results
    # delivered into page-locked memory by the pass itself, round 6) and the multi-slot predictor's pieces
    _lib.set_option("flush_list_cap", 1)
    try:
        s_cap, a_cap = ex.predict_batch(ms, batch, nd=0)
        assert np.array_equal(s_cap, s_fused) and np.array_equal(a_cap, a_fused)
        s_mc, a_mc = MultiPredictor(gm, fs, n_slots=2).predict(list(pcm), nd=0)
        assert np.array_equal(s_mc, s_fused) and np.array_equal(a_mc, a_fused)
    finally:
        _lib.set_option("flush_list_cap", 0)
    # (graph, delay): with a delay the host is held back in front of the stream capture until the plain pass before it has finished
    # on the device -- the order in which a host-side clear of the tick's flags during capture lost "frames in the band" (a race that
    # failed this test once in ~10 full-suite runs before the clear moved in front of the tick, csrc/stream.cpp)
    for graph, delay in ((False, 0), (True, 0), (True, 20)):
        if delay:
            _lib.set_option("debug_capture_delay_ms", delay)
        try:
            st = ServingStream(ex, ms, n_win, win, nd=0, graph=graph)
            st.submit(pcm)
            st.submit(pcm)
            for _ in range(2):
                s_st, a_st, _ms = st.collect()
                assert np.array_equal(s_st, s_fused) and np.array_equal(a_st, a_fused), (graph, delay, float(np.max(np.abs(s_st - s_fused))), int(np.sum(s_st != s_fused)), _lib.last_score_kernel())
        finally:
            _lib.set_option("debug_capture_delay_ms", 0)
    mp_ = MultiPredictor(gm, fs, n_slots=2)
    s_m, a_m = mp_.predict(list(pcm), nd=0)
    assert np.array_equal(s_m, s_fused) and np.array_equal(a_m, a_fused)
    assert _lib.flush_stats()[0] >= calls0 + 5


def test_update_pcm_leaves_the_transfer_in_flight_and_the_callers_buffer_free(built_lib):
    """sr_batch_update_pcm of a serving-size batch returns without waiting for the device (round 6): the samples were copied to
    the batch's page-locked staging area, so the caller may overwrite its buffer at once; two updates in a row wait for the
    first transfer to leave the staging area; results equal those of a freshly built batch, bit for bit.  Also: a result set too
    large for host delivery (> 64 KiB) takes the copying path and gives the same bits for its first rows."""
    from speaker_recognition_amd import synth
    from speaker_recognition_amd.core import Batch, MfccExtractor, ModelSet
    from speaker_recognition_amd.pygmm import GMM
    fs = 16000
    ex = MfccExtractor(fs)
    ubm = synth.synth_gmm(64, 39, 3)
    ms = ModelSet([GMM.from_arrays(*m) for m in [ubm] + [synth.synth_map_speaker(ubm, 40 + s) for s in range(19)]])
    a = synth.synth_speech(1, 1.5, fs)
    b = synth.synth_speech(2, 1.5, fs)[:len(a)]
    want_a = ex.predict_batch(ms, Batch.from_pcm([a]), nd=2)
    want_b = ex.predict_batch(ms, Batch.from_pcm([b]), nd=2)
    assert not np.array_equal(want_a[0], want_b[0])
    batch = Batch.from_pcm([b])
    for i in range(20):
        src, want = ((a, want_a), (b, want_b))[i & 1]
        buf = src.copy()
        batch.update_pcm(buf)
        buf[:] = 12345                              # the caller's buffer is its own again
        if i % 5 == 4:
            batch.update_pcm(((b, a)[i & 1]).copy())   # an update nobody scored ...
            batch.update_pcm(src.copy())               # ... and the one that counts, right behind it
        sums, arg = ex.predict_batch(ms, batch, nd=2)
        assert np.array_equal(sums, want[0]) and np.array_equal(arg, want[1]), i
    # either side of the delivery limits (at most 256 utterances and 64 KiB of results land by themselves, more are copied), and
    # well beyond them: every utterance's row equals the two-utterance batch's, bit for bit
    small = ex.predict_batch(ms, Batch.from_pcm([a[:8000], b[:8000]]), nd=2)
    for n_utt in (255, 256, 257, 1000):
        clips = ([a[:8000], b[:8000]] * ((n_utt + 1) // 2))[:n_utt]
        big = ex.predict_batch(ms, Batch.from_pcm(clips), nd=2)
        assert np.array_equal(big[0][:2], small[0]) and np.array_equal(big[1][:2], small[1]), n_utt
        assert np.array_equal(big[0][::2], np.repeat(small[0][:1], (n_utt + 1) // 2, axis=0)), n_utt
        assert np.array_equal(big[1][1::2], np.repeat(small[1][1:], n_utt // 2)), n_utt


def test_serving_stream_double_buffered_equals_synchronous(built_lib):
    """sr_stream_*: ticks submitted two deep (H2D of tick i+1 on its own HIP stream while tick i
    computes) return exactly what the synchronous fused step returns for the same windows, with
    plain launches and with the tick replayed as a captured hipGraph (SR_STREAM_GRAPH)."""
    from speaker_recognition_amd import synth
    from speaker_recognition_amd.core import Batch, MfccExtractor, ModelSet, ServingStream
    from speaker_recognition_amd.pygmm import GMM
    fs, nwin = 8000, 6
    ex = MfccExtractor(fs)
    ms = ModelSet([GMM.from_arrays(*synth.synth_gmm(32, 13, 60 + s)) for s in range(5)])
    audio = synth.synth_speech(4, 12.0, fs)
    ticks = [np.stack([audio[(t * nwin + j) * 3000:(t * nwin + j) * 3000 + fs] for j in range(nwin)]) for t in range(5)]
    want = [ex.predict_batch(ms, Batch.from_pcm(list(tk)), nd=0) for tk in ticks]
    for graph in (False, True):         # plain launches, then the captured hipGraph replay
        st = ServingStream(ex, ms, nwin, fs, graph=graph)
        got = []
        st.submit(ticks[0])
        for t in range(1, 5):
            st.submit(ticks[t])             # two in flight
            if graph and t == 3:
                # other API traffic between ticks rewrites the library's cached workspaces: the
                # session must notice and re-capture instead of replaying stale pointers / tables
                ex.predict_batch(ms, Batch.from_pcm([audio[:3 * fs], audio[fs:5 * fs]]), nd=0)
            got.append(st.collect())
        got.append(st.collect())
        for (ws, wa), (gs, ga, ms_dev) in zip(want, got):
            assert np.array_equal(ws, gs) and np.array_equal(wa, ga), graph
            assert ms_dev > 0
    with pytest.raises(Exception):
        st.collect()                    # nothing in flight


def test_model_interface_default_mix_feature_roundtrip(built_lib, tmp_path):
    """Reference defaults end to end: mix_feature (13 MFCC + 15 LPC = 28 dims, 32/16 ms), 32-mixture
    speaker GMMs, dump -> load -> predict; batch prediction equals one-by-one prediction."""
    from speaker_recognition_amd import synth
    from speaker_recognition_amd.interface import ModelInterface
    fs = 16000
    m = ModelInterface(gmm_kwargs={"seed": 2}, verbose=False)
    for s in (0, 9, 18):
        m.enroll("spk%d" % s, fs, synth.synth_speech(s, 8.0, fs, seed=1000 + s))
    assert len(m.features["spk0"][0]) == 28
    m.train()
    assert m.gmmset.gmms[0].get_dim() == 28 and m.gmmset.gmms[0].get_nr_mixtures() == 32
    f = str(tmp_path / "m.bin")
    m.dump(f)
    m2 = ModelInterface.load(f)
    tests = [(fs, synth.synth_speech(s, 4.0, fs, seed=2000 + s)) for s in (0, 9, 18)]
    one = [m2.predict(*t) for t in tests]
    assert one == ["spk0", "spk9", "spk18"]
    assert m2.predict_many(tests) == one
    assert m2.predict(fs, np.zeros(100, np.int16)) is None          # too short: the reference swallows it (interface.py:89-93)


def test_plain_c_host_example(built_lib, tmp_path):
    """examples/predict_pcm.c -- the ABI driven from C99: models through the reference's text format,
    PCM -> MFCC -> all speakers -> decisions in one call."""
    import subprocess
    from test_abi_cpu import _build_c_example
    exe = _build_c_example(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert r.stdout.count("-> speaker") == 3 and "kernel:" in r.stdout


def test_multi_slot_prediction_equals_single_device(built_lib):
    """One host process, several slots (a host thread + stream per device, models
    replicated, utterances dealt by length, rows gathered on the host).  On a single-GPU box the
    surplus slots share device 0 -- the threading, partitioning and gather are the same code -- so
    1, 2 and 3 slots must all reproduce the single-device fused step bit for bit (an utterance's
    results do not depend on its batch), whatever the utterance lengths (incl. too-short ones)."""
    from speaker_recognition_amd import _lib, synth
    from speaker_recognition_amd.core import Batch, MfccExtractor, ModelSet, MultiPredictor
    from speaker_recognition_amd.pygmm import GMM
    kw = dict(win_length_ms=25, win_shift_ms=10)
    models = [GMM.from_arrays(*synth.synth_gmm(64, 39, 7 + s)) for s in range(5)]
    rng = np.random.default_rng(5)
    secs = [1.0, 0.4, 2.5, 0.05, 1.7, 0.9, 3.1, 0.6, 1.2]          # 0.05 s: too short for any frame
    sigs = [synth.synth_speech(int(rng.integers(0, 30)), s, 16000, seed=100 + i) for i, s in enumerate(secs)]
    ex = MfccExtractor(16000, **kw)
    want_sums, want_arg = ex.predict_batch(ModelSet(models), Batch.from_pcm(sigs), nd=2)
    try:
        for merge in (0, 1):
            # 0: every slot its own thread and share, also when slots share a device (the threaded path on a one-GPU box);
            # 1 (the default, round 4): the slots of one device are one queue -- its first slot takes their work
            _lib.set_option("multi_merge_same_device", merge)
            for n_slots in (1, 2, 3):
                mp = MultiPredictor(models, 16000, n_slots=n_slots, **kw)
                assert mp.n_slots == n_slots and all(0 <= d < _lib.device_count() for d in mp.slot_devices())
                sums, arg = mp.predict(sigs, nd=2)
                assert np.array_equal(arg, want_arg), (merge, n_slots)
                assert np.array_equal(sums, want_sums), (merge, n_slots)
                devs = mp.slot_devices()
                working = [k for k in range(n_slots) if not merge or devs[k] not in devs[:k]]
                assert mp.slot_seconds.shape == (n_slots,) and np.all(mp.slot_seconds[working] > 0), (merge, n_slots)
                assert np.all(np.delete(mp.slot_seconds, working) == 0)
    finally:
        _lib.set_option("multi_merge_same_device", 1)
    # handles are bound to their device: a thread parked on another device index is refused
    if _lib.device_count() == 1:
        _lib.set_thread_device(1)
        try:
            with pytest.raises(_lib.SRError):
                Batch.from_pcm(sigs)                 # device 1 does not exist on this box
        finally:
            _lib.set_thread_device(0)


def test_cfg0_at_stated_size_cli_vs_cpu_restatement(built_lib, tmp_path):
    """BASELINE configs[0] at its stated size: 10 speakers x 30 s of 16 kHz mono WAV to enroll, another
    10 x 30 s to predict, 25 ms / 10 ms frames (2998 frames per file), 13 MFCC, a 16-mixture diagonal GMM
    per speaker, through speaker-recognition.py's enroll / predict tasks.  Every clip is recognised, and
    with the SAME trained models the CPU restatement (float64 MFCC.py port + the C restatement of the
    reference's scorer) takes every decision the same way with per-utterance scores within 2e-3."""
    from oracle import gmm_oracle as go, mfcc_oracle as mo
    from speaker_recognition_amd import cli, synth
    from speaker_recognition_amd.interface import ModelInterface
    fs = 16000
    spk = [3 * i for i in range(10)]
    for s in spk:
        d = tmp_path / ("spk%d" % s)
        d.mkdir()
        wavfile.write(str(d / "enroll.wav"), fs, synth.synth_speech(s, 30.0, fs, seed=1000 + s))
        wavfile.write(str(tmp_path / ("test_spk%d.wav" % s)), fs, synth.synth_speech(s, 30.0, fs, seed=2000 + s))
    model = str(tmp_path / "model.out")
    cli.main(["-t", "enroll", "-i", str(tmp_path / "spk*"), "-m", model, "--mixtures", "16",
              "--win-length-ms", "25", "--win-shift-ms", "10", "--seed", "3", "--no-lpc"])
    args = cli.get_args(["-t", "predict", "-i", str(tmp_path / "test_*.wav"), "-m", model])
    res = cli.task_predict(args.input, args.model)
    assert len(res) == 10
    for f, label in res:
        assert os.path.basename(f).replace("test_", "").replace(".wav", "") == label
    # the same through the one-process multi-GPU path (3 slots: on a one-GPU box they share device 0)
    assert cli.task_predict(args.input, args.model, gpus=3) == res
    m = ModelInterface.load(model)
    kw = dict(win_length_ms=25, win_shift_ms=10)
    params = [go.GMMParams(*g.params()) for g in m.gmmset.gmms]
    assert all(p.K == 16 and p.D == 13 for p in params)
    for f, label in res:
        _, sig = wavfile.read(f)
        feat = mo.extract(fs, sig, **kw)
        assert feat.shape == (2998, 13)                     # T = 2998
        scores = [go.score_all(p, feat) / len(feat) for p in params]
        assert m.gmmset.y[int(np.argmax(scores))] == label
        dev = np.array(m.gmmset.predict_one_scores(m._features(fs, sig))) / len(feat)
        assert np.max(np.abs(dev - np.array(scores)) / np.abs(scores)) < 2e-3


def test_serving_stream_with_hybrid_set_under_graph_capture(built_lib):
    """A set in the hybrid form scores as two sub-sets per call; their model-group tables live with the sub-sets, so a
    serving tick -- also replayed as a captured hipGraph, which cannot take a stream synchronisation -- re-uploads
    nothing in steady state (the shared workspace table used to flip between the two halves on every call; found by
    scripts/debug/fuzz_stream.py).  Stream results equal the synchronous call bit for bit."""
    from speaker_recognition_amd import _lib, synth
    from speaker_recognition_amd.core import Batch, MfccExtractor, ModelSet, ServingStream
    from speaker_recognition_amd.pygmm import GMM
    fs, nwin = 8000, 3
    w, mu, sg = (a.copy() for a in synth.synth_gmm(64, 13, 99))
    sg[3] = 0.04
    mu[3] = mu.mean(0) + 2.0
    ubm = (w, mu, sg)
    ms = ModelSet([GMM.from_arrays(*ubm)] + [GMM.from_arrays(*synth.synth_map_speaker(ubm, 70 + s)) for s in range(14)])
    assert ms.info()["hybrid_vector_mixtures"] >= 1
    ex = MfccExtractor(fs)
    audio = synth.synth_speech(4, 8.0, fs)
    ticks = [np.stack([audio[(t * nwin + j) * 2400:(t * nwin + j) * 2400 + fs] for j in range(nwin)]) for t in range(4)]
    want = [ex.predict_batch(ms, Batch.from_pcm(list(tk)), nd=0) for tk in ticks]
    assert _lib.last_score_kernel().startswith("hybrid")
    for graph in (False, True):
        st = ServingStream(ex, ms, nwin, fs, graph=graph)
        st.submit(ticks[0])
        got = []
        for t in range(1, 4):
            st.submit(ticks[t])
            got.append(st.collect())
        got.append(st.collect())
        for t in range(4):
            assert np.array_equal(got[t][0], want[t][0]) and np.array_equal(got[t][1], want[t][1]), (graph, t)


def test_reference_side_effects_are_opt_in(built_lib, tmp_path, monkeypatch, capfd):
    """The reference's train_model prints its parameter block (pygmm.cc:31-41, :64) and its trainer writes
    ./gmm-training-intermediate-dump.model after every second iteration, announced on stdout (gmm.cc:622-630).  Off by
    default; with sr_set_option("reference_side_effects", 1) the legacy entry point does both, and the file is the
    model as `dump` writes it at that iteration (here the last one)."""
    import ctypes as C
    from speaker_recognition_amd import _lib, synth
    L = built_lib
    monkeypatch.chdir(tmp_path)
    true = synth.synth_gmm(4, 6, 3)
    X = np.ascontiguousarray(synth.draw_frames(true, 1500, 8).astype(np.float64))
    n, d = X.shape
    rows = (C.POINTER(C.c_double) * n)(*[C.cast(X[i].ctypes.data, C.POINTER(C.c_double)) for i in range(n)])
    p = _lib.Parameter(nr_instance=n, nr_dim=d, nr_mixture=4, min_covar=1e-3, threshold=0.0, nr_iteration=2, init_with_kmeans=0,
                       concurrency=2, verbosity=0)
    dumpf = tmp_path / "gmm-training-intermediate-dump.model"
    for on in (0, 1):
        _lib.set_option("reference_side_effects", on)
        L.new_gmm.restype = C.c_void_p
        h = C.c_void_p(L.new_gmm(4, 1))
        L.train_model(h, rows, C.byref(p))
        C.CDLL(None).fflush(None)             # the library prints through C stdio, as the reference does
        out = capfd.readouterr().out
        if not on:
            assert out == "" and not dumpf.exists()
        else:
            assert out.startswith("nr_instance   :   %d\nnr_dim        :   %d\nnr_mixture    :   4\nmin_covar     :   0.001000\n" % (n, d)), out
            assert "init_with_kmeans: 0\nconcurrency   :   2\nverbosity     :   0\n" in out
            assert "dumping model to gmm-training-intermediate-dump.model ...\nmodel dumped to gmm-training-intermediate-dump.model ...\n" in out
            final = tmp_path / "final.model"
            L.dump(h, str(final).encode())
            assert dumpf.read_bytes() == final.read_bytes()
        L.sr_free_gmm(h)
    _lib.set_option("reference_side_effects", 0)


def test_kmeans_fast_assign_equals_the_exact_pass(built_lib):
    """The k-means initialiser's full nearest-centre search decides by ||c||^2 - 2 x.c (one fused multiply-add per point,
    centre and dimension) wherever best and second best are further apart than its error bound allows them to swap, forms
    the winner's distance the reference's way, and leaves every other point to the exact pass (round 3).  The initial
    means it produces must be the exact pass's to the last bit -- also when centres coincide (rows repeated many times:
    exact ties, first centre wins) -- and the counters must show both paths were taken."""
    from speaker_recognition_amd import _lib
    from speaker_recognition_amd.pygmm import GMM
    rng = np.random.default_rng(12)
    cases = []
    cent = rng.normal(0, 3, (64, 39))
    cases.append(((cent[rng.integers(0, 64, 20000)] + rng.normal(0, 1, (20000, 39))).astype(np.float32), 64, 16))
    base = rng.normal(0, 2, (40, 13)).astype(np.float32)                  # 40 distinct rows, each 200 times: duplicate centres are certain
    cases.append((np.repeat(base, 200, axis=0)[rng.permutation(8000)], 48, 5))
    cases.append(((rng.normal(0, 1, (3000, 6)) * 50 + 1000).astype(np.float32), 16, 3))   # far from the origin: a wide tau
    for X, K, conc in cases:
        got = []
        for eng in (1, 0):
            _lib.set_option("kmeans_assign_engine", eng)
            before = _lib.kmeans_fast_stats()
            g = GMM(nr_mixture=K, nr_iteration=0, init_with_kmeans=1, seed=4, concurrency=conc)
            try:
                g.fit(X)
                got.append(g.params())
            except _lib.SRError as e:                 # (duplicate rows can leave a cluster empty: the reference divides by zero there)
                got.append(str(e))
            after = _lib.kmeans_fast_stats()
            assert (after[0] > before[0]) == (eng == 0)
        _lib.set_option("kmeans_assign_engine", 0)
        if isinstance(got[0], str):
            assert got[1] == got[0]
        else:
            for a, b in zip(got[0], got[1]):
                assert np.array_equal(a, b), (K, np.max(np.abs(a - b)))
    assert _lib.kmeans_fast_stats()[1] > 0            # some points did go through the exact pass (the repeated rows)


@pytest.mark.parametrize("cfg", ["configs[1]", "configs[2]"])
def test_per_frame_ll_from_pcm_end_to_end(built_lib, oracle_built, cfg):
    """north_star's criterion on identical INPUTS: int16 PCM -> (device: MFCC -> CMVN -> deltas -> GMM) per-frame log-likelihoods
    against (float64 restatement of MFCC.py:49-79 + utils.py:24-31, never rounded to float32) -> the reference's scoring
    arithmetic (gmm.cc:237-244 via oracle mode FAST): |d| <= 1e-4 max(1, |LL|) per frame, identical clamp decisions, per-utterance
    sums and argmax.  52 utterances of the configs' own audio (SURVEY.md 8d speakers, seed 2000 + u), 3 s each;
    configs[1]: 100 models x 64 mixtures (a strided 20 of them checked per frame), configs[2]: 512-mixture UBM + 9 MAP speakers."""
    import bench
    from oracle import mfcc_oracle as mo
    from speaker_recognition_amd import synth
    from speaker_recognition_amd.core import Batch, MfccExtractor, ModelSet
    from speaker_recognition_amd.pygmm import GMM
    go = oracle_built
    if cfg == "configs[1]":
        raw = [synth.synth_gmm(bench.CFG1_MIX, bench.DIM, bench.MODEL_SEED + s) for s in range(bench.CFG1_MODELS)]
        check = list(range(0, bench.CFG1_MODELS, 5))
    else:
        ubm = synth.synth_gmm(bench.CFG2_MIX, bench.DIM, 99)
        raw = [ubm] + [synth.synth_map_speaker(ubm, 500 + s) for s in range(9)]
        check = list(range(len(raw)))
    ms = ModelSet([GMM.from_arrays(*m) for m in raw])
    n_utt = 52
    pcm = [synth.synth_speech(u % 100, 3.0, bench.FS, seed=bench.AUDIO_SEED + u) for u in range(n_utt)]
    ex = MfccExtractor(bench.FS, **bench.MFCC_KW)
    fb = ex.extract_batch(Batch.from_pcm(pcm), nd=bench.ND)
    sums, arg, fll = ms.score(fb, frame_ll=True)
    off = fb.offsets()
    X64 = np.concatenate([mo.extract(bench.FS, p, diff=True, nd=bench.ND, **bench.MFCC_KW) for p in pcm])
    assert len(X64) == off[-1] and X64.dtype == np.float64
    worst = 0.0
    want = np.zeros((n_utt, len(check)))
    for j, s in enumerate(check):
        ll = go.score_batch(go.GMMParams(*[np.asarray(a, dtype=np.float64) for a in raw[s]]), X64, go.MODE_FAST)
        d = fll[s].astype(np.float64)
        worst = max(worst, float(np.max(np.abs(d - ll) / np.maximum(1.0, np.abs(ll)))))
        assert np.array_equal(fll[s] == np.float32(go.LN_1E_15), ll == go.LN_1E_15)
        want[:, j] = [ll[off[u]:off[u + 1]].sum() for u in range(n_utt)]
    assert worst <= 1e-4, worst                               # the gate
    assert worst <= 2e-5, worst                               # what the float64 feature stage leaves (measured ~3e-6)
    got = sums[:, check]
    assert np.max(np.abs(got - want) / np.maximum(1.0, np.abs(want))) < 1e-5
    assert np.array_equal(np.argmax(got, axis=1), np.argmax(want, axis=1))
