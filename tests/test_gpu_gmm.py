"""GPU parity for the scoring path: HIP kernels (through the C ABI) vs the golden vectors
recorded from the reference DSO and vs the oracle on seeded inputs.  Gate:
per-frame |dLL| <= 1e-4 * max(1, |LL|); utterance argmax identical."""
import ctypes as C

import numpy as np
import pytest

from conftest import ll_close

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _gmm(g, c):
    from speaker_recognition_amd.pygmm import GMM
    return GMM.from_arrays(g[c + "_w"], g[c + "_mean"], g[c + "_sigma"])


def is_h2(name):
    """the split-fp16 shared-sigma engine, in any of its workgroup shapes"""
    return "gmm_score_h2s_kernel" in name or "gmm_score_h2p_kernel" in name or "gmm_score_h2m_kernel" in name


SHAPE_NAME = {1: "waves=4>", 2: "waves=12>", 3: "pipelined in the wave>", 4: "models split>"}     # score_h2s_shape -> last_score_kernel()


@pytest.fixture(autouse=True)
def _reset_options(built_lib):
    from speaker_recognition_amd import _lib
    yield
    for k in ("score_frames_per_lane", "score_model_groups", "score_packed", "score_engine", "score_mfma_ft", "mfcc_generic",
              "score_h2s_force_exc", "score_h2s_shape", "score_split_shape"):
        _lib.set_option(k, 0)
    _lib.set_option("flush_order", 2)
    _lib.set_option("flush_list_cap", 0)
    _lib.set_option("score_h2s_pack_tails", 1)


def test_golden_per_frame_ll_all_variants(built_lib, gmm_golden):
    """Every kernel variant (1/2/4 frames per lane, scalar and packed FMA) on the reference's
    shipped UBMs (34-dim, 32/64/256 mixtures) and the synthetic models."""
    from speaker_recognition_amd import _lib
    g = gmm_golden
    for c in g["cases"]:
        m = _gmm(g, c)
        ref = g[c + "_ll"]
        for F, pk, eng, ft in ((0, 0, 1, 0), (1, 0, 1, 0), (2, -1, 1, 0), (4, -1, 1, 0), (2, 1, 1, 0), (4, 1, 1, 0),
                               (0, 0, 3, 1), (0, 0, 3, 2), (0, 0, 5, 1), (0, 0, 5, 2), (0, 0, 0, 0)):
            _lib.set_option("score_frames_per_lane", F)
            _lib.set_option("score_packed", pk)
            _lib.set_option("score_engine", eng)      # 1: vector ALU, 3: split-bf16, 5: split-fp16 matrix cores, 0: auto
            _lib.set_option("score_mfma_ft", ft)
            ll = m.score(g[c + "_X"])
            assert ll_close(ll, ref) < TOL, (c, F, pk, eng, ft, ll_close(ll, ref))
            # the two outlier frames hit the reference's underflow clamp exactly (gmm.cc:34-38)
            assert np.all(ll[-2:] == np.float32(np.log(1e-15)))
            s = m.score_all(g[c + "_X"])
            assert abs(s - float(g[c + "_sum"])) < TOL * abs(float(g[c + "_sum"]))


def test_legacy_double_pp_abi(built_lib, gmm_golden, tmp_path):
    """The ten reference symbols, called the way src/gmm/python/pygmm.py calls them: row
    pointers (double**), score_batch into a caller buffer, score_all, score_instance."""
    L, g = built_lib, gmm_golden
    from speaker_recognition_amd.pygmm import GMM
    c = "ubm32"
    f = tmp_path / "u.model"
    _gmm(g, c).dump(str(f))
    h = C.c_void_p(L.load(str(f).encode()))
    assert L.get_dim(h) == 34 and L.get_nr_mixtures(h) == 32
    X = np.ascontiguousarray(g[c + "_X"])
    n, d = X.shape
    rows = (C.POINTER(C.c_double) * n)(*[C.cast(X[i].ctypes.data, C.POINTER(C.c_double)) for i in range(n)])
    out = (C.c_double * n)()
    L.score_batch(h, rows, out, n, d, 8)
    ll = np.array(out[:])
    # dump keeps 6 significant digits (gmm.cc:655-662): compare against the oracle on the SAME rounded model
    from oracle import gmm_oracle as go
    p = go.parse_model_text(f.read_text())
    want = go.score_batch(p, X)
    assert ll_close(ll, want) < TOL
    tot = L.score_all(h, rows, n, d, 1)
    assert abs(tot - want.sum()) < TOL * abs(want.sum())
    one = L.score_instance(h, X[3].ctypes.data_as(C.POINTER(C.c_double)), d)   # the reference aborts here
    assert abs(one - want[3]) < TOL * max(1, abs(want[3]))
    L.sr_free_gmm(h)


def test_speaker_set_ragged_batch_vs_oracle(built_lib, oracle_built):
    """S models x ragged utterances (empty, 1 frame, tile-boundary lengths) in one fused launch:
    per-frame LL, per-utterance sums and argmax against the oracle."""
    from speaker_recognition_amd import _lib, synth
    from speaker_recognition_amd.core import Batch, ModelSet
    from speaker_recognition_amd.pygmm import GMM
    go = oracle_built
    S, K, D = 7, 24, 39            # K not a multiple of 32 -> partial chunk; 24 -> 6 records
    models = [synth.synth_gmm(K if s != 3 else 5, D, 50 + s) for s in range(S)]   # one odd-sized model
    lens = [0, 1, 255, 256, 257, 511, 513, 1024, 1025, 2, 0, 700]
    utts = [synth.draw_frames(models[u % S], n, 900 + u, outlier_frac=0.01) for u, n in enumerate(lens)]
    ms = ModelSet([GMM.from_arrays(*m) for m in models])
    X = np.concatenate(utts)
    want = np.stack([go.score_batch(go.GMMParams(*m), X.astype(np.float64)) for m in models])
    off = np.concatenate([[0], np.cumsum(lens)])
    want_sums = np.array([[want[s, off[u]:off[u + 1]].sum() for s in range(S)] for u in range(len(lens))])
    for F, pk, G, eng in ((0, 0, 0, 1), (1, 0, 1, 1), (2, -1, 3, 1), (4, -1, 7, 1), (4, 1, 2, 1), (2, 1, 0, 1),
                          (0, 0, 0, 5), (0, 0, 3, 5), (0, 0, 0, 3), (0, 0, 2, 3), (0, 0, 0, 0)):
        _lib.set_option("score_frames_per_lane", F)
        _lib.set_option("score_packed", pk)
        _lib.set_option("score_model_groups", G)
        _lib.set_option("score_engine", eng)
        sums, arg, fll = ms.score(Batch.from_features(utts), frame_ll=True)
        assert ll_close(fll, want) < TOL, (F, pk, G, eng, ll_close(fll, want))
        for u, n in enumerate(lens):
            if n == 0:
                assert arg[u] == -1 and np.all(sums[u] == 0)
            else:
                assert np.max(np.abs(sums[u] - want_sums[u])) < 2e-5 * n * 60, (u, F)
                assert arg[u] == int(np.argmax(want_sums[u])), (u, F, pk, G, eng)


def test_argmax_first_maximum_wins(built_lib):
    """Duplicate models -> exact ties; the reference's max(enumerate(...)) keeps the first."""
    from speaker_recognition_amd import synth
    from speaker_recognition_amd.core import Batch, ModelSet
    from speaker_recognition_amd.pygmm import GMM
    a, b = synth.synth_gmm(8, 13, 1), synth.synth_gmm(8, 13, 2)
    ms = ModelSet([GMM.from_arrays(*m) for m in (b, a, a, b, a)])
    utts = [synth.draw_frames(a, 300, 5), synth.draw_frames(b, 300, 6)]
    sums, arg = ms.score(Batch.from_features(utts))
    assert arg.tolist() == [1, 0]
    assert sums[0, 1] == sums[0, 2] == sums[0, 4] and sums[1, 0] == sums[1, 3]


def test_utterance_sums_long_utterances_many_models(built_lib):
    """gmm_finalize_kernel's order (round 4: an utterance's tiles in segments, a thread per (segment, model), the segment sums
    in order; several passes over the models when segments x models exceed its LDS): one utterance of 50 001 frames (1563
    32-frame tiles = 49 segments) beside short ones and an empty one, against 150 models (3 passes) and against 1 model --
    each sum equals the float64 sum of the same call's per-frame values, the argmax is the first maximum, and the sums do
    not depend on the batch around the utterance."""
    from speaker_recognition_amd import synth
    from speaker_recognition_amd.core import Batch, ModelSet
    from speaker_recognition_amd.pygmm import GMM
    base = [synth.synth_gmm(32, 13, 300 + s) for s in range(5)]
    models = [base[s % 5] for s in range(150)]            # duplicates: ties for the argmax, first maximum wins
    ms = ModelSet([GMM.from_arrays(*m) for m in models])
    long_ = synth.draw_frames(base[3], 50001, 1)
    utts = [synth.draw_frames(base[1], 70, 2), long_, np.zeros((0, 13), np.float32), synth.draw_frames(base[4], 3000, 3)]
    sums, arg, fll = ms.score(Batch.from_features(utts), frame_ll=True)
    off = np.concatenate([[0], np.cumsum([len(u) for u in utts])])
    for u in range(len(utts)):
        want = fll[:, off[u]:off[u + 1]].astype(np.float64).sum(axis=1)
        assert np.allclose(sums[u], want, rtol=1e-12, atol=1e-9), u
    assert arg.tolist() == [1, 3, -1, 4]
    alone, arg1 = ms.score(Batch.from_features([long_]))
    assert np.array_equal(alone[0], sums[1]) and arg1[0] == 3
    one = ModelSet([GMM.from_arrays(*base[3])])
    s1, a1, f1 = one.score(Batch.from_features([long_]), frame_ll=True)
    assert np.allclose(s1[0, 0], f1[0].astype(np.float64).sum(), rtol=1e-12) and a1[0] == 0


def test_deterministic_and_partition_invariant(built_lib):
    """Two runs are bit-identical; splitting an utterance changes nothing but the grouping of
    the (double) partial sums; per-frame values do not depend on the batch they sit in."""
    from speaker_recognition_amd import synth
    from speaker_recognition_amd.core import Batch, ModelSet
    from speaker_recognition_amd.pygmm import GMM
    models = [synth.synth_gmm(64, 39, 70 + s) for s in range(5)]
    ms = ModelSet([GMM.from_arrays(*m) for m in models])
    X = synth.draw_frames(models[2], 5000, 11)
    s1, a1, f1 = ms.score(Batch.from_features([X]), frame_ll=True)
    s2, a2, f2 = ms.score(Batch.from_features([X]), frame_ll=True)
    assert np.array_equal(s1, s2) and np.array_equal(f1, f2)
    s3, a3, f3 = ms.score(Batch.from_features([X[:1234], X[1234:]]), frame_ll=True)
    assert np.array_equal(f3, f1)
    assert np.allclose(s3.sum(axis=0), s1[0], rtol=1e-12, atol=1e-6)
    assert a1[0] == 2


def test_full_size_cfg1_properties(built_lib, oracle_built):
    """BASELINE configs[1] at full size (1e6 frames x 100 speakers x 64 mix x 39 dim): too big for
    the oracle, so check a strided sample of utterances against it and size-independent
    properties on the rest (frames drawn from model s are won by s; sums are finite)."""
    from speaker_recognition_amd import synth
    from speaker_recognition_amd.core import Batch, ModelSet
    from speaker_recognition_amd.pygmm import GMM
    go = oracle_built
    S, K, D, U, T = 100, 64, 39, 1000, 1000
    models = [synth.synth_gmm(K, D, 7 + s) for s in range(S)]
    ms = ModelSet([GMM.from_arrays(*m) for m in models])
    utts = [synth.draw_frames(models[u % S], T, 42 + u, outlier_frac=0.001) for u in range(U)]
    sums, arg = ms.score(Batch.from_features(utts))
    assert np.all(np.isfinite(sums))
    assert np.array_equal(arg, np.arange(U) % S)          # own model wins by a wide margin
    for u in (0, 333, 999):
        want = np.array([go.score_all(go.GMMParams(*m), utts[u].astype(np.float64)) for m in models])
        assert np.max(np.abs(sums[u] - want) / np.abs(want)) < 2e-5
        assert int(np.argmax(want)) == arg[u]


def test_full_size_cfg2_properties(built_lib, oracle_built):
    """BASELINE configs[2] at full size (1e7 frames, 39-dim, a 512-mixture UBM + 200 MAP-adapted
    speakers of 512 mixtures each): 200 distinct utterances repeated 50 times.  Size-independent
    properties: every speaker's utterance is won by that speaker among the speakers, repeated
    utterances give bit-identical sums wherever they sit in the batch, the open-set margin
    (best speaker - UBM, gmmset.py:69-81) is positive; two utterances are checked against the
    oracle on a subset of the models."""
    from speaker_recognition_amd import synth
    from speaker_recognition_amd.core import Batch, ModelSet
    from speaker_recognition_amd.pygmm import GMM
    go = oracle_built
    S, K, D, T, REP = 200, 512, 39, 1000, 50
    ubm = synth.synth_gmm(K, D, 99)
    spk = [synth.synth_map_speaker(ubm, 500 + s) for s in range(S)]
    models = [ubm] + spk
    ms = ModelSet([GMM.from_arrays(*m) for m in models])
    base = [synth.draw_frames(spk[s], T, 9000 + s) for s in range(S)]
    utts = [base[u % S] for u in range(S * REP)]
    sums, arg = ms.score(Batch.from_features(utts))
    from speaker_recognition_amd import _lib
    assert "shared" in _lib.last_score_kernel()        # sigma and weights are common: the shared-sigma engine
    assert sums.shape == (S * REP, S + 1) and np.all(np.isfinite(sums))
    assert np.array_equal(np.argmax(sums[:, 1:], axis=1), np.arange(S * REP) % S)
    assert np.all(sums[:, 1:].max(axis=1) > sums[:, 0])
    for r in (1, 17, REP - 1):
        assert np.array_equal(sums[r * S:(r + 1) * S], sums[:S])
    for u in (3, 9999):
        sub = [0, 1 + u % S, 1 + (u + 1) % S, 1 + (u + 77) % S]
        want = np.array([go.score_all(go.GMMParams(*models[i]), utts[u].astype(np.float64)) for i in sub])
        assert np.max(np.abs(sums[u, sub] - want) / np.abs(want)) < 2e-5


def test_cfg2_ubm_map_speakers_vs_oracle(built_lib, oracle_built):
    """BASELINE configs[2] shape at a size the oracle can follow: a 512-mixture UBM plus MAP-adapted
    speaker models (means shifted, sigmas and weights shared -- gmmubm.cc:40-81), 39-dim; per-frame
    LL of every model and the rejection margin (best speaker - UBM) against the oracle."""
    from speaker_recognition_amd import _lib, synth
    from speaker_recognition_amd.core import Batch, ModelSet
    from speaker_recognition_amd.pygmm import GMM
    go = oracle_built
    ubm = synth.synth_gmm(512, 39, 99)
    spk = [synth.synth_map_speaker(ubm, 500 + s) for s in range(6)]
    models = [ubm] + spk
    utts = [synth.draw_frames(spk[u % 6], 150 + 37 * u, 700 + u) for u in range(4)]
    X = np.concatenate(utts).astype(np.float64)
    want = np.stack([go.score_batch(go.GMMParams(*m), X) for m in models])
    ms = ModelSet([GMM.from_arrays(*m) for m in models])
    off = np.concatenate([[0], np.cumsum([len(u) for u in utts])])
    for eng in (1, 3, 5, 0):
        _lib.set_option("score_engine", eng)
        sums, arg, fll = ms.score(Batch.from_features(utts), frame_ll=True)
        assert ll_close(fll, want) < TOL, (eng, ll_close(fll, want))
        for u in range(4):
            w = np.array([want[s, off[u]:off[u + 1]].sum() for s in range(7)])
            assert int(np.argmax(w[1:])) == int(np.argmax(sums[u, 1:])) == u % 6
            margin = (sums[u, 1:].max() - sums[u, 0]) / len(utts[u])
            assert abs(margin - (w[1:].max() - w[0]) / len(utts[u])) < 1e-3


def test_streaming_shape_short_windows(built_lib, oracle_built):
    """BASELINE configs[4] shape: many short windows (61-98 frames, 1 s of 8 kHz audio) against a
    256-mixture model set; both engines, every tile mostly empty."""
    from speaker_recognition_amd import _lib, synth
    from speaker_recognition_amd.core import Batch, ModelSet
    from speaker_recognition_amd.pygmm import GMM
    go = oracle_built
    models = [synth.synth_gmm(256, 13, 300 + s) for s in range(3)]
    ms = ModelSet([GMM.from_arrays(*m) for m in models])
    lens = [61, 98, 98, 61, 77, 1, 98]
    utts = [synth.draw_frames(models[u % 3], n, 50 + u) for u, n in enumerate(lens)]
    X = np.concatenate(utts).astype(np.float64)
    want = np.stack([go.score_batch(go.GMMParams(*m), X) for m in models])
    off = np.concatenate([[0], np.cumsum(lens)])
    for eng in (1, 3, 0):
        _lib.set_option("score_engine", eng)
        sums, arg, fll = ms.score(Batch.from_features(utts), frame_ll=True)
        assert ll_close(fll, want) < TOL, eng
        for u in range(len(lens)):
            w = np.array([want[s, off[u]:off[u + 1]].sum() for s in range(3)])
            assert int(np.argmax(w)) == arg[u]


def test_edge_shapes_and_errors(built_lib, oracle_built):
    """K = 1, K not a multiple of the 4- / 32-mixture packing, the widest dim of the matrix-core engines
    (64), dims that need zero padding (D=5 -> 8, D=20 -> 24), wide rows (65, 84 = MFCC + LPC with both
    deltas, 128: vector-ALU engine only), zero-weight mixtures, many tiny utterances; dim mismatches
    are refused with a message (no silent truncation)."""
    from speaker_recognition_amd import _lib, synth
    from speaker_recognition_amd.core import Batch, ModelSet
    from speaker_recognition_amd.pygmm import GMM
    go = oracle_built
    for K, D in ((1, 13), (33, 39), (5, 64), (70, 5), (64, 20), (9, 65), (33, 84), (6, 128)):
        models = [synth.synth_gmm(K, D, 900 + K + s) for s in range(3)]
        if K >= 5:
            w, mu, sg = models[1]
            w = w.copy()
            w[2] = 0.0                                  # a dead mixture: contributes exactly nothing
            models[1] = (w, mu, sg)
        ms = ModelSet([GMM.from_arrays(*m) for m in models])
        utts = [synth.draw_frames(models[u % 3], n, 40 + u) for u, n in enumerate([3, 130, 257, 1])]
        X = np.concatenate(utts).astype(np.float64)
        want = np.stack([go.score_batch(go.GMMParams(*m), X) for m in models])
        for eng in ((1, 3) if D <= 64 else (0, 1)):
            _lib.set_option("score_engine", eng)
            sums, arg, fll = ms.score(Batch.from_features(utts), frame_ll=True)
            assert ll_close(fll, want) < TOL, (K, D, eng, ll_close(fll, want))
            if D > 64:
                assert "vector ALU" in _lib.last_score_kernel()
    _lib.set_option("score_engine", 0)
    # 3000 utterances of 1..7 frames
    m = synth.synth_gmm(8, 13, 5)
    rng = np.random.default_rng(0)
    lens = rng.integers(1, 8, size=3000)
    utts = [synth.draw_frames(m, int(n), 1000 + i) for i, n in enumerate(lens)]
    ms = ModelSet([GMM.from_arrays(*m), GMM.from_arrays(*synth.synth_gmm(8, 13, 6))])
    sums, arg = ms.score(Batch.from_features(utts))
    X = np.concatenate(utts).astype(np.float64)
    ll0 = go.score_batch(go.GMMParams(*m), X)
    off = np.concatenate([[0], np.cumsum(lens)])
    want0 = np.array([ll0[off[i]:off[i + 1]].sum() for i in range(3000)])
    assert np.max(np.abs(sums[:, 0] - want0) / np.maximum(1, np.abs(want0))) < 1e-5
    # refused shapes (dim > 128 is served since round 6: test_wide_rows_beyond_128_dims_vs_oracle; what is left is a sanity bound)
    with pytest.raises(_lib.SRError, match="65536"):
        ModelSet([GMM.from_arrays(*synth.synth_gmm(1, 65537, 1))])
    g13 = GMM.from_arrays(*synth.synth_gmm(4, 13, 1))
    with pytest.raises(_lib.SRError, match="dim"):
        g13.score(np.zeros((10, 12), np.float32))
    with pytest.raises(_lib.SRError, match="dim"):
        ModelSet([g13, GMM.from_arrays(*synth.synth_gmm(4, 12, 1))])


def test_wide_rows_beyond_128_dims_vs_oracle(built_lib, oracle_built):
    """The reference has no limit on the feature dimension (src/gmm/src/gmm.cc:40-51).  Rows wider than a lane keeps in
    registers (128) go through gmm_score_wide_kernel: the same direct form with the D loop cut into slices of 64 dimensions.
    D = 129 (one dimension into the third slice), 200, 300, 512 (whole slices), 1100 (beyond the LDS row
    of the partial-product kernel): per-frame LL incl. the clamp on outlier frames, sums, argmax, ragged utterances, model
    groups, odd mixture counts, a dead mixture; the legacy single-model ABI on the same rows."""
    from speaker_recognition_amd import _lib, synth
    from speaker_recognition_amd.core import Batch, ModelSet
    from speaker_recognition_amd.pygmm import GMM
    go = oracle_built
    floor32 = np.float32(np.log(1e-15))
    for D, K, S in ((129, 9, 3), (200, 33, 4), (300, 64, 3), (512, 7, 2), (1100, 5, 2)):
        models = [synth.synth_gmm(K if s != 1 else K + 3, D, 700 + D + s) for s in range(S)]
        if D >= 512:
            # sigma ~ U(0.2, 1.5) puts a 512-dim density below DBL_MIN for every frame (-1.15 nats per dim): tighter mixtures
            # keep the frames scoreable -- and put all of them in the band of the reference's partial-product flushes
            # (gmm_flush.hip; at D = 1100 its rows no longer fit LDS)
            models = [(w, mu, np.vectorize(lambda v: float("%g" % v))(sg * 0.45)) for w, mu, sg in models]
        w, mu, sg = models[0]
        w = w.copy()
        w[K // 2] = 0.0                                  # a dead mixture: contributes exactly nothing
        models[0] = (w, mu, sg)
        lens = [0, 1, 255, 256, 257, 40, 600]
        utts = [synth.draw_frames(models[u % S], n, 300 + u, outlier_frac=0.02 if n > 100 else 0.0) for u, n in enumerate(lens)]
        X = np.concatenate(utts).astype(np.float64)
        want = np.stack([go.score_batch(go.GMMParams(*m), X) for m in models])
        off = np.concatenate([[0], np.cumsum(lens)])
        ms = ModelSet([GMM.from_arrays(*m) for m in models])
        for G in (0, 1, S):
            _lib.set_option("score_model_groups", G)
            sums, arg, fll = ms.score(Batch.from_features(utts), frame_ll=True)
            assert "gmm_score_wide_kernel" in _lib.last_score_kernel()
            clamped = want == np.log(1e-15)
            assert np.array_equal(fll == floor32, clamped), (D, G)
            assert ll_close(fll[~clamped], want[~clamped]) < TOL, (D, G, ll_close(fll[~clamped], want[~clamped]))
            for u, n in enumerate(lens):
                ws = np.array([want[s, off[u]:off[u + 1]].sum() for s in range(S)])
                if n == 0:
                    assert arg[u] == -1 and np.all(sums[u] == 0)
                else:
                    assert np.max(np.abs(sums[u] - ws) / np.maximum(1, np.abs(ws))) < 1e-5, (D, u)
                    assert arg[u] == int(np.argmax(ws)), (D, u)
        _lib.set_option("score_model_groups", 0)
        one = GMM.from_arrays(*models[1])
        ll = one.score(X[:300])
        assert ll_close(ll, want[1, :300]) < TOL
        # bit-identical reruns, and an utterance alone = the same utterance inside the batch
        sums2, arg2 = ms.score(Batch.from_features(utts))
        assert np.array_equal(sums, sums2) and np.array_equal(arg, arg2)
        alone, _ = ms.score(Batch.from_features([utts[6]]))
        assert np.array_equal(alone[0], sums[6])


def test_engine_selection_and_split_bf16_accuracy(built_lib, oracle_built):
    """The dispatcher: well-conditioned sets take the split-bf16 matrix-core kernel, sets whose
    expanded form would cancel (means far apart in units of sigma) or whose 32-mixture tiles are
    mostly padding take the direct-form vector kernel.  And the split is fp32-grade: against the
    float64 oracle its per-frame error is no larger than twice the fp32 FMA-chain engines'."""
    from speaker_recognition_amd import _lib, synth
    from speaker_recognition_amd.core import Batch, ModelSet
    from speaker_recognition_amd.pygmm import GMM
    go = oracle_built
    rng = np.random.default_rng(3)
    S, K, D = 5, 64, 39
    models = [synth.synth_gmm(K, D, 70 + s) for s in range(S)]
    utts = [synth.draw_frames(models[u % S], 700, 10 + u, outlier_frac=0.0) for u in range(4)]
    X = np.concatenate(utts).astype(np.float64)
    want = np.stack([go.score_batch(go.GMMParams(*m), X, go.MODE_LOGSUMEXP, clamp_compat=False) for m in models])
    ms = ModelSet([GMM.from_arrays(*m) for m in models])
    err = {}
    for eng in (1, 3, 5, 0):
        _lib.set_option("score_engine", eng)
        sums, arg, fll = ms.score(Batch.from_features(utts), frame_ll=True)
        err[eng] = float(np.max(np.abs(fll - want) / np.maximum(1.0, np.abs(want))))
        if eng == 0:
            assert "f16x2" in _lib.last_score_kernel()     # well conditioned, moderate sigma range: 3 products
    assert err[3] < 5e-6 and err[3] <= 6.0 * err[1] + 1e-7, err     # (vector ALU: 3.7e-7; the fp32 MFMA chain of rounds 1-4 measured 1.7e-6, bf16x3 1.2e-6)
    assert err[5] < 1e-5 and err[0] == err[5], err            # two fp16 parts: 22 bits per operand
    _lib.set_option("score_engine", 0)
    # ill-conditioned expanded form: means ~30 sigma apart -> direct form on the vector ALU
    far = []
    for s in range(3):
        w, mu, sg = synth.synth_gmm(K, D, 200 + s)
        far.append((w, mu * 40.0, sg))
    utts = [synth.draw_frames(far[u % 3], 300, 90 + u, outlier_frac=0.0) for u in range(3)]
    X = np.concatenate(utts).astype(np.float64)
    want = np.stack([go.score_batch(go.GMMParams(*m), X) for m in far])
    sums, arg, fll = ModelSet([GMM.from_arrays(*m) for m in far]).score(Batch.from_features(utts), frame_ll=True)
    assert "vector ALU" in _lib.last_score_kernel()
    assert ll_close(fll, want) < TOL
    # K = 8: three quarters of every 32-mixture tile would be padding -> vector kernel
    small = [synth.synth_gmm(8, 13, 300 + s) for s in range(3)]
    ModelSet([GMM.from_arrays(*m) for m in small]).score(Batch.from_features([synth.draw_frames(small[0], 200, 1)]))
    assert "vector ALU" in _lib.last_score_kernel()


def test_random_shapes_all_engines(built_lib, oracle_built):
    """Seeded sweep over random shapes -- dims 1..64, mixtures 1..300, 1..6 models of different sizes,
    ragged utterances incl. empty ones -- with every engine forced in turn: per-frame LL, sums and
    argmax against the oracle."""
    from speaker_recognition_amd import _lib, synth
    from speaker_recognition_amd.core import Batch, ModelSet
    from speaker_recognition_amd.pygmm import GMM
    go = oracle_built
    rng = np.random.default_rng(2024)
    for case in range(48):
        D = int(rng.integers(1, 65))
        S = int(rng.integers(1, 7))
        Ks = [int(rng.integers(1, 301)) for _ in range(S)]
        models = [synth.synth_gmm(Ks[s], D, 4000 + 10 * case + s) for s in range(S)]
        lens = [int(v) for v in rng.choice([0, 1, 2, 31, 32, 33, 127, 128, 129, 255, 256, 257, 300, 640], size=int(rng.integers(1, 7)))]
        if sum(lens) == 0:
            lens.append(5)
        utts = [synth.draw_frames(models[u % S], n, 77 + 100 * case + u, outlier_frac=0.02) for u, n in enumerate(lens)]
        X = np.concatenate(utts).astype(np.float64)
        want = np.stack([go.score_batch(go.GMMParams(*m), X) for m in models])
        off = np.concatenate([[0], np.cumsum(lens)])
        ms = ModelSet([GMM.from_arrays(*m) for m in models])
        for eng in (1, 3, 5, 0):
            _lib.set_option("score_engine", eng)
            _lib.set_option("score_model_groups", int(rng.integers(0, 4)))
            sums, arg, fll = ms.score(Batch.from_features(utts), frame_ll=True)
            assert ll_close(fll, want) < TOL, (case, D, Ks, lens, eng, ll_close(fll, want))
            for u, n in enumerate(lens):
                if n == 0:
                    assert arg[u] == -1 and np.all(sums[u] == 0)
                    continue
                w = want[:, off[u]:off[u + 1]].sum(axis=1)
                assert np.max(np.abs(sums[u] - w)) < 2e-5 * n * 60 + 1e-3, (case, eng, u)
                order = np.sort(w)
                if S == 1 or order[-1] - order[-2] > 1e-3 * n:      # skip near-ties
                    assert arg[u] == int(np.argmax(w)), (case, eng, u)


def test_wide_split_shapes_equal_the_four_wave_kernel(built_lib, oracle_built):
    """gmm_score_splitp_kernel (round 4: one wide workgroup per CU, a 32-frame tile per wave, a chunk's log-sum-exp under the next
    chunk's MFMAs) runs the 4-wave kernel's arithmetic value by value, on the same 32-frame tiles, and adds a tile's frames up in the
    same order: per-frame log-likelihoods AND per-utterance sums are EQUAL bit for bit (which form a batch's size selects does not
    show in the results); and both agree with the oracle.  Shapes around every border the kernel
    has: 1 .. 13 chunks per model (stage of 4 or 2, ring of 3), 1 .. 37 models (slab flush every 16), model groups, ragged
    utterances, tiles that end inside a workgroup, frames at the reference's underflow clamp, with it on and off."""
    from speaker_recognition_amd import _lib, synth
    from speaker_recognition_amd.core import Batch, ModelSet
    from speaker_recognition_amd.pygmm import GMM
    go = oracle_built
    rng = np.random.default_rng(404)
    cases = [(39, 64, 5), (39, 256, 1), (13, 32, 37), (26, 96, 17), (39, 416, 2), (20, 160, 3), (9, 33, 16), (40, 64, 4), (48, 32, 2), (34, 128, 33)]
    for case, (D, K, S) in enumerate(cases):
        models = [synth.synth_gmm(K, D, 900 + 10 * case + s) for s in range(S)]
        lens = [int(v) for v in rng.choice([1, 31, 32, 33, 100, 511, 512, 513, 700, 1030], size=int(rng.integers(2, 7)))]
        utts = [synth.draw_frames(models[u % S], n, 50 + 100 * case + u, outlier_frac=0.02) for u, n in enumerate(lens)]
        X = np.concatenate(utts).astype(np.float64)
        want = np.stack([go.score_batch(go.GMMParams(*m), X) for m in models])
        ms = ModelSet([GMM.from_arrays(*m) for m in models])
        _lib.set_option("score_engine", 5)
        for clamp in (True, False):
            _lib.set_option("score_split_shape", 1)
            _lib.set_option("score_model_groups", 0)
            sums0, arg0, fll0 = ms.score(Batch.from_features(utts), frame_ll=True, clamp_compat=clamp)
            assert "gmm_score_split_kernel" in _lib.last_score_kernel()
            if clamp:
                assert ll_close(fll0, want) < TOL
            for waves in (8, 12, 16):
                for groups in (0, 1, 3):
                    _lib.set_option("score_split_shape", waves)
                    _lib.set_option("score_model_groups", groups)
                    sums, arg, fll = ms.score(Batch.from_features(utts), frame_ll=True, clamp_compat=clamp)
                    name = _lib.last_score_kernel()
                    assert "gmm_score_splitp_kernel" in name, (case, waves, name)
                    assert np.array_equal(fll, fll0), (case, D, K, S, waves, groups, clamp, float(np.max(np.abs(fll - fll0))))
                    assert np.array_equal(arg, arg0)
                    assert np.array_equal(sums, sums0), (case, waves, groups, clamp, float(np.max(np.abs(sums - sums0))))
                    sums2, arg2 = ms.score(Batch.from_features(utts), clamp_compat=clamp)          # without the per-frame output; deterministic
                    assert np.array_equal(sums2, sums) and np.array_equal(arg2, arg)
    # models of different orders: the wide form declines, the 4-wave kernel takes the set
    mixed = [synth.synth_gmm(64, 13, 1), synth.synth_gmm(128, 13, 2)]
    _lib.set_option("score_split_shape", 16)
    ModelSet([GMM.from_arrays(*m) for m in mixed]).score(Batch.from_features([synth.draw_frames(mixed[0], 300, 3)]))
    assert "gmm_score_split_kernel" in _lib.last_score_kernel()


def test_concurrent_host_threads(built_lib, oracle_built):
    """ctypes releases the GIL, so Python threads reach the library concurrently; the entry points
    serialise on one lock (one stream and cached workspaces per process) and every thread still gets
    its own results."""
    import threading
    from speaker_recognition_amd import synth
    from speaker_recognition_amd.core import Batch, ModelSet
    from speaker_recognition_amd.pygmm import GMM
    sets, batches, want = [], [], []
    for t in range(4):
        D, K, S = (13, 32, 3) if t % 2 else (39, 64, 5)
        models = [synth.synth_gmm(K, D, 800 + 10 * t + s) for s in range(S)]
        utts = [synth.draw_frames(models[u % S], 100 + 173 * u + 50 * t, 60 + u + 10 * t) for u in range(4)]
        ms = ModelSet([GMM.from_arrays(*m) for m in models])
        b = Batch.from_features(utts)
        sets.append(ms)
        batches.append(b)
        want.append(ms.score(b))
    got = [[] for _ in range(4)]

    def worker(i):
        for _ in range(25):
            got[i].append(sets[i].score(batches[i]))

    th = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for i in range(4):
        assert len(got[i]) == 25
        for sums, arg in got[i]:
            assert np.array_equal(sums, want[i][0]) and np.array_equal(arg, want[i][1])


def test_shared_sigma_engine_vs_oracle(built_lib, oracle_built):
    """Sets whose models share sigma and weights (a UBM and speakers MAP-adapted from it,
    gmmubm.cc:40-81) take the shared-sigma kernel: the quadratic half of the contraction once per
    block of 15 models.  Per-frame LL, sums and argmax against the oracle for set sizes around the
    block size (phantom padding), K not a multiple of 32, ragged utterances, several model groups;
    the general engines agree; sets that do not qualify refuse engine 4."""
    from speaker_recognition_amd import _lib, synth
    from speaker_recognition_amd.core import Batch, ModelSet
    from speaker_recognition_amd.pygmm import GMM
    go = oracle_built
    for K, D, S in ((64, 39, 14), (40, 13, 15), (96, 26, 31), (33, 39, 16), (128, 48, 12)):
        ubm = synth.synth_gmm(K, D, 1234 + K)
        models = [ubm] + [synth.synth_map_speaker(ubm, 7000 + s) for s in range(S)]
        lens = [0, 1, 127, 128, 129, 300, 33]
        utts = [synth.draw_frames(models[1 + u % S], n, 11 + u, outlier_frac=0.01) for u, n in enumerate(lens)]
        X = np.concatenate(utts).astype(np.float64)
        want = np.stack([go.score_batch(go.GMMParams(*m), X) for m in models])
        off = np.concatenate([[0], np.cumsum(lens)])
        ms = ModelSet([GMM.from_arrays(*m) for m in models])
        # engine 6 = the split-fp16 shared-sigma kernel (reference-offset log-sum-exp); its third
        # entry forces every workgroup through the exception (online) pass
        # a fourth entry = workgroup shape of engine 6 (1: 4 waves, 2: 12 waves sharing one LDS copy of the stream)
        for eng, G, force, cols in ((0, 0, 0, 0), (4, 1, 0, 0), (4, 2, 0, 0), (4, 3, 0, 0), (6, 1, 0, 1), (6, 2, 0, 1),
                                    (6, 3, 0, 1), (6, 0, 1, 1), (6, 1, 0, 2), (6, 2, 0, 2), (6, 3, 0, 2), (6, 0, 1, 2),
                                    (6, 1, 0, 3), (6, 2, 0, 3), (6, 3, 0, 3), (6, 0, 1, 3),
                                    (6, 1, 0, 4), (6, 2, 0, 4), (6, 0, 0, 4), (6, 0, 1, 4), (3, 0, 0, 0), (1, 0, 0, 0)):
            _lib.set_option("score_engine", eng)
            _lib.set_option("score_model_groups", G)
            _lib.set_option("score_h2s_force_exc", force)
            _lib.set_option("score_h2s_shape", cols)
            sums, arg, fll = ms.score(Batch.from_features(utts), frame_ll=True)
            if cols:      # (shape 3, the loop pipelined inside the wave, exists up to 8 MFMAs per chain: D <= 42)
                assert SHAPE_NAME[2 if cols == 3 and D > 42 else cols] in _lib.last_score_kernel(), (D, cols, _lib.last_score_kernel())
            if eng in (4, 6) or (eng == 0 and (K, S) == (64, 14)):      # auto also weighs the padding (phantom models, K % 32)
                assert "shared" in _lib.last_score_kernel(), (K, D, S, eng)
            if eng == 6 or (eng == 0 and (K, S) == (64, 14)):
                assert is_h2(_lib.last_score_kernel()), (K, D, S, eng)
            assert ll_close(fll, want) < TOL, (K, D, S, eng, G, ll_close(fll, want))
            for u, n in enumerate(lens):
                if n == 0:
                    assert arg[u] == -1 and np.all(sums[u] == 0)
                    continue
                w = want[:, off[u]:off[u + 1]].sum(axis=1)
                assert np.max(np.abs(sums[u] - w)) < 2e-5 * n * 60 + 1e-3, (K, D, S, eng, u)
    _lib.set_option("score_model_groups", 0)
    _lib.set_option("score_h2s_force_exc", 0)
    _lib.set_option("score_h2s_shape", 0)
    # different sigmas -> not eligible
    other = [synth.synth_gmm(64, 39, 900 + s) for s in range(14)]
    ms = ModelSet([GMM.from_arrays(*m) for m in other])
    _lib.set_option("score_engine", 4)
    with pytest.raises(Exception):
        ms.score(Batch.from_features([synth.draw_frames(other[0], 50, 1)]))
    _lib.set_option("score_engine", 0)
    ms.score(Batch.from_features([synth.draw_frames(other[0], 50, 1)]))
    assert "shared" not in _lib.last_score_kernel()


def test_fp16_engine_range_fallback(built_lib, oracle_built):
    """The split-fp16 engine works on x' = (x - centre) * 2^-e per dimension; a frame whose scaled
    |x'| reaches 255 saturates, is reported by the kernel, and the batch is re-scored by the
    fp32-grade engines: results equal engine 3's bit for bit, and frames inside the range are
    untouched by the presence of the flag logic."""
    from speaker_recognition_amd import _lib, synth
    from speaker_recognition_amd.core import Batch, ModelSet
    from speaker_recognition_amd.pygmm import GMM
    go = oracle_built
    S, K, D = 3, 64, 20
    models = [synth.synth_gmm(K, D, 900 + s) for s in range(S)]
    ms = ModelSet([GMM.from_arrays(*m) for m in models])
    utts = [synth.draw_frames(models[u % S], 200, 5 + u) for u in range(3)]
    far = [u.copy() for u in utts]
    far[1][17, 3] += 5000.0                       # one coordinate of one frame far outside fp16's reach
    for compat in (True, False):                  # with and without the reference's clamp
        _lib.set_option("score_engine", 3)
        want = ms.score(Batch.from_features(far), frame_ll=True, clamp_compat=compat)
        _lib.set_option("score_engine", 0)
        got = ms.score(Batch.from_features(far), frame_ll=True, clamp_compat=compat)
        assert "bf16x3" in _lib.last_score_kernel()           # the re-run
        for a, b in zip(want, got):
            assert np.array_equal(a, b)
        got = ms.score(Batch.from_features(utts), frame_ll=True, clamp_compat=compat)
        assert "f16x2" in _lib.last_score_kernel()
        X = np.concatenate(utts).astype(np.float64)
        ref = np.stack([go.score_batch(go.GMMParams(*m), X, go.MODE_LOGSUMEXP, clamp_compat=compat) for m in models])
        assert ll_close(got[2], ref) < TOL


def test_clamp_band_matches_reference_all_engines(built_lib, clamp_golden):
    """SURVEY 8a-12 at the boundary: frames whose largest term w_k p_k walks through DBL_MIN
    (goldens from the reference DSO, tests/golden/make_clamp_golden.py).  Every engine returns
    exactly ln(1e-15) where the reference does -- including the ln K wide band where the log of the
    SUM is still above -708.396 -- and the true value (1e-4 relative) elsewhere."""
    from speaker_recognition_amd import _lib
    g = clamp_golden
    floor32 = np.float32(np.log(1e-15))
    for c in g["cases"]:
        m = _gmm(g, c)
        X, ref = g[c + "_X"], g[c + "_ll"]
        clamped = ref == np.log(1e-15)
        for eng, wide in ((1, 0), (3, 0), (5, 1), (5, 8), (5, 12), (5, 16), (0, 0)):
            _lib.set_option("score_engine", eng)
            _lib.set_option("score_split_shape", wide)       # the generic split-fp16 engine: 4-wave kernel / the wide pipelined forms
            ll = m.score(X)
            assert np.array_equal(ll == floor32, clamped), (c, eng, wide, _lib.last_score_kernel())
            assert ll_close(ll[~clamped], ref[~clamped]) < TOL, (c, eng, wide)
            assert abs(m.score_all(X) - float(np.sum(ref))) < TOL * abs(float(np.sum(ref)))
            if wide > 1:
                assert "gmm_score_splitp_kernel" in _lib.last_score_kernel() and "waves=%d" % wide in _lib.last_score_kernel()
        _lib.set_option("score_split_shape", 0)


def test_partial_product_flushes_match_reference_all_engines(built_lib, oracle_built, flush_golden):
    """SURVEY 8a-12, closed in round 3: the reference's FTZ arithmetic zeroes a mixture when ANY intermediate of its
    density product dips below DBL_MIN (gmm.cc:192-195) or one dimension reaches fastexp.cc's exponent floor
    (:104-131), also when the full product is representable -- frames ~37 sigma out whose offset sits in a few
    dimensions while sigma < 0.399 elsewhere.  Goldens from the reference DSO (tests/golden/make_flush_golden.py; the
    full-product rule of round 2 is wrong on a sixth of them, by up to ~650 nats).  Every engine returns the reference's
    value: exactly ln(1e-15) where the reference clamps, the survivors' log-sum elsewhere; per frame, per utterance sum,
    argmax; also through the legacy one-model path, the shared-sigma engines' exception pass and a hybrid set."""
    from conftest import flush_models
    from speaker_recognition_amd import _lib
    from speaker_recognition_amd.core import Batch, ModelSet
    from speaker_recognition_amd.pygmm import GMM
    go, g = oracle_built, flush_golden
    floor32 = np.float32(np.log(1e-15))
    conditioned = {"d39_flat32", "d20_k64", "d39_flat256", "d39_k32c", "d39_ubm64"}     # fit the matrix-core engines' range
    calls0, pairs0, frames0 = _lib.flush_stats()
    n_wrong_before = 0
    for c in g["cases"]:
        X, ref = g[c + "_X"], g[c + "_ll"]
        models = flush_models(g, c)
        clamped = ref == np.log(1e-15)
        n_wrong_before += int(np.sum(np.abs(g[c + "_full_rule_ll"] - ref) > 1e-3))
        # ---- the legacy one-model path (score / score_all of every engine that fits the model)
        m = GMM.from_arrays(*models[0])
        for eng in ((1, 3, 5, 0) if c in conditioned else (1, 0)):
            _lib.set_option("score_engine", eng)
            ll = m.score(X)
            assert np.array_equal(ll == floor32, clamped[0]), (c, eng, _lib.last_score_kernel())
            assert ll_close(ll[~clamped[0]], ref[0][~clamped[0]]) < TOL, (c, eng)
            assert abs(m.score_all(X) - float(np.sum(ref[0]))) < TOL * abs(float(np.sum(ref[0]))), (c, eng)
        # ---- a set, ragged utterances (tile boundaries inside and between them), sums + argmax with and without per-frame output
        cuts = [0, 1, 34, 35, 99, len(X)]
        utts = [X[a:b] for a, b in zip(cuts[:-1], cuts[1:])]
        ms = ModelSet([GMM.from_arrays(*mm) for mm in models])
        want_sums = np.array([[ref[s][a:b].sum() for s in range(len(models))] for a, b in zip(cuts[:-1], cuts[1:])])
        engines = [(1, 0, 0, 0), (0, 0, 0, 0)]
        if c in conditioned:
            engines += [(3, 0, 0, 0), (5, 0, 0, 1), (5, 0, 0, 8), (5, 0, 0, 12), (5, 0, 0, 16)]
        if len(models) >= 12:
            engines += [(4, 0, 0, 0), (6, 1, 0, 0), (6, 2, 0, 0), (6, 3, 0, 0), (6, 4, 0, 0), (6, 1, 1, 0), (6, 4, 1, 0)]
        for eng, shape, force, wide in engines:
            _lib.set_option("score_engine", eng)
            _lib.set_option("score_h2s_shape", shape)
            _lib.set_option("score_h2s_force_exc", force)
            _lib.set_option("score_split_shape", wide)
            sums, arg, fll = ms.score(Batch.from_features(utts), frame_ll=True)
            assert np.array_equal(fll == floor32, clamped), (c, eng, shape, force, wide, _lib.last_score_kernel())
            assert ll_close(fll[~clamped], ref[~clamped]) < TOL, (c, eng, wide)
            assert np.max(np.abs(sums - want_sums) / np.maximum(1.0, np.abs(want_sums))) < TOL, (c, eng, shape, force, wide)
            sums2, arg2 = ms.score(Batch.from_features(utts))
            assert np.array_equal(sums2, sums) and np.array_equal(arg2, arg), (c, eng)       # same with sums only; deterministic
            _lib.set_option("score_h2s_force_exc", 0)
        _lib.set_option("score_engine", 0)
        _lib.set_option("score_h2s_shape", 0)
        _lib.set_option("score_split_shape", 0)
        # ---- the source's order of the partial products instead of the compiler's (a DSO built without reassociation)
        _lib.set_option("flush_order", 1)
        ll = m.score(X)
        _lib.set_option("flush_order", 2)
        src = g[c + "_src_order_ll"][0]
        keep = np.abs(src - ref[0]) > 1e-3          # (frames were drawn with a margin on the compiled order's decisions only)
        cl = src == np.log(1e-15)
        assert np.array_equal((ll == floor32)[keep], cl[keep]), c
        assert ll_close(ll[keep & ~cl], src[keep & ~cl]) < TOL, c
    calls1, pairs1, frames1 = _lib.flush_stats()
    assert n_wrong_before >= 150 and calls1 > calls0 and pairs1 > pairs0 and frames1 > frames0
    # ---- a hybrid set (two collapsed mixtures far from the centre go to the vector engine, the rest to the matrix cores;
    #      the band is judged on the merged value against the whole model): the oracle's mode 0 -- equal to the DSO on
    #      every golden above -- is the reference here
    c = "d39_k32c"
    w, mu, sg = flush_models(g, c)[0]
    r6 = np.vectorize(lambda v: float("%g" % v))
    rng = np.random.default_rng(5)
    mu2 = np.concatenate([mu[:-2], r6(1.5 + 0.1 * rng.standard_normal((2, mu.shape[1])))])   # (32 mixtures: no tile padding)
    sg2 = np.concatenate([sg[:-2], np.full((2, mu.shape[1]), 0.01)])
    w2 = r6(np.concatenate([w[:-2], [0.01, 0.01]]))
    hy = ModelSet([GMM.from_arrays(w2, mu2, sg2)])
    X = g[c + "_X"]
    want = go.score_batch(go.GMMParams(w2, mu2, sg2), X, go.MODE_FASTEXP)
    sums, arg, fll = hy.score(Batch.from_features([X]), frame_ll=True)
    assert "hybrid" in _lib.last_score_kernel(), _lib.last_score_kernel()
    cl = want == np.log(1e-15)
    assert np.array_equal(fll[0] == floor32, cl)
    assert ll_close(fll[0][~cl], want[~cl]) < TOL
    assert abs(sums[0, 0] - want.sum()) < TOL * abs(want.sum())


def test_partial_product_band_list_overflow(built_lib, flush_golden):
    """The list of (tile, model) pairs a pass notes for the partial-product path has a capacity; a pass that notes more keeps
    counting, and the results are fetched from a second pass with a list of the counted length.  Forced here with a
    capacity of 1 (`flush_list_cap`): values, sums and argmax equal the default path's, bit for bit."""
    from conftest import flush_models
    from speaker_recognition_amd import _lib
    from speaker_recognition_amd.core import Batch, ModelSet
    from speaker_recognition_amd.pygmm import GMM
    g = flush_golden
    for c in ("d39_k8", "d39_ubm64"):
        X = g[c + "_X"]
        ms = ModelSet([GMM.from_arrays(*m) for m in flush_models(g, c)])
        utts = [X[:50], X[50:51], X[51:]]
        for eng in (1, 0):
            _lib.set_option("score_engine", eng)
            _lib.set_option("flush_list_cap", 0)
            s0, a0, f0 = ms.score(Batch.from_features(utts), frame_ll=True)
            calls0 = _lib.flush_stats()[0]
            _lib.set_option("flush_list_cap", 1)
            s1, a1, f1 = ms.score(Batch.from_features(utts), frame_ll=True)
            s2, a2 = ms.score(Batch.from_features(utts))
            assert _lib.flush_stats()[0] >= calls0 + 2
            assert np.array_equal(f0, f1) and np.array_equal(s0, s1) and np.array_equal(a0, a1), (c, eng)
            assert np.array_equal(s0, s2) and np.array_equal(a0, a2), (c, eng)
            assert ll_close(f1, g[c + "_ll"]) < TOL, (c, eng)


def test_h2s_offset_engine_accuracy_and_exceptions(built_lib, oracle_built):
    """The split-fp16 shared-sigma engine (score_engine 6, the configs[2]/[3] default): per-frame
    error against the float64 oracle on a 512-mixture UBM + 30 MAP speakers stays at the fp32 engines'
    level; frames the reference-offset form cannot vouch for -- +60 sigma outliers (clamped by the
    reference), a speaker model far from the UBM -- travel through the exception pass and come out
    right; results are bit-identical across reruns and do not depend on the batch around an utterance."""
    from speaker_recognition_amd import _lib, synth
    from speaker_recognition_amd.core import Batch, ModelSet
    from speaker_recognition_amd.pygmm import GMM
    go = oracle_built
    K, D, S = 512, 39, 28            # + UBM + rogue = 30 models = two full blocks of 15
    ubm = synth.synth_gmm(K, D, 4242)
    spk = [synth.synth_map_speaker(ubm, 600 + s) for s in range(S)]
    # one "speaker" that is NOT close to the UBM (same sigma and weights, means moved by 3 sigma): its
    # likelihoods sit hundreds of nats away from the offset
    w, mu, sg = ubm
    rogue = (w, np.vectorize(lambda v: float("%g" % v))(mu + 3.0 * sg), sg)
    models = [ubm] + spk + [rogue]
    utts = [synth.draw_frames(spk[u % S], 260 + 11 * u, 31 + u, outlier_frac=0.01 if u % 2 else 0.0) for u in range(6)]
    X = np.concatenate(utts).astype(np.float64)
    ms = ModelSet([GMM.from_arrays(*m) for m in models])
    by_shape = {}
    for compat, cols in ((True, 1), (False, 1), (True, 2), (False, 2), (True, 3), (False, 3), (True, 4), (False, 4)):
        want = np.stack([go.score_batch(go.GMMParams(*m), X, go.MODE_LOGSUMEXP, clamp_compat=compat) for m in models])
        _lib.set_option("score_engine", 0)
        _lib.set_option("score_h2s_shape", cols)
        sums, arg, fll = ms.score(Batch.from_features(utts), frame_ll=True, clamp_compat=compat)
        assert is_h2(_lib.last_score_kernel()) and SHAPE_NAME[cols] in _lib.last_score_kernel()
        # which shape a batch's size selects does not show in the results: the model-split shape of the smallest batches
        # (round 4) leaves the bits of the plain 4-wave one
        by_shape[(compat, cols)] = (sums, fll)
        if cols == 4:
            assert np.array_equal(sums, by_shape[(compat, 1)][0]) and np.array_equal(fll, by_shape[(compat, 1)][1]), compat
        rel = np.abs(fll - want) / np.maximum(1.0, np.abs(want))
        assert rel.max() < 1e-5, (compat, rel.max())
        again = ms.score(Batch.from_features(utts), frame_ll=True, clamp_compat=compat)
        assert np.array_equal(again[0], sums) and np.array_equal(again[2], fll)
        alone = ms.score(Batch.from_features([utts[3]]), clamp_compat=compat)
        assert np.array_equal(alone[0][0], sums[3])
        _lib.set_option("score_engine", 4)
        s4, a4, f4 = ms.score(Batch.from_features(utts), frame_ll=True, clamp_compat=compat)
        assert np.array_equal(a4, arg)
        assert np.max(np.abs(f4 - fll) / np.maximum(1.0, np.abs(want))) < 1e-5


def test_cfg3_shape_k2048_map_speakers_vs_oracle(built_lib, oracle_built):
    """BASELINE configs[3]'s model shape at a size the oracle can follow: a 2048-mixture UBM + 44 MAP
    speakers (45 models = three full blocks of the shared-sigma engines), 39-dim, ragged utterances with
    clamped outliers: per-frame LL of EVERY model against the oracle for the default engine (split-fp16
    shared-sigma), its exception pass alone, and the split-bf16 shared-sigma engine; sums and argmax too."""
    from speaker_recognition_amd import _lib, synth
    from speaker_recognition_amd.core import Batch, ModelSet
    from speaker_recognition_amd.pygmm import GMM
    go = oracle_built
    K, D, S = 2048, 39, 44
    ubm = synth.synth_gmm(K, D, 99)
    models = [ubm] + [synth.synth_map_speaker(ubm, 500 + s) for s in range(S)]
    lens = [130, 64, 1, 257]
    utts = [synth.draw_frames(models[1 + (7 * u) % S], n, 3000 + u, outlier_frac=0.01) for u, n in enumerate(lens)]
    X = np.concatenate(utts).astype(np.float64)
    want = np.stack([go.score_batch(go.GMMParams(*m), X) for m in models])
    off = np.concatenate([[0], np.cumsum(lens)])
    gm = [GMM.from_arrays(*m) for m in models]
    ms = ModelSet(gm)
    _lib.set_option("score_engine", 4)      # sets of more than 65536 mixtures pack only the layout in force at creation
    ms4 = ModelSet(gm)
    for eng, force, cols in ((0, 0, 1), (6, 1, 1), (0, 0, 2), (6, 1, 2), (0, 0, 3), (6, 1, 3), (0, 0, 4), (6, 1, 4), (4, 0, 0)):
        _lib.set_option("score_engine", eng)
        _lib.set_option("score_h2s_force_exc", force)
        _lib.set_option("score_h2s_shape", cols)
        sums, arg, fll = (ms4 if eng == 4 else ms).score(Batch.from_features(utts), frame_ll=True)
        assert is_h2(_lib.last_score_kernel()) == (eng != 4)
        assert ll_close(fll, want) < TOL, (eng, force, ll_close(fll, want))
        for u, n in enumerate(lens):
            w = want[:, off[u]:off[u + 1]].sum(axis=1)
            assert np.max(np.abs(sums[u] - w) / np.abs(w)) < 2e-5, (eng, u)
            assert arg[u] == int(np.argmax(w)), (eng, u)


def test_hybrid_form_of_ill_conditioned_sets(built_lib, oracle_built):
    """A set whose expanded form would cancel in fp32 because of a FEW mixtures (collapsed components at a tiny sigma,
    far from the centre: amp = sum_d (mu'/sigma)^2 in the tens of thousands) is cut in two: those mixtures on the
    direct-form vector engine, the rest on the matrix cores, the per-frame values merged by a log-add-exp.  Per-frame
    LL (also of frames that belong to the tight mixtures, and of +60 sigma outliers under the reference's clamp), sums
    and argmax against the oracle -- for independent models and for a UBM + MAP speakers (the rest takes the
    shared-sigma split-fp16 engine); forcing the vector engine gives the same answers; sets made of such mixtures
    only stay on the vector engine."""
    from speaker_recognition_amd import _lib, synth
    from speaker_recognition_amd.core import Batch, ModelSet
    from speaker_recognition_amd.pygmm import GMM
    go = oracle_built

    def spoil(model, ks, seed):
        w, mu, sg = (a.copy() for a in model)
        rng = np.random.default_rng(seed)
        for k in ks:
            sg[k] = 0.04 + 0.01 * rng.random(sg.shape[1])
            mu[k] = np.round(mu.mean(0) + 2.0 * rng.choice([-1.0, 1.0], size=mu.shape[1]), 4)
        return w, mu, np.round(sg, 5)

    K, D = 64, 20
    indep = [spoil(synth.synth_gmm(K, D, 4100 + s), (3, 40 + s), s) for s in range(5)]
    ubm = spoil(synth.synth_gmm(K, D, 4200), (5, 33), 77)
    shared = [ubm] + [synth.synth_map_speaker(ubm, 4300 + s) for s in range(14)]
    for models, expect in ((indep, "split_kernel"), (shared, "gmm_score_h2")):      # (h2s / h2p / h2m: whichever shape the batch takes)
        utts = [synth.draw_frames(models[u % len(models)], n, 70 + u, outlier_frac=0.01 if u % 2 else 0.0)
                for u, n in enumerate([300, 1, 257, 40, 513])]
        X = np.concatenate(utts).astype(np.float64)
        off = np.concatenate([[0], np.cumsum([len(u) for u in utts])])
        ms = ModelSet([GMM.from_arrays(*m) for m in models])
        assert ms.info()["hybrid_vector_mixtures"] == 2
        for compat in (True, False):
            want = np.stack([go.score_batch(go.GMMParams(*m), X, go.MODE_LOGSUMEXP, clamp_compat=compat) for m in models])
            _lib.set_option("score_engine", 0)
            sums, arg, fll = ms.score(Batch.from_features(utts), frame_ll=True, clamp_compat=compat)
            name = _lib.last_score_kernel()
            assert name.startswith("hybrid") and expect in name, name
            assert ll_close(fll, want) < TOL, (expect, compat, ll_close(fll, want))
            again = ms.score(Batch.from_features(utts), frame_ll=True, clamp_compat=compat)
            assert np.array_equal(again[0], sums) and np.array_equal(again[2], fll)
            for u in range(len(utts)):
                w_ = want[:, off[u]:off[u + 1]].sum(axis=1)
                assert np.max(np.abs(sums[u] - w_) / np.maximum(1.0, np.abs(w_))) < 2e-5
                assert arg[u] == int(np.argmax(w_))
            _lib.set_option("score_engine", 1)
            s1, a1, f1 = ms.score(Batch.from_features(utts), frame_ll=True, clamp_compat=compat)
            assert "vector ALU" in _lib.last_score_kernel() and not _lib.last_score_kernel().startswith("hybrid")
            assert np.array_equal(a1, arg) and ll_close(f1, want) < TOL
        _lib.set_option("score_engine", 0)
    # every mixture tight and far: nothing to gain, the whole set stays on the vector engine
    allbad = [spoil(synth.synth_gmm(8, D, 4400 + s), range(8), s) for s in range(3)]
    ms = ModelSet([GMM.from_arrays(*m) for m in allbad])
    assert ms.info()["hybrid_vector_mixtures"] == 0
    ms.score(Batch.from_features([synth.draw_frames(allbad[0], 100, 1)]))
    assert "vector ALU" in _lib.last_score_kernel() and not _lib.last_score_kernel().startswith("hybrid")


def test_fp16_padding_mixtures_never_outscore_real_ones(built_lib, oracle_built):
    """fp16 layouts cannot hold "minus infinity": the padding of a 32-mixture tile (K = 190, 27) and mixtures of weight 0
    are copies of a real mixture 60000 log2 units down, not a bare constant of -60000 -- which a frame 60 sigma from every
    mixture (true log2 density ~ -2e5) would score BELOW.  Found by scripts/debug/fuzz_shared.py; only visible with the
    reference's clamp OFF (with it on both sides of the comparison are ln 1e-15)."""
    from speaker_recognition_amd import _lib, synth
    from speaker_recognition_amd.core import Batch, ModelSet
    from speaker_recognition_amd.pygmm import GMM
    go = oracle_built
    for K, D, S in ((190, 32, 28), (27, 45, 22), (33, 39, 3)):
        ubm = synth.synth_gmm(K, D, 5000 + K)
        if S >= 12:
            models = [ubm] + [synth.synth_map_speaker(ubm, 5100 + s) for s in range(S - 1)]
        else:
            models = [synth.synth_gmm(K, D, 5200 + s) for s in range(S)]
            w, mu, sg = models[1]
            w = w.copy()
            w[4] = 0.0
            models[1] = (w, mu, sg)
        utts = [synth.draw_frames(models[u % S], n, 5300 + u, outlier_frac=0.05) for u, n in enumerate([257, 100, 33])]
        X = np.concatenate(utts).astype(np.float64)
        ms = ModelSet([GMM.from_arrays(*m) for m in models])
        for compat in (False, True):
            want = np.stack([go.score_batch(go.GMMParams(*m), X, go.MODE_LOGSUMEXP, clamp_compat=compat) for m in models])
            assert compat or want.min() < -60000 * np.log(2)          # the outliers really are below the fp16 floor
            for eng in ((0, 5, 6) if S >= 12 else (0, 5)):
                _lib.set_option("score_engine", eng)
                sums, arg, fll = ms.score(Batch.from_features(utts), frame_ll=True, clamp_compat=compat)
                assert ll_close(fll, want) < TOL, (K, D, S, compat, eng, ll_close(fll, want), _lib.last_score_kernel())
    _lib.set_option("score_engine", 0)


def test_feature_space_far_from_origin(built_lib, oracle_built):
    """Means 40 units from the origin with sigmas of 0.01-0.08 (|mu| / sigma in the thousands): every engine works on
    x - centre (the set's mean of means), so x*s + m keeps its digits -- the vector engine's direct form lost 3 of them
    on raw x (2.3e-3 relative; found by scripts/debug/fuzz_generic.py).  Per-frame LL against the oracle."""
    from speaker_recognition_amd import _lib, synth
    from speaker_recognition_amd.core import Batch, ModelSet
    from speaker_recognition_amd.pygmm import GMM
    go = oracle_built
    r6 = np.vectorize(lambda v: float("%g" % v))
    for D, shift, scale in ((61, -40.0, 0.05), (24, -40.0, 0.05), (14, 5.0, 0.05), (30, -40.0, 30.0)):
        models = []
        for s in range(3):
            w, mu, sg = synth.synth_gmm(40 + 30 * s, D, 6000 + 10 * D + s)
            models.append((w, r6(mu * scale + shift), r6(sg * scale)))
        utts = [synth.draw_frames(models[u % 3], n, 6100 + u) for u, n in enumerate([300, 129, 64])]
        X = np.concatenate(utts).astype(np.float64)
        want = np.stack([go.score_batch(go.GMMParams(*m), X, go.MODE_LOGSUMEXP) for m in models])
        ms = ModelSet([GMM.from_arrays(*m) for m in models])
        for eng in (0, 1, 3):
            _lib.set_option("score_engine", eng)
            sums, arg, fll = ms.score(Batch.from_features(utts), frame_ll=True)
            assert ll_close(fll, want) < TOL, (D, shift, scale, eng, ll_close(fll, want), _lib.last_score_kernel())
    _lib.set_option("score_engine", 0)


def test_matrix_peak_probe_reports_a_plausible_rate(built_lib):
    """sr_mfma_peak_probe (csrc/probe.hip; bench.py's roofline.sustained_mfma): fp16 MFMA chains on every SIMD -- between a
    third of the nominal 2.5 PFLOP/s (a throttled box) and the nominal peak itself, at a clock between 0.8 and 2.5 GHz."""
    from speaker_recognition_amd import _lib
    tflops, mhz = _lib.mfma_peak_probe(20.0)
    assert 800.0 < tflops < 2600.0, tflops
    assert 800.0 < mhz < 2500.0, mhz
    # the two figures describe the same run: 256 CUs x 4 SIMDs x 1024 flop per cycle at a pipe that is (nearly) always busy
    assert 0.85 < tflops * 1e12 / (mhz * 1e6 * 1024 * 1024) <= 1.02, (tflops, mhz)


def test_packed_tail_tiles_leave_every_utterance_its_own_bits(built_lib, oracle_built):
    """The pipelined shared-sigma kernel packs the ragged TAIL tiles of different utterances (count < 32) up to four to a wave
    (csrc/gmm_score.hip: ensure_work_table, gmm_score_h2_shared.hip: h2s_close_block_packed).  An utterance's sums, per-frame
    values and its trip through the exception pass must not depend on what it was packed with: packing on / off bit-identical,
    every utterance scored ALONE bit-identical to its row in the batch, a rogue frame (+60 on every dimension: offset form
    refuses, the online pass takes the tile) in ONE tail leaves its pack-mates' results where they were, all against the oracle."""
    from speaker_recognition_amd import _lib, synth
    from speaker_recognition_amd.core import Batch, ModelSet
    from speaker_recognition_amd.pygmm import GMM
    go = oracle_built
    ubm = synth.synth_gmm(64, 39, 4242)
    models = [ubm] + [synth.synth_map_speaker(ubm, 8100 + s) for s in range(16)]
    ms = ModelSet([GMM.from_arrays(*m) for m in models])
    lens = [1000, 33, 8, 40, 63, 32, 1, 97, 159, 31, 64, 200, 5, 17] * 3           # tails of 8, 1, 8, 31, 1, 1, 31, 31, 8, 5, 17 ... and none
    utts = [synth.draw_frames(models[1 + u % 16], n, 300 + u) for u, n in enumerate(lens)]
    rogue = 3                                                                     # its tail tile (8 frames) is packed with neighbours'
    utts[rogue] = utts[rogue].copy()
    utts[rogue][-2] += 60.0
    _lib.set_option("score_engine", 6)
    _lib.set_option("score_h2s_shape", 3)                                         # the pipelined kernel
    out = {}
    for pack in (1, 0):
        _lib.set_option("score_h2s_pack_tails", pack)
        out[pack] = ms.score(Batch.from_features(utts), frame_ll=True)
        assert "pipelined" in _lib.last_score_kernel()
    _lib.set_option("score_h2s_pack_tails", 1)
    for a, b in zip(out[1], out[0]):
        assert np.array_equal(a, b)
    sums, arg, fll = out[1]
    off = np.concatenate([[0], np.cumsum(lens)])
    for u in (0, 1, 2, 3, 4, 6, 9, 12, 13, 20, 41):
        s1, a1, f1 = ms.score(Batch.from_features([utts[u]]), frame_ll=True)     # alone: nothing to be packed with
        assert np.array_equal(s1[0], sums[u]) and np.array_equal(f1, fll[:, off[u]:off[u + 1]]), u
    X = np.concatenate(utts).astype(np.float64)
    want = np.stack([go.score_batch(go.GMMParams(*m), X) for m in models])
    assert ll_close(fll, want) < TOL
    for u in range(len(lens)):
        w = want[:, off[u]:off[u + 1]].sum(axis=1)
        assert np.max(np.abs(sums[u] - w) / np.maximum(1.0, np.abs(w))) < 1e-5
    # the rogue frame's tile went through the exception pass, its pack-mates' did not change (checked above against the unpacked
    # run); and with EVERY tile forced through the online pass the answers stay within the gate
    _lib.set_option("score_h2s_force_exc", 1)
    s_f, a_f, f_f = ms.score(Batch.from_features(utts), frame_ll=True)
    assert ll_close(f_f, want) < TOL and np.array_equal(a_f, arg)


def test_split_fp16_engines_on_trained_models_vs_oracle(built_lib, oracle_built):
    """The 22-bit split-fp16 engines (a1 b1 dropped) on models that were TRAINED, not drawn: a 64-mixture UBM by EM on the MFCC
    features of 24 synthetic speakers and their means-only MAP adaptations (what enrolment produces: collapsed and wide mixtures side by
    side, weights over four decades), scored on held-out audio.  The dispatcher's guards (amp <= 1000, sigma ratio <= 256, coefficients
    within fp16) decide what runs; whatever does must sit inside the 1e-4 gate per frame against the float64 oracle, and the forced
    fp16 engines -- where the set qualifies -- within 2e-5."""
    import bench
    from speaker_recognition_amd import _lib
    from speaker_recognition_amd.core import Batch, MfccExtractor, ModelSet
    go = oracle_built
    ex = MfccExtractor(bench.FS, **bench.MFCC_KW)
    n_samples = 600 * ex.FRAME_SHIFT + ex.FRAME_LEN
    base = bench.base_clips(24, n_samples)
    ubm, spk = bench.train_cfg2_models(ex, base, 64, em_iters=8)
    models = [ubm] + spk
    ms = ModelSet(models)
    info = ms.info()
    test = [np.rint(base[s] * 0.8).astype(np.int16) for s in range(0, 24, 3)]
    fb = ex.extract_batch(Batch.from_pcm(test), nd=bench.ND)
    X = fb.download().astype(np.float64)
    want = np.stack([go.score_batch(go.GMMParams(*m.params()), X) for m in models])
    worst = {}
    for eng in (0, 1, 3, 4, 5, 6):
        _lib.set_option("score_engine", eng)
        try:
            sums, arg, fll = ms.score(fb, frame_ll=True)
        except Exception as e:                        # a forced engine the set does not qualify for says so
            assert eng in (4, 5, 6) and ("does not qualify" in str(e) or "layout" in str(e)), (eng, str(e))
            continue
        worst[eng] = ll_close(fll, want)
        assert worst[eng] < TOL, (eng, worst[eng], info, _lib.last_score_kernel())
        if eng in (5, 6):
            assert worst[eng] < 2e-5, (eng, worst[eng], info)
    assert 0 in worst and 1 in worst
    _lib.set_option("score_engine", 0)
    # held-out renditions are identified (speaker s of `test` is base[3 s])
    sums, arg = ms.score(fb)
    assert np.array_equal(np.argmax(sums[:, 1:], axis=1), np.arange(0, 24, 3))
