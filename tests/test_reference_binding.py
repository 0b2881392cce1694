"""The reference's OWN Python binding (src/gmm/python/pygmm.py) and speaker-set classes
(src/testbench/gmmset.py) executed against lib/pygmm.so: the drop-in claim of INTEGRATION.md run
with the reference's code rather than with a look-alike.

oracle/make_ref_py.py (a build recipe, like the reference DSO's Makefile) turns those two files into
importable Python-3 modules under oracle/_ref/ -- git-ignored, never committed, they travel to the GPU
box with the snapshot -- by mechanical edits only (library path, the restype stub of INTEGRATION.md,
print / bytes / iteritems).  Where oracle/_ref/ was not generated the tests skip and say so."""
import importlib
import os
import pickle
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFDIR = os.path.join(ROOT, "oracle", "_ref")


@pytest.fixture(scope="module")
def refmods(built_lib):
    from speaker_recognition_amd import _lib
    if not os.path.exists(os.path.join(REFDIR, "ref_pygmm_py3.py")):
        if os.path.isdir("/root/reference"):
            import subprocess
            subprocess.check_call([sys.executable, os.path.join(ROOT, "oracle", "make_ref_py.py")])
        else:
            pytest.skip("oracle/_ref/ref_pygmm_py3.py was not generated (needs /root/reference: python oracle/make_ref_py.py)")
    os.environ["SR_REF_BINDING_LIB"] = _lib.LIB_PATH
    if REFDIR not in sys.path:
        sys.path.insert(0, REFDIR)
    return importlib.import_module("ref_pygmm_py3"), importlib.import_module("ref_gmmset_py3")


def _write_model(go, g, c, path):
    p = go.GMMParams(g[c + "_w"], g[c + "_mean"], g[c + "_sigma"])
    with open(path, "w") as f:
        f.write(go.format_model_text(p))
    return p


def test_reference_binding_host_side(refmods, oracle_built, gmm_golden, tmp_path):
    """No GPU needed: handles survive 64-bit Pythons, the text format round-trips through the
    reference's own dump / dumps / loads, GMMSet.load_gmm and the pickle hooks work, and a compute call
    reaches our library and fails LOUDLY there (no CPU fallback)."""
    rp, rg = refmods
    go, g = oracle_built, gmm_golden
    m = rp.GMM(7)
    assert m.get_nr_mixtures() == 7 and m.get_dim() == 0
    path = str(tmp_path / "ubm64.model")
    p = _write_model(go, g, "ubm64", path)
    m = rp.GMM.load(path)
    assert m.get_nr_mixtures() == 64 and m.get_dim() == p.D     # (load() re-runs __init__: the nr_mixture ATTRIBUTE is back at 10, pygmm.py:62)
    assert m.dumps() == open(path).read()                   # byte-identical through the reference's /tmp round trip
    m2 = rp.GMM.loads(m.dumps())
    assert m2.get_nr_mixtures() == 64 and m2.get_dim() == p.D
    gs = rg.GMMSetPyGMM(concurrency=3)
    gs.load_gmm("alice", path)
    gs.load_gmm("bob", path)
    assert gs.y == ["alice", "bob"] and gs.gmms[0].concurrency == 3
    gs.before_pickle()
    blob = pickle.dumps(gs)
    gs2 = pickle.loads(blob)
    gs2.after_pickle()
    assert [x.get_nr_mixtures() for x in gs2.gmms] == [64, 64]
    from speaker_recognition_amd import _lib
    if _lib.device_count() == 0:
        s = m.score_all(g["ubm64_X"][:4])
        assert np.isnan(s) and b"no HIP device" in _lib.lib().sr_last_error()


@pytest.mark.gpu
def test_reference_binding_scores_fits_and_predicts_on_the_gpu(refmods, oracle_built, gmm_golden, tmp_path):
    """The reference's GMM.score / score_all / fit(ubm=...) and GMMSetPyGMM.predict_one /
    predict_one_with_rejection, unmodified, on lib/pygmm.so: results against the vectors recorded from
    the reference DSO (tests/golden/make_golden.py)."""
    rp, rg = refmods
    go, g = oracle_built, gmm_golden
    for c in ("ubm32", "ubm256", "syn64x39"):
        path = str(tmp_path / (c + ".model"))
        _write_model(go, g, c, path)
        m = rp.GMM.load(path)
        X, ref = g[c + "_X"], g[c + "_ll"]
        ll = m.score(X)                                     # one ctypes array per frame, as the reference builds them
        assert np.max(np.abs(ll - ref) / np.maximum(1, np.abs(ref))) < 1e-4, c
        assert np.all(ll[-2:] == np.float32(np.log(1e-15)))
        assert abs(m.score_all(X) - float(g[c + "_sum"])) < 1e-4 * abs(float(g[c + "_sum"]))
    # MAP adaptation through the reference's fit(X, ubm): 1 and 4 iterations vs the reference trainer's dumps
    upath = str(tmp_path / "ubm16.model")
    _write_model(go, g, "syn16x13", upath)
    ubm = rp.GMM.load(upath)
    for iters in (1, 4):
        spk = rp.GMM(16, nr_iteration=iters, concurrency=2)
        spk.fit(g["map_X"], ubm)
        q = go.parse_model_text(spk.dumps())
        assert np.max(np.abs(q.mean - g["map%d_mean" % iters])) < 2e-5 * max(1.0, float(np.max(np.abs(g["map%d_mean" % iters]))))
        assert np.array_equal(q.weights, g["map%d_w" % iters]) and np.array_equal(q.sigma, g["map%d_sigma" % iters])
    # identification: three models of one shape, utterances drawn from each (gmmset.py:93-99)
    from speaker_recognition_amd import synth
    gs = rg.GMMSetPyGMM()
    models = [synth.synth_gmm(16, 13, 40 + s) for s in range(3)]
    for s, mdl in enumerate(models):
        pth = str(tmp_path / ("spk%d.model" % s))
        with open(pth, "w") as f:
            f.write(go.format_model_text(go.GMMParams(*mdl)))
        gs.load_gmm("spk%d" % s, pth)
    for s, mdl in enumerate(models):
        x = synth.draw_frames(mdl, 120, 9 + s).astype(np.float64)
        assert gs.predict_one(x) == "spk%d" % s
        want = [float(np.sum(go.score_batch(go.GMMParams(*mm), x))) for mm in models]
        got = gs.predict_one_scores(x)
        assert np.max(np.abs(np.array(got) - want) / np.abs(want)) < 1e-5
    # open-set rule (gmmset.py:69-81) with the UBM being model 0 itself: margin 0 < threshold -> None
    gs.ubm = gs.gmms[0]
    x = synth.draw_frames(models[0], 80, 3).astype(np.float64)
    assert gs.predict_one_with_rejection(x) is None
    gs.reject_threshold = -1.0
    assert gs.predict_one_with_rejection(x) == "spk0"
