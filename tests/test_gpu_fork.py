"""The reference's fit-then-fork pattern on the GPU (src/test/test-nperson.py:126-139, src/test/test-gmm.py:120-133):
the parent trains through the ABI (GPU runtime up), THEN creates a multiprocessing.Pool with the *fork* start method,
the workers call predict_one.  With the reference's own gmmset.py / pygmm.py (oracle/_ref, mechanical Python-3 edits
of the reference's files) and with this package's mirrors.  Expected: the parent's answers, from every worker,
within a timeout -- no hang, no crash (include/pygmm_hip.h "Processes"; csrc/fork_proxy.cpp)."""
import importlib
import multiprocessing as mp
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFDIR = os.path.join(ROOT, "oracle", "_ref")

pytestmark = pytest.mark.gpu

_STATE = {}          # what the forked workers inherit (the reference passes the set as a task argument; a set of ctypes handles
                     # does not pickle in either Python, so its drivers in effect rely on the inherited copy as well)


def _predict_task(i):
    gs = _STATE["set"]
    return os.getpid(), gs.predict_one(_STATE["utts"][i])


def _scores_task(i):
    gs = _STATE["set"]
    return [float(v) for v in gs.predict_one_scores(_STATE["utts"][i])]


def _pickled_task(args):
    blob, i = args
    import pickle
    gs = pickle.loads(blob)
    gs.after_pickle()
    return gs.predict_one(_STATE["utts"][i])


def _train_task(i):
    """training inside a forked worker: MAP adaptation of one more speaker from the inherited UBM"""
    cls, ubm = _STATE["gmm_cls"], _STATE["ubm"]
    m = cls(ubm.get_nr_mixtures(), nr_iteration=2)
    m.fit(_STATE["utts"][i], ubm)
    return m.dumps()


def _run_pool(fn, args, workers=4, timeout=240):
    ctx = mp.get_context("fork")
    pool = ctx.Pool(workers)
    try:
        res = [pool.apply_async(fn, (a,)) for a in args]
        pool.close()
        return [r.get(timeout=timeout) for r in res]      # a hang in a dead runtime would end here, as a TimeoutError
    finally:
        pool.terminate()


def _speakers(n_spk, dim, frames, seed):
    from speaker_recognition_amd import synth
    models = [synth.synth_gmm(8, dim, seed + s) for s in range(n_spk)]
    train = [synth.draw_frames(m, frames, 100 + s).astype(np.float64) for s, m in enumerate(models)]
    test = [synth.draw_frames(m, 150, 200 + s).astype(np.float64) for s, m in enumerate(models)]
    return train, test


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


def test_reference_gmmset_fit_then_fork_pool(refmods, built_lib):
    rp, rg = refmods
    train, test = _speakers(6, 13, 600, 31)
    gs = rg.GMMSetPyGMM(gmm_order=8, nr_iteration=5, concurrency=2)
    gs.fit(train, list(range(6)))                              # train_model on the GPU, in THIS process
    assert built_lib.sr_gpu_runtime_lost() == 0
    want = [gs.predict_one(x) for x in test]
    want_scores = [[float(v) for v in gs.predict_one_scores(x)] for x in test]
    assert want == list(range(6))
    _STATE.update(set=gs, utts=test)
    got = _run_pool(_predict_task, range(len(test)))
    assert [label for _, label in got] == want
    assert len({pid for pid, _ in got}) >= 1 and os.getpid() not in {pid for pid, _ in got}
    # the numbers, not only the decisions: the helper runs the same kernels on the same data
    got_scores = _run_pool(_scores_task, range(len(test)))
    assert np.max(np.abs(np.array(got_scores) - np.array(want_scores)) / np.abs(np.array(want_scores))) < 1e-6
    # the pickle hooks of the reference's set (gmmset.py:101-105) across the fork as well
    import pickle
    gs.before_pickle()
    blob = pickle.dumps(gs)
    gs.after_pickle()
    # (ONE worker: the reference's GMM.loads goes through the fixed file /tmp/tmp-gmm.load, pygmm.py:84-90 -- two workers
    # unpickling at once read each other's models, whatever library is underneath)
    assert _run_pool(_pickled_task, [(blob, i) for i in range(len(test))], workers=1) == want
    # the parent still works after its children are gone
    assert [gs.predict_one(x) for x in test] == want


def test_package_gmmset_and_training_in_forked_workers(built_lib):
    from speaker_recognition_amd.gmmset import GMMSetPyGMM
    from speaker_recognition_amd.pygmm import GMM
    train, test = _speakers(5, 13, 600, 77)
    ubm = GMM(8, nr_iteration=6, seed=5)
    ubm.fit(np.concatenate(train))
    gs = GMMSetPyGMM(ubm=ubm, nr_iteration=3)
    gs.fit(train, list("abcde"))                               # MAP on the GPU in the parent
    want = gs.predict(test)
    _STATE.update(set=gs, utts=test, gmm_cls=GMM, ubm=ubm)
    assert [label for _, label in _run_pool(_predict_task, range(len(test)))] == want
    # train_model_from_ubm / sr_train_f32 from a forked worker == the same call in the parent
    ref = []
    for x in test[:3]:
        m = GMM(8, nr_iteration=2)
        m.fit(x, ubm)
        ref.append(m.dumps())
    assert _run_pool(_train_task, range(3), workers=3) == ref


def test_forked_workers_score_the_whole_set_in_one_conversation(built_lib):
    """predict_one_scores in a forked worker is ONE conversation with its helper and one fused pass there (sr_score_models_f32) --
    the reference's logged run calls it per utterance against 80 speakers (test-nperson.py:133-146).  Its sums must be the bits of
    the parent's fused pass: utterances of different lengths one after the other in the same worker, a set that names a model
    twice, a model re-trained between two pools (new parameters behind the same handle), and a helper that keeps fewer models than
    the set has (every call evicts, misses and sends the list again)."""
    from speaker_recognition_amd import _lib, synth
    from speaker_recognition_amd.gmmset import GMMSet
    from speaker_recognition_amd.pygmm import GMM
    raw = [synth.synth_gmm(8, 13, 50 + s) for s in range(20)]
    gs = GMMSet(gmm_order=8)
    for s, m in enumerate(raw):
        gs._append("s%d" % s, GMM.from_arrays(*m))
    gs._append("again", gs.gmms[3])                              # the same handle twice
    utts = [synth.draw_frames(raw[u % 20], n, 400 + u).astype(np.float64) for u, n in enumerate([300, 40, 1000, 33, 300, 300])]
    want = [[float(v) for v in gs.predict_one_scores(x)] for x in utts]      # parent: in-process fused pass, GPU runtime up
    _STATE.update(set=gs, utts=utts)
    assert _run_pool(_scores_task, range(len(utts)), workers=1) == want
    assert _run_pool(_scores_task, range(len(utts)), workers=3) == want
    _lib.set_option("debug_helper_max_models", 5)
    try:
        assert _run_pool(_scores_task, range(len(utts)), workers=1) == want
    finally:
        _lib.set_option("debug_helper_max_models", 0)
    gs.gmms[7].fit(utts[2][:, :13], None)                        # new parameters behind the same handle
    want2 = [[float(v) for v in gs.predict_one_scores(x)] for x in utts]
    assert want2 != want
    assert _run_pool(_scores_task, range(len(utts)), workers=2) == want2


def _fresh_child_scores(conn):
    """a pool created BEFORE the first compute call: the worker initialises its own runtime, no helper involved"""
    try:
        from speaker_recognition_amd import _lib, synth
        from speaker_recognition_amd.core import Batch, ModelSet
        from speaker_recognition_amd.pygmm import GMM
        models = [synth.synth_gmm(16, 13, 3 + s) for s in range(3)]
        ms = ModelSet([GMM.from_arrays(*m) for m in models])
        x = synth.draw_frames(models[1], 200, 5)
        sums, arg = ms.score(Batch.from_features([x]))
        conn.send((int(_lib.lib().sr_gpu_runtime_lost()), int(arg[0]), [float(v) for v in sums[0]]))
    except BaseException as e:      # noqa: BLE001
        conn.send(("error", repr(e)))
    finally:
        conn.close()
        os._exit(0)


def test_pool_before_first_compute_call_initialises_per_worker():
    """Run in a fresh interpreter (this pytest process has long used the GPU): import the package, fork, compute in
    the children with the batched interface."""
    import subprocess
    code = (
        "import os, sys, multiprocessing as mp\n"
        "sys.path.insert(0, %r)\n"
        "sys.path.insert(0, %r)\n"
        "import test_gpu_fork as t\n"
        "from speaker_recognition_amd import _lib\n"
        "_lib.lib()\n"                                   # library loaded, runtime untouched
        "out = []\n"
        "for k in range(2):\n"
        "    r, s = mp.Pipe(duplex=False)\n"
        "    pid = os.fork()\n"
        "    if pid == 0:\n"
        "        t._fresh_child_scores(s)\n"
        "    s.close()\n"
        "    assert r.poll(120), 'child hung'\n"
        "    out.append(r.recv()); os.waitpid(pid, 0)\n"
        "print(out)\n"
        "assert all(o[0] == 0 and o[1] == 1 for o in out), out\n"
        "assert out[0] == out[1]\n"
    ) % (ROOT, os.path.join(ROOT, "tests"))
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr


def _many_models_task(_):
    """one forked worker scores MORE models than its helper keeps (the test hook lowers the bound to 64), then the first ones again: the
    helper has evicted them, says so, and the library sends them again (csrc/fork_proxy.cpp: HELPER_MAX_MODELS, "fork helper miss")"""
    models, x = _STATE["many"], _STATE["utts"][0]
    first = [m.score_all(x) for m in models]
    again = [m.score_all(x) for m in models[:8]]
    return first, again


def test_forked_worker_with_more_models_than_its_helper_keeps(built_lib):
    from speaker_recognition_amd import synth
    from speaker_recognition_amd.pygmm import GMM
    models = [GMM.from_arrays(*synth.synth_gmm(4, 13, 900 + s)) for s in range(80)]
    x = synth.draw_frames(synth.synth_gmm(4, 13, 900), 120, 5).astype(np.float64)
    want = [m.score_all(x) for m in models]                 # parent: GPU runtime up before the fork
    _STATE["many"], _STATE["utts"] = models, [x]
    from speaker_recognition_amd import _lib
    _lib.set_option("debug_helper_max_models", 64)          # (logged here, replayed into every helper at its start)
    try:
        (first, again), = _run_pool(_many_models_task, [0], workers=1)
    finally:
        _lib.set_option("debug_helper_max_models", 0)
    assert np.allclose(first, want, rtol=0, atol=0) and np.allclose(again, want[:8], rtol=0, atol=0)
    # ... and with the default bound (the reference's logged run: 80 speakers per worker) nothing is evicted: same values
    (first, again), = _run_pool(_many_models_task, [0], workers=1)
    assert np.allclose(first, want, rtol=0, atol=0) and np.allclose(again, want[:8], rtol=0, atol=0)
