"""fork() and the C ABI (include/pygmm_hip.h "Processes"; csrc/common.cpp, csrc/fork_proxy.cpp).

The reference's own drivers fit in the parent and THEN fork a multiprocessing.Pool whose workers call
predict_one (src/test/test-nperson.py:126-139, src/test/test-gmm.py:120-133).  A HIP runtime does not survive
fork(); a child of a process that had used it must never call into it.  What a forked child gets instead:
host-only entry points as always, the per-model compute entry points through a helper process, every other
device entry point a clean error.  This file runs without a GPU: the helper then answers what the library
answers on such a box ("no HIP device ... no CPU path"), which is exactly what shows that the request went
all the way to a fresh process and back.  tests/test_gpu_fork.py runs the reference's pattern for real."""
import multiprocessing as mp
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _child_report(conn, fn):
    try:
        conn.send(("ok", fn()))
    except BaseException as e:      # noqa: BLE001 -- everything goes back to the parent
        conn.send(("error", repr(e)))
    finally:
        conn.close()
        os._exit(0)


def _in_forked_child(fn, timeout=60):
    """fn() in a child made by a plain os.fork() of THIS process; its result, or a failure if it hangs or dies."""
    recv, send = mp.Pipe(duplex=False)
    pid = os.fork()
    if pid == 0:
        recv.close()
        _child_report(send, fn)
    send.close()
    try:
        if not recv.poll(timeout):
            os.kill(pid, 9)
            pytest.fail("the forked child did not answer within %d s (hung in the inherited GPU runtime?)" % timeout)
        kind, value = recv.recv()
    except EOFError:
        _, status = os.waitpid(pid, 0)
        pytest.fail("the forked child died without an answer (wait status %d)" % status)
    os.waitpid(pid, 0)
    assert kind == "ok", value
    return value


def test_forked_child_after_runtime_use(built_lib):
    """Parent touches the runtime (device count), forks; the child: host-only entry points work, a per-model compute
    call is answered by the helper process, a batched call is refused with the remedy in the message, inherited
    handles can be dropped, nothing hangs."""
    from speaker_recognition_amd import _lib
    from speaker_recognition_amd.core import Batch
    from speaker_recognition_amd.gmmset import GMMSet
    from speaker_recognition_amd.pygmm import GMM
    L = built_lib
    have_gpu = L.sr_device_count() > 0           # (this call is what initialises the runtime in the parent)
    assert L.sr_gpu_runtime_lost() == 0
    rng = np.random.default_rng(3)
    g = GMM.from_arrays(np.full(4, 0.25), rng.standard_normal((4, 5)), np.full((4, 5), 0.8))
    x = rng.standard_normal((20, 5)).astype(np.float32)
    text = g.dumps()

    def child():
        out = {"lost": L.sr_gpu_runtime_lost(), "devices": L.sr_device_count()}
        out["dumps_equal"] = g.dumps() == text and GMM.loads(text).get_nr_mixtures() == 4
        try:
            out["score_all"] = float(g.score_all(x))
        except _lib.SRError as e:
            out["score_all_error"] = str(e)
        try:
            Batch.from_features([x])
            out["batch"] = "created"
        except _lib.SRError as e:
            out["batch_error"] = str(e)
        gs = GMMSet()
        gs.gmms, gs.y = [g, g], ["a", "b"]
        try:
            out["predict_one"] = gs.predict_one(x)
        except _lib.SRError as e:
            out["predict_one_error"] = str(e)
        g2 = GMM.loads(text)
        del g2                                       # sr_free_gmm in a forked child
        return out

    out = _in_forked_child(child)
    assert out["lost"] == 1 and out["devices"] == 0 and out["dumps_equal"]
    assert "forked after its parent" in out["batch_error"] and "spawn" in out["batch_error"]
    if have_gpu:
        assert np.isfinite(out["score_all"]) and out["predict_one"] == "a"
    else:
        # the request reached a fresh process that loaded the library and found what this box has: no GPU
        assert "no HIP device" in out["score_all_error"] and "forked" not in out["score_all_error"]
        assert "no HIP device" in out["predict_one_error"]
    assert L.sr_gpu_runtime_lost() == 0              # the parent is untouched


def test_forked_child_without_helper_binary_says_so(built_lib, monkeypatch):
    from speaker_recognition_amd import _lib
    from speaker_recognition_amd.pygmm import GMM
    L = built_lib
    L.sr_device_count()
    g = GMM.from_arrays(np.full(2, 0.5), np.zeros((2, 3)), np.ones((2, 3)))
    x = np.zeros((4, 3), dtype=np.float32)
    monkeypatch.setenv("SR_FORK_HELPER", "/nonexistent/sr_fork_helper")

    def child():
        try:
            g.score_all(x)
            return "scored"
        except _lib.SRError as e:
            return str(e)

    msg = _in_forked_child(child)
    assert "/nonexistent/sr_fork_helper" in msg and "missing" in msg


def test_legacy_symbols_through_the_helper_keep_their_error_convention(built_lib):
    """score_all returns NaN and parks the message (pygmm.hh has no status channel) in a forked child as in the parent."""
    import ctypes as C
    from speaker_recognition_amd.pygmm import GMM
    L = built_lib
    if L.sr_device_count() > 0:
        pytest.skip("error-path test for boxes without a GPU")
    g = GMM.from_arrays(np.full(2, 0.5), np.zeros((2, 3)), np.ones((2, 3)))
    rows = [(C.c_double * 3)(0.1, 0.2, 0.3) for _ in range(4)]
    X = (C.POINTER(C.c_double) * 4)(*[C.cast(r, C.POINTER(C.c_double)) for r in rows])

    def child():
        s = L.score_all(g.gmm, X, 4, 3, 1)
        return (bool(np.isnan(s)), L.sr_last_error().decode())

    isnan, msg = _in_forked_child(child)
    assert isnan and "no HIP device" in msg


def test_fork_helper_is_built_next_to_the_library(built_lib):
    from speaker_recognition_amd import _lib
    helper = os.path.join(os.path.dirname(_lib.LIB_PATH), "sr_fork_helper")
    assert os.access(helper, os.X_OK), "lib/sr_fork_helper is missing: make -C speaker-recognition_amd/csrc"
