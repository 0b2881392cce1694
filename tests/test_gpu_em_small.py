"""A speaker-sized EM / MAP fit WHOLE in one launch (csrc/em_small.hip; sr_last_em_stats_engine() == 4): the loop of
GMMTrainerBaseline::train (src/gmm/src/gmm.cc:581-653) -- E-step :439-498, M-step :388-437 / gmmubm.cc:53-74, stop rule
:622-650 -- on the device in float64.  Checked against the float64 oracle iterated (oracle/gmm_oracle.c, pinned to the
reference trainer's goldens), against the iteration-at-a-time path (csrc/em.hip, em_stats_engine 3) incl. the iteration
the stop rule ends on, and for the frames it must hand over.  The reference trainer's own goldens run through this kernel
in tests/test_gpu_pipeline.py (test_em_training_vs_reference_trainer_golden, test_train_from_scratch_vs_reference_trainer_golden)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _data(rng, n, K, D, spread=2.0, shift=0.0):
    cent = shift + rng.normal(0, spread, (K, D))
    return (cent[rng.integers(0, K, n)] + rng.normal(0, 0.7, (n, D))).astype(np.float32), cent


def _fit(eng, X, K, iters, threshold, km=0, seed=7, ubm=None, start=None):
    from speaker_recognition_amd import _lib
    from speaker_recognition_amd.pygmm import GMM
    _lib.set_option("em_stats_engine", eng)
    try:
        if start is not None:
            g = GMM.from_arrays(*start)
            g.nr_iteration, g.init_with_kmeans, g.threshold = iters, -1, threshold       # -1: warm start (extension)
        else:
            g = GMM(K, nr_iteration=iters, threshold=threshold, init_with_kmeans=km, seed=seed)
        it = g.fit(X, ubm=ubm) if ubm is not None else g.fit(X)
        return it, g.params(), _lib.last_em_stats_engine()
    finally:
        _lib.set_option("em_stats_engine", 0)


def test_whole_fit_vs_oracle_iterated(built_lib, oracle_built):
    """N iterations with the stop rule off (threshold 0) against the oracle's iteration applied N times, EM and MAP: float64 on
    both sides -- models of 1..32 mixtures, 3..40 dims, frame counts around the 64-frame chunks up to the kernel's limit
    (8192 frames: 128 workgroups, the sums shared out behind a second barrier)."""
    go = oracle_built
    rng = np.random.default_rng(31)
    r6 = np.vectorize(lambda v: float("%g" % v))
    for n, K, D, N in ((3000, 16, 13, 7), (900, 32, 39, 4), (150, 3, 40, 6), (4097, 9, 26, 5), (64, 1, 3, 3), (65, 2, 5, 2), (8192, 32, 34, 3), (8129, 4, 5, 4)):
        X, cent = _data(rng, n, K, D, shift=3.0)
        start = go.GMMParams(np.full(K, 1.0 / K), r6(cent + 0.2 * rng.standard_normal(cent.shape)), np.full((K, D), 0.9))
        want = start
        for _ in range(N):
            want = go.em_iteration(want, X.astype(np.float64))
        it, p, eng = _fit(0, X, K, N, 0.0, start=(start.weights, start.mean, start.sigma))
        assert it == N and eng == 4, (n, K, D, it, eng)
        err = (np.max(np.abs(p[0] - want.weights)), np.max(np.abs(p[1] - want.mean)), np.max(np.abs(p[2] - want.sigma) / want.sigma))
        assert err[0] < 1e-7 and err[1] < 1e-6 and err[2] < 1e-6, (n, K, D, err)
        m = min(n, 300)
        want = start
        for _ in range(N):
            want = go.em_iteration(want, X[:m].astype(np.float64), map_relevance=16.0, ubm=start)
        from speaker_recognition_amd.pygmm import GMM
        ubm = GMM.from_arrays(start.weights, start.mean, start.sigma)
        it, p, eng = _fit(0, X[:m], K, N, 0.0, ubm=ubm)
        assert it == N and eng == 4
        assert np.array_equal(p[0], start.weights) and np.array_equal(p[2], start.sigma)        # means only, gmmubm.cc:29-38
        assert np.max(np.abs(p[1] - want.mean)) < 1e-6, (n, K, D, np.max(np.abs(p[1] - want.mean)))


def test_whole_fit_vs_iteration_at_a_time_with_the_stop_rule(built_lib):
    """Both paths from the same initialisation (random frames / k-means||, the reference's own random numbers) under the
    reference's default stop rule (threshold 0.01, checked after every second iteration, gmm.cc:622-650): the same iteration
    ends both, the models agree inside the training gates; verbosity 1 prints the totals the rule saw."""
    rng = np.random.default_rng(5)
    for n, K, D, km in ((2998, 16, 13, 0), (2998, 16, 13, 1), (1250, 32, 34, 1), (5000, 8, 20, 0), (64, 4, 3, 0), (333, 5, 39, 0)):
        X, _ = _data(rng, n, K, D)
        it3, p3, e3 = _fit(3, X, K, 200, 0.01, km)
        it0, p0, e0 = _fit(0, X, K, 200, 0.01, km)
        assert e0 == 4 and e3 in (1, 2, 3), (e0, e3)
        assert it0 == it3, (n, K, D, km, it0, it3)
        err = (np.max(np.abs(p0[0] - p3[0])), np.max(np.abs(p0[1] - p3[1])), np.max(np.abs(p0[2] - p3[2]) / p3[2]))
        assert err[0] < 2e-5 and err[1] < 2e-4 and err[2] < 1e-3, (n, K, D, km, err)
    # an odd and an even iteration limit below the stop rule's iteration (the total after the LAST iteration is taken when that one is odd)
    X, _ = _data(rng, 2000, 8, 13)
    for iters in (1, 2, 3, 4):
        it3, p3, _ = _fit(3, X, 8, iters, 0.0)
        it0, p0, e0 = _fit(0, X, 8, iters, 0.0)
        assert e0 == 4 and it0 == it3 == iters
        assert np.max(np.abs(p0[1] - p3[1])) < 2e-4


def test_whole_fit_frames_without_responsibility_and_the_flush_band(built_lib):
    """A frame far from every mixture has no surviving term: no responsibility, ln 1e-15 in the total (gmm.cc:482-498, :34-38)
    -- the kernel's own rule, against the other path.  A live frame within 110 nats of the underflow boundary is where the
    reference's flushes of PARTIAL products decide (gmm_flush.hip): the kernel raises its flag and the fit runs iteration at
    a time -- same bits as asking for that path."""
    rng = np.random.default_rng(9)
    X, cent = _data(rng, 500, 8, 13)
    X[::50] += 1000.0
    start = (np.full(8, 1.0 / 8), cent, np.full((8, 13), 0.9))
    it3, p3, e3 = _fit(3, X, 8, 4, 0.0, start=start)
    it0, p0, e0 = _fit(0, X, 8, 4, 0.0, start=start)
    assert e0 == 4 and it0 == it3 == 4
    assert np.max(np.abs(p0[0] - p3[0])) < 2e-5 and np.max(np.abs(p0[1] - p3[1])) < 2e-4 and np.max(np.abs(p0[2] - p3[2]) / p3[2]) < 1e-3
    X, cent = _data(rng, 500, 1, 13)
    X[::50] = (cent[0] + 9.1).astype(np.float32)          # 13 x (9.1 / 0.9)^2 / 2 = 664 nats down
    start = (np.ones(1), cent, np.full((1, 13), 0.9))
    it3, p3, e3 = _fit(3, X, 1, 1, 0.0, start=start)
    it0, p0, e0 = _fit(0, X, 1, 1, 0.0, start=start)
    assert e0 != 4 and e0 == e3
    assert all(np.array_equal(a, b) for a, b in zip(p0, p3))


def test_whole_fit_same_bits_on_every_run_and_limits(built_lib):
    rng = np.random.default_rng(13)
    X, _ = _data(rng, 2998, 16, 13)
    a = _fit(0, X, 16, 200, 0.01, 1, seed=3)
    b = _fit(0, X, 16, 200, 0.01, 1, seed=3)
    assert a[2] == b[2] == 4 and a[0] == b[0] and all(np.array_equal(x, y) for x, y in zip(a[1], b[1]))
    # beyond the kernel's shapes: 33 mixtures / 41 dims go to the float64 iteration engine (em_f64.hip), 8193 frames to an iteration per launch
    for n, K, D, want in ((3000, 33, 13, (5,)), (3000, 8, 41, (5,)), (8193, 4, 5, (1, 2, 3))):
        X, _ = _data(rng, n, K, D)
        it, _, eng = _fit(0, X, K, 2, 0.0)
        assert it == 2 and eng in want, (n, K, D, eng)


def test_whole_fit_progress_lines_and_the_legacy_symbol(built_lib, tmp_path):
    """verbosity 1 prints `iter i: ll x` after every second iteration (gmm.cc:641): the whole-fit kernel keeps the totals its
    stop rule saw and the host prints them afterwards -- the same iterations, the same totals to 1e-6 relative, as an
    iteration per launch prints while it runs.  Through the legacy symbol train_model (double**, pygmm.hh:33) in a fresh
    process, as the reference's binding calls it."""
    import os
    import subprocess
    import sys
    rng = np.random.default_rng(17)
    X, _ = _data(rng, 1500, 8, 13)
    xp = str(tmp_path / "X.npy")
    np.save(xp, X)
    code = ("import sys, numpy as np; sys.path.insert(0, %r)\n"
            "from speaker_recognition_amd import _lib\n"
            "from speaker_recognition_amd.pygmm import GMM\n"
            "_lib.set_option('em_stats_engine', int(sys.argv[2]))\n"
            "g = GMM(8, nr_iteration=40, verbosity=1, seed=5)\n"
            "it = g.fit(np.load(sys.argv[1]))\n"
            "sys.stdout.flush(); print('done', it, _lib.last_em_stats_engine())\n") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = {}
    for eng in (0, 3):
        r = subprocess.run([sys.executable, "-c", code, xp, str(eng)], capture_output=True, text=True, check=True, timeout=300)
        lines = r.stdout.splitlines()
        out[eng] = ([(int(l.split()[1].rstrip(":")), float(l.split()[3])) for l in lines if l.startswith("iter ")],
                    [l for l in lines if l.startswith("done")][0].split())
    assert out[0][1][2] == "4" and out[3][1][2] in ("1", "2", "3")
    assert out[0][1][1] == out[3][1][1]                                     # iterations carried out
    assert len(out[0][0]) == len(out[3][0]) >= 1 and [i for i, _ in out[0][0]] == [i for i, _ in out[3][0]]
    assert all(i % 2 == 1 for i, _ in out[0][0])
    for (_, a), (_, b) in zip(out[0][0], out[3][0]):
        assert abs(a - b) <= 1e-6 * abs(b), (a, b)


def test_whole_fit_gives_up_when_a_workgroup_never_arrives(built_lib):
    """The whole-fit kernel is an ordinary launch whose workgroups meet at a grid-wide barrier; should part of the grid never
    start (other processes holding the chip), the workgroups that wait give the grid up after ~0.1 s of polling and the fit
    runs an iteration per launch instead.  Test hook (option debug_em_small_absent_workgroup): workgroup 1 stays away from
    the third barrier."""
    from speaker_recognition_amd import _lib
    rng = np.random.default_rng(19)
    X, _ = _data(rng, 1500, 8, 13)
    whole = _fit(0, X, 8, 6, 0.0, seed=5)
    per_launch = _fit(3, X, 8, 6, 0.0, seed=5)
    _lib.set_option("debug_em_small_absent_workgroup", 1)
    try:
        absent = _fit(0, X, 8, 6, 0.0, seed=5)
    finally:
        _lib.set_option("debug_em_small_absent_workgroup", 0)
    assert whole[2] == 4 and absent[2] != 4 and absent[2] == per_launch[2]
    assert absent[0] == per_launch[0] == whole[0] == 6
    assert all(np.array_equal(a, b) for a, b in zip(absent[1], per_launch[1]))          # the other path from the start: its bits
    again = _fit(0, X, 8, 6, 0.0, seed=5)                                                # (the next fit is a whole one again)
    assert again[2] == 4 and all(np.array_equal(a, b) for a, b in zip(again[1], whole[1]))


def test_whole_fits_from_several_processes_at_once(built_lib):
    """Four processes enrol on the one device at the same time (scripts/debug/em_small_stress.py): every fit comes back whole,
    with the bits of the process's first one.  What this pins: a workgroup that another process's kernel keeps off the chip in
    the middle of adding up iteration t's partial sums must not see a faster workgroup's sums of t + 1 (two sets of partials,
    by the iteration's parity -- with one set, 3 of 4 processes got different models from fit to fit), and whole-fit grids of
    different processes do not meet on the chip (a per-device advisory lock)."""
    import os
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "debug", "em_small_stress.py"), "4", "10"],
                       capture_output=True, text=True, timeout=300)
    line = [l for l in r.stdout.splitlines() if "whole fits" in l]
    assert r.returncode == 0 and line, (r.returncode, r.stdout[-500:], r.stderr[-500:])
    m = re.search(r"whole fits (\d+) of (\d+), fits whose bits differ from the process's first (\d+)", line[-1])
    assert m and m.group(1) == m.group(2) == "40" and m.group(3) == "0", line[-1]
