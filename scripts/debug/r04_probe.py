#!/usr/bin/env python3
"""Round-4 probes on the GPU box: (1) how many host cores this box really gives us, (2) where the device MFCC differs most from the
float64 oracle on the bench's own audio, (3) sr_multi_predict_pcm against the number of pieces."""
import os, sys, time, subprocess
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
what = sys.argv[1:] or ["cores", "mfcc", "multi"]
if "cores" in what:
    print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
    for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
        if os.path.exists(f):
            print(f, open(f).read().strip())
    import multiprocessing as mp
    def burn(_):
        t0 = time.perf_counter(); x = 0
        for i in range(3_000_000): x += i * i
        return time.perf_counter() - t0
    for n in (1, 8, 32, 128, 256):
        with mp.get_context("fork").Pool(n) as pool:
            t0 = time.perf_counter(); ts = pool.map(burn, range(n)); el = time.perf_counter() - t0
        print("procs %3d: wall %.2f s, per-task %.2f s -> speedup vs serial %.1f" % (n, el, float(np.mean(ts)), n * ts[0] / el if n == 1 else n * base / el), flush=True) if n > 1 else None
        if n == 1: base = ts[0]; print("1 proc: %.2f s" % base)
import bench
from speaker_recognition_amd import _lib, synth
from speaker_recognition_amd.core import Batch, MfccExtractor, ModelSet, MultiPredictor
from speaker_recognition_amd.pygmm import GMM
ex = MfccExtractor(bench.FS, **bench.MFCC_KW)
n_samples = (bench.FRAMES_PER_UTT + bench.ND - 1) * ex.FRAME_SHIFT + ex.FRAME_LEN
base_c = bench.base_clips(bench.CFG1_MODELS, n_samples)
if "mfcc" in what:
    from oracle import mfcc_oracle as mo
    cat, off = bench.make_pcm(base_c, 200, 0)
    fb = ex.extract_batch(Batch.from_pcm((cat, off)), nd=2)
    X = fb.download(); fo = fb.offsets()
    raw = ex.extract_batch(Batch.from_pcm((cat, off)), nd=0, cmvn=False).download()
    for u in range(200):
        p = cat[off[u]:off[u + 1]]
        ref = mo.extract(bench.FS, p, diff=True, nd=2, **bench.MFCC_KW)
        d = np.abs(X[fo[u]:fo[u + 1]] - ref)
        i = np.unravel_index(np.argmax(d), d.shape)
        if u < 2 or d.max() > 1e-3:
            rr = mo.get_mfcc_extractor(bench.FS, **bench.MFCC_KW).raw_cepstra(p.astype(float))
            print("utt %2d: max %.3e at frame %d dim %d (dev %.5f ref %.5f) mean %.2e; per-block max: c %.2e d1 %.2e d2 %.2e" % (
                u, d.max(), i[0], i[1], X[fo[u] + i[0], i[1]], ref[i], d.mean(), d[:, :13].max(), d[:, 13:26].max(), d[:, 26:].max()), flush=True)
            if rr is not None:
                T = rr.shape[0]
                dr = np.abs(raw[u * T:(u + 1) * T] - rr)
                j = np.unravel_index(np.argmax(dr), dr.shape)
                print("        raw cepstra: max %.3e at frame %d coef %d (dev %.5f ref %.5f); column std of ref %s" % (
                    dr.max(), j[0], j[1], raw[u * T + j[0], j[1]], rr[j], np.array2string(rr.std(axis=0), precision=2)))
if "multi" in what:
    gm = [GMM.from_arrays(*synth.synth_gmm(bench.CFG1_MIX, bench.DIM, bench.MODEL_SEED + s)) for s in range(bench.CFG1_MODELS)]
    cat, off = bench.make_pcm(base_c, bench.CFG1_UTTS, 0)
    ms = ModelSet(gm)
    pcm = Batch.from_pcm((cat, off))
    step = lambda: ex.predict_batch(ms, pcm, nd=2)
    step(); step()
    el, _ = bench.timed(step, 0, 10, _lib.synchronize)
    print("resident step %.2f ms" % (1e3 * el / 10))
    _lib.host_register(cat)
    for pieces in (1, 2, 4, 8):
        _lib.set_option("multi_pieces", pieces)
        mp_ = MultiPredictor(gm, bench.FS, n_slots=1, **bench.MFCC_KW)
        f = lambda: mp_.predict_concat(cat, off, nd=2)
        f(); f()
        _lib.profile_enable(True); _lib.profile_reset()
        el, _ = bench.timed(f, 0, 10)
        kt = {k: _lib.profile_get(v)[0] / 10 for k, v in (("mfcc", _lib.T_MFCC), ("cmvn", _lib.T_CMVN), ("score", _lib.T_SCORE), ("fin", _lib.T_FINALIZE))}
        print("pieces %d: %.2f ms per call; kernels per call (ms): %s  %s" % (pieces, 1e3 * el / 10, {k: round(v, 3) for k, v in kt.items()}, _lib.last_score_kernel()[:60]), flush=True)
        _lib.profile_enable(False)
        del mp_
    _lib.host_unregister(cat)
