#!/usr/bin/env python3
"""Generate the committed known-answer vectors from the REFERENCE ITSELF.

Run in the build container only (needs /root/reference):

    make -C oracle ref && python tests/golden/make_golden.py

* GMM scoring: loads ``oracle/_ref/pygmm_ref.so`` (the reference's C++ compiled from
  /root/reference/src/gmm/src by oracle/Makefile) and records ``score_batch`` /
  ``score_all`` (pygmm.hh:36-37) on the reference's shipped UBM fixtures
  (src/gui/model/*.model) and on synthetic models written through its own text format.
  The model parameters are stored as parsed float64 arrays (not as copies of the files).
* MFCC: executes the reference's ``src/feature/MFCC.py`` + ``utils.py`` text in memory
  after the mechanical Python-2 -> 3 edits listed in SURVEY.md section 8c (nothing is
  written back; no reference source is copied into the repo) on seeded synthetic audio.

Outputs: tests/golden/gmm_golden.npz, tests/golden/mfcc_golden.npz.
"""
import os
import sys
import tempfile
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF = "/root/reference"

from oracle import gmm_oracle as go  # noqa: E402
from speaker_recognition_amd import synth  # noqa: E402


def frames_for(params, n, seed, outliers=2):
    rng = np.random.default_rng(seed)
    w = params.weights / params.weights.sum()
    k = rng.choice(params.K, size=n, p=w)
    x = params.mean[k] + params.sigma[k] * rng.standard_normal((n, params.D))
    x[n - outliers:] += 60.0                      # underflow -> safe_log clamp (gmm.cc:34-38)
    return x.astype(np.float32).astype(np.float64)  # fp32-representable: both sides see identical inputs


def gmm_golden():
    ref = go.RefLib()
    out = {}
    cases = []
    for name, fname in (("ubm32", "ubm.mixture-32.utt-300.model"),
                        ("ubm64", "ubm.mixture-64.utt-300.model"),
                        ("ubm256", "ubm.mixture-256.nperson-300.immature.model")):
        path = os.path.join(REF, "src/gui/model", fname)
        cases.append((name, path, go.parse_model_text(open(path).read())))
    tmpdir = tempfile.mkdtemp()
    for name, K, D, seed in (("syn16x13", 16, 13, 7), ("syn64x39", 64, 39, 8), ("syn5x3", 5, 3, 9)):
        w, mu, sg = synth.synth_gmm(K, D, seed)
        p = go.GMMParams(w, mu, sg)
        path = os.path.join(tmpdir, name + ".model")
        with open(path, "w") as f:
            f.write(go.format_model_text(p))
        # what the reference's loader will see == what our parser sees
        cases.append((name, path, go.parse_model_text(open(path).read())))
    for i, (name, path, p) in enumerate(cases):
        h = ref.load(path)
        assert ref.lib.get_nr_mixtures(h) == p.K and ref.lib.get_dim(h) == p.D
        X = frames_for(p, 48, 100 + i)
        ll1 = ref.score_batch(h, X, 1)
        ll3 = ref.score_batch(h, X, 3)           # threads only partition frames (gmm.cc:536-553)
        assert np.array_equal(ll1, ll3)
        out[name + "_w"] = p.weights
        out[name + "_mean"] = p.mean
        out[name + "_sigma"] = p.sigma
        out[name + "_X"] = X
        out[name + "_ll"] = ll1
        out[name + "_sum"] = np.float64(ref.score_all(h, X, 2))
    # ---- MAP adaptation (train_model_from_ubm, pygmm.hh:34): deterministic in the reference (the
    # trainer starts from a copy of the UBM, gmmubm.cc:25-38), so its result pins the E-step
    # responsibilities, N_k and the means-only update.  Models come back through dump() (6 digits).
    import ctypes as C
    from speaker_recognition_amd._lib import Parameter
    name, path, p = [c for c in cases if c[0] == "syn16x13"][0]
    ubm_h = ref.load(path)
    rng = np.random.default_rng(77)
    k = rng.choice(p.K, size=700, p=p.weights / p.weights.sum())
    Xa = (p.mean[k] + 0.25 + 0.8 * p.sigma[k] * rng.standard_normal((700, p.D))).astype(np.float32).astype(np.float64)
    out["map_X"] = Xa
    saved = os.dup(1)
    devnull = os.open(os.devnull, os.O_WRONLY)
    for iters in (1, 4):
        h = ref.lib.new_gmm(p.K, 1)
        par = Parameter(nr_instance=len(Xa), nr_dim=p.D, nr_mixture=p.K, min_covar=1e-3, threshold=0.01,
                        nr_iteration=iters, init_with_kmeans=0, concurrency=2, verbosity=0)
        rows, keep = ref.rows(Xa)
        sys.stdout.flush()
        os.dup2(devnull, 1)                      # the reference prints its parameter block
        try:
            ref.lib.train_model_from_ubm(h, ubm_h, rows, C.byref(par))
        finally:
            os.dup2(saved, 1)
        dump_path = os.path.join(tmpdir, "map%d.model" % iters)
        ref.lib.dump(h, dump_path.encode())
        q = go.parse_model_text(open(dump_path).read())
        out["map%d_mean" % iters] = q.mean
        out["map%d_w" % iters] = q.weights
        out["map%d_sigma" % iters] = q.sigma
    for f in ("gmm-training-intermediate-dump.model",):
        if os.path.exists(f):
            os.remove(f)
    # ---- full EM (train_model, pygmm.hh:33).  The reference draws its initial means with a
    # rand()-seeded engine; in a FRESH process with the same call sequence that draw repeats, so
    # "nr_iteration = 0" (initialisation only, gmm.cc:591-592) in one process and "nr_iteration = N"
    # in another give the start and the end of the same run.  The start is stored exactly: the
    # picked rows are identified by matching the dumped (6-digit) means back to rows of X.
    import subprocess
    true = synth.synth_gmm(8, 13, 31)
    Xe = synth.draw_frames(true, 1500, 32).astype(np.float64)
    xpath = os.path.join(tmpdir, "em_X.npy")
    np.save(xpath, Xe)
    helper = os.path.join(ROOT, "tests/golden/_ref_train.py")

    def ref_train(iters):
        mp = os.path.join(tmpdir, "em_%d.model" % iters)
        subprocess.run([sys.executable, helper, xpath, "8", str(iters), mp], check=True, stdout=subprocess.DEVNULL)
        return go.parse_model_text(open(mp).read())

    init = ref_train(0)
    init_again = ref_train(0)
    assert np.array_equal(init.mean, init_again.mean), "reference initialisation is not repeatable across processes"
    rows_picked = [int(np.argmin(np.sum((Xe - m) ** 2, axis=1))) for m in init.mean]
    assert np.max(np.abs(Xe[rows_picked] - init.mean) / np.maximum(1, np.abs(init.mean))) < 1e-5
    out["em_X"] = Xe
    out["em_init_rows"] = np.array(rows_picked)
    out["em_init_sigma_dumped"] = init.sigma
    for iters in (1, 2, 6):
        q = ref_train(iters)
        out["em%d_w" % iters], out["em%d_mean" % iters], out["em%d_sigma" % iters] = q.weights, q.mean, q.sigma
    out["cases"] = np.array([c[0] for c in cases])
    np.savez_compressed(os.path.join(ROOT, "tests/golden/gmm_golden.npz"), **out)
    print("gmm_golden.npz:", [c[0] for c in cases])


def load_reference_mfcc():
    """Exec the reference MFCC.py under Python 3 (four mechanical edits + xrange)."""
    utils_src = open(os.path.join(REF, "src/feature/utils.py")).read()
    utils = types.ModuleType("utils")
    exec(compile(utils_src, "ref:utils.py", "exec"), utils.__dict__)
    sys.modules["utils"] = utils
    src = open(os.path.join(REF, "src/feature/MFCC.py")).read()

    def sub(old, new, count):
        nonlocal src
        assert src.count(old) == count, (old, src.count(old))
        src = src.replace(old, new)

    sub("xrange", "range", 3)
    sub("(len(signal) - self.FRAME_LEN) / self.FRAME_SHIFT + 1",
        "(len(signal) - self.FRAME_LEN) // self.FRAME_SHIFT + 1", 1)          # MFCC.py:57
    sub("[:self.FFT_SIZE / 2 + 1]", "[:self.FFT_SIZE // 2 + 1]", 1)           # MFCC.py:66
    sub("cast['float'](signal)", "signal.astype(float)", 1)                   # MFCC.py:128
    sub("b4 = min(fn2,", "b4 = _builtin_min(fn2,", 1)                         # MFCC.py:94
    mod = types.ModuleType("ref_MFCC")
    mod.__dict__["_builtin_min"] = min
    exec(compile(src, "ref:MFCC.py", "exec"), mod.__dict__)
    return mod, utils


def mfcc_golden():
    ref, utils = load_reference_mfcc()
    out = {}
    names = []
    cases = [
        # name, fs, seconds, speaker, kwargs
        ("default16k", 16000, 1.2, 0, {}),
        ("cfg0_16k", 16000, 1.0, 3, dict(win_length_ms=25, win_shift_ms=10)),
        ("fft512_16k", 16000, 1.0, 5, dict(win_length_ms=25, win_shift_ms=10, FFT_SIZE=512, n_filters=40)),
        ("default8k", 8000, 1.5, 7, {}),
        ("cfg5_8k", 8000, 1.0, 9, dict(win_length_ms=25, win_shift_ms=10, FFT_SIZE=256, n_filters=24, n_ceps=12)),
    ]
    for name, fs, sec, spk, kw in cases:
        pcm = synth.synth_speech(spk, sec, fs)
        if name == "cfg0_16k":
            pcm[4000:4800] = 0                      # digital silence: exercises the 1e-100 floor (MFCC.py:67)
        ex = ref.get_mfcc_extractor(fs, **kw)
        feat = ref.extract(fs, pcm, **kw)
        # raw cepstra before CMVN: rerun the loop body through the reference's own matrices
        sig = pcm.astype(float)
        raw = []
        for f in range((len(sig) - ex.FRAME_LEN) // ex.FRAME_SHIFT + 1):
            frame = sig[f * ex.FRAME_SHIFT: f * ex.FRAME_SHIFT + ex.FRAME_LEN] * ex.window
            frame[1:] -= frame[:-1] * ex.PRE_EMPH
            X = abs(np.fft.fft(frame, ex.FFT_SIZE)[:ex.FFT_SIZE // 2 + 1]) ** 2
            X[X < ref.POWER_SPECTRUM_FLOOR] = ref.POWER_SPECTRUM_FLOOR
            raw.append(np.dot(ex.D, np.log(np.dot(ex.M, X))))
        raw = np.vstack(raw)
        mu, sd = raw.mean(axis=0), raw.std(axis=0)
        assert np.allclose((raw - mu) / sd, feat, rtol=0, atol=1e-9)
        out[name + "_pcm"] = pcm
        out[name + "_fs"] = np.int64(fs)
        out[name + "_kw"] = np.array(repr(kw))
        out[name + "_raw"] = raw
        out[name + "_feat"] = feat
        out[name + "_d1"] = ref.extract(fs, pcm, diff=True, **kw)           # utils.py:24-31, nd=1
        out[name + "_d2"] = utils.diff_feature(feat, 2)                     # nd=2 -> 3x dims
        names.append(name)
    # constants that define the chain (small ones in full; the bank as its nonzero pattern)
    ex = ref.get_mfcc_extractor(16000)
    out["hamming4"] = ref.hamming(4)
    out["default16k_window"] = ex.window
    out["default16k_D"] = ex.D
    out["default16k_M_nnz_rows"], out["default16k_M_nnz_cols"] = np.nonzero(ex.M)
    out["default16k_M_vals"] = ex.M[np.nonzero(ex.M)]
    out["cases"] = np.array(names)
    np.savez_compressed(os.path.join(ROOT, "tests/golden/mfcc_golden.npz"), **out)
    print("mfcc_golden.npz:", names)


if __name__ == "__main__":
    mfcc_golden()   # first: loading the -ffast-math reference DSO flips the process to FTZ/DAZ
    gmm_golden()
