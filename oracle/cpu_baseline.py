"""oracle/cpu_baseline.py -- TEST / MEASUREMENT INFRASTRUCTURE ONLY.

The CPU leg of bench.py: on a bounded sample of the bench workload it times

* the reference's own compiled C++ (oracle/_ref/pygmm_ref.so, ``score_batch`` through its C ABI,
  pygmm.cc:104-117 -> gmm.cc:533-578) with ``concurrency`` = host cores AND with ``concurrency`` = 1
  (the latter on a sub-sample of the models, extrapolated linearly);
* the float64 numpy restatement of its MFCC.py (oracle/mfcc_oracle.py) single-process AND through
  ``multiprocessing.Pool(nproc)`` over utterances (what test-gmm.py:128-133 / :207-212 does);

and prints one JSON line with the per-utterance sums and argmax, so that bench.py can compare the
device path on exactly this sample.  Run as a subprocess so that the reference DSO's -ffast-math
FTZ/DAZ switch and its thread pool stay out of the benchmark process.  Falls back to the C
restatement (kind "port", 1 core) when the reference DSO is not present.

spec keys: fs, mfcc_kw, nd, n_utt, seconds, seed (audio: synth.synth_speech(u % n_speakers, ...)),
n_speakers, and either model_files (text-format model files written by the bench) or
n_models/n_mix/dim/model_seed (synth.synth_gmm); single_core_models (how many models the
concurrency = 1 leg scores).
"""
import json
import multiprocessing as mp
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

_SPEC = None


def _extract(p):
    from oracle import mfcc_oracle as mo
    return mo.extract(_SPEC["fs"], p, diff=_SPEC["nd"] > 0, nd=max(1, _SPEC["nd"]), **_SPEC["mfcc_kw"])


def main():
    global _SPEC
    spec = _SPEC = json.loads(sys.argv[1])
    from oracle import gmm_oracle as go
    from speaker_recognition_amd import synth

    fs = spec["fs"]
    n_utt, seconds = spec["n_utt"], spec["seconds"]
    cores = os.cpu_count() or 1
    n_spk = spec.get("n_speakers", spec.get("n_models", 1))
    pcm = [synth.synth_speech(u % n_spk, seconds, fs, seed=spec["seed"] + u) for u in range(n_utt)]

    t0 = time.perf_counter()
    feats = [_extract(p) for p in pcm]
    t_mfcc = time.perf_counter() - t0
    n_frames = int(sum(len(f) for f in feats))
    # Pool(nproc) over utterances, as the reference's test drivers do (fork: workers inherit the spec)
    procs = min(cores, max(1, n_utt))
    t_pool = None
    try:
        with mp.get_context("fork").Pool(procs) as pool:
            pool.map(_extract, pcm[:procs])                       # start-up outside the timed part
            reps = max(1, (4 * procs) // max(1, n_utt))
            t0 = time.perf_counter()
            for _ in range(reps):
                pool.map(_extract, pcm)
            t_pool = (time.perf_counter() - t0) / reps
    except Exception:
        t_pool = None
    X = np.ascontiguousarray(np.concatenate(feats).astype(np.float32).astype(np.float64))
    off = np.concatenate([[0], np.cumsum([len(f) for f in feats])])

    tmp = tempfile.mkdtemp()
    if spec.get("model_files"):
        paths = list(spec["model_files"])
        params = [go.parse_model_text(open(p).read()) for p in paths]
    else:
        params, paths = [], []
        for s in range(spec["n_models"]):
            p = go.GMMParams(*synth.synth_gmm(spec["n_mix"], spec["dim"], spec["model_seed"] + s))
            path = os.path.join(tmp, "m%d.model" % s)
            with open(path, "w") as f:
                f.write(go.format_model_text(p))
            params.append(p)
            paths.append(path)
    S, D = len(params), params[0].D
    sums = np.zeros((n_utt, S))
    use_ref = os.path.exists(go.REF_SO)
    t_gmm_1 = None
    if use_ref:
        ref = go.RefLib()
        handles = [ref.load(p) for p in paths]
        rows, keep = ref.rows(X)
        out = np.empty(len(X))
        outp = out.ctypes.data_as(go.C.POINTER(go.C.c_double))
        t0 = time.perf_counter()
        for s, h in enumerate(handles):
            ref.lib.score_batch(h, rows, outp, len(X), D, cores)
            sums[:, s] = [out[off[u]:off[u + 1]].sum() for u in range(n_utt)]
        t_gmm = time.perf_counter() - t0
        n1 = max(1, min(S, int(spec.get("single_core_models", 2))))
        n1_frames = int(off[min(n_utt, 2)])                       # two utterances are enough for a rate
        rows1, keep1 = ref.rows(X[:n1_frames])
        t0 = time.perf_counter()
        for h in handles[:n1]:
            ref.lib.score_batch(h, rows1, outp, n1_frames, D, 1)
        t_gmm_1 = (time.perf_counter() - t0) * (S / n1) * (len(X) / max(1, n1_frames))   # scaled to the sample
        kind, used = "reference", cores
    else:
        t0 = time.perf_counter()
        for s, p in enumerate(params):
            ll = go.score_batch(p, X)
            sums[:, s] = [ll[off[u]:off[u + 1]].sum() for u in range(n_utt)]
        t_gmm = time.perf_counter() - t0
        kind, used = "port", 1
    print(json.dumps({
        "kind": kind, "cores": used, "n_frames": n_frames, "n_models": S,
        "t_mfcc_s": t_mfcc, "t_mfcc_pool_s": t_pool, "pool_procs": procs, "t_gmm_s": t_gmm, "t_gmm_1core_s": t_gmm_1,
        "frames_per_s": n_frames / ((t_pool if t_pool else t_mfcc) + t_gmm),
        "mfcc_frames_per_s_1proc": n_frames / t_mfcc,
        "mfcc_frames_per_s_pool": (n_frames / t_pool) if t_pool else None,
        "gmm_frames_per_s": n_frames / t_gmm,
        "gmm_frames_per_s_1core": (n_frames / t_gmm_1) if t_gmm_1 else None,
        "argmax": np.argmax(sums, axis=1).tolist(), "sums": sums.tolist(),
    }))


if __name__ == "__main__":
    main()
