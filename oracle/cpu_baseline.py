"""oracle/cpu_baseline.py -- TEST / MEASUREMENT INFRASTRUCTURE ONLY.

The CPU leg of bench.py: on a bounded sample of the bench workload it times

* the reference's own compiled C++ (oracle/_ref/pygmm_ref.so, ``score_batch`` through its C ABI,
  pygmm.cc:104-117 -> gmm.cc:533-578) with ``concurrency`` = host cores AND with ``concurrency`` = 1
  (the latter on a sub-sample of the models, extrapolated linearly);
* the float64 numpy restatement of its MFCC.py (oracle/mfcc_oracle.py) single-process AND through
  ``multiprocessing.Pool(nproc)`` over utterances (what test-gmm.py:128-133 / :207-212 does);

* and -- "pool_utts" > 0 -- the reference at its best on this host: the way its own drivers use the
  cores, a multiprocessing.Pool over utterances (src/test/test-gmm.py:128-133), here over (utterance,
  group of models) tasks so that every core has work: each worker loads the reference DSO and the
  models it needs once and calls ``score_batch(..., concurrency = 1)``; features from the Pool'd MFCC
  leg; ALL the models, >= 64 utterances.  Its rate is what bench.py reports as ``cpu_baseline.value``
  (the single-call concurrency = cores figure stays beside it), and its per-utterance sums / argmax
  are the end-to-end parity sample (reference MFCC.py restatement -> reference DSO);

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

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import hostinfo  # noqa: E402

hostinfo.single_thread_blas()          # every pool worker does its own BLAS calls: one thread each

import numpy as np  # noqa: E402

_SPEC = None
_POOL_FEATS = None        # the Pool leg's utterances (float64 [T, D] each), inherited by its forked workers
_POOL_PATHS = None
_POOL_STATE = {}          # per worker: the reference library and the model handles it has loaded


def _extract(p):
    from oracle import mfcc_oracle as mo
    return mo.extract(_SPEC["fs"], p, diff=_SPEC["nd"] > 0, nd=max(1, _SPEC["nd"]), **_SPEC["mfcc_kw"])


def _pool_score(task):
    """(utterance, first model, one past the last) -> (utterance, first model, [sum of per-frame LL per model])"""
    from oracle import gmm_oracle as go
    u, s0, s1 = task
    st = _POOL_STATE
    if "ref" not in st:
        st["ref"] = go.RefLib()
        st["handles"] = {}
        st["rows"] = {}
    ref = st["ref"]
    if u not in st["rows"]:
        st["rows"] = {u: ref.rows(_POOL_FEATS[u])}              # (one utterance's row pointers at a time)
    rows, keep = st["rows"][u]
    n, d = keep.shape
    out = np.empty(n)
    outp = out.ctypes.data_as(go.C.POINTER(go.C.c_double))
    sums = []
    for s in range(s0, s1):
        h = st["handles"].get(s)
        if h is None:
            h = st["handles"][s] = ref.load(_POOL_PATHS[s])
        ref.lib.score_batch(h, rows, outp, n, d, 1)
        sums.append(float(out.sum()))
    return u, s0, sums


def pool_leg(spec, cores):
    """The reference DSO over (utterance, model group) tasks on a Pool of all cores + the MFCC restatement on the same Pool size."""
    global _POOL_FEATS, _POOL_PATHS
    from oracle import gmm_oracle as go
    from speaker_recognition_amd import synth
    U = int(spec["pool_utts"])
    n_spk = spec.get("n_speakers", 1)
    pcm = [synth.synth_speech(u % n_spk, spec["seconds"], spec["fs"], seed=spec["seed"] + u) for u in range(U)]
    procs = max(1, min(cores, U))
    with mp.get_context("fork").Pool(procs) as pool:
        pool.map(_extract, pcm[:procs])                           # start-up outside the timed part
        t0 = time.perf_counter()
        feats = pool.map(_extract, pcm)
        t_mfcc = time.perf_counter() - t0
    # (the device scores fp32 features; so does this leg, as the other one)
    _POOL_FEATS = [np.ascontiguousarray(f.astype(np.float32).astype(np.float64)) for f in feats]
    _POOL_PATHS = list(spec["pool_model_files"])
    S = len(_POOL_PATHS)
    n_frames = int(sum(len(f) for f in _POOL_FEATS))
    if not os.path.exists(go.REF_SO):
        return None
    # model groups: about 4 tasks per core, no group larger than 16 models (a straggler's length)
    groups_per_utt = max(1, min(S, -(-4 * cores // U)))
    gsz = max(1, min(16, -(-S // groups_per_utt)))
    tasks = [(u, s0, min(S, s0 + gsz)) for u in range(U) for s0 in range(0, S, gsz)]
    procs = max(1, min(cores, len(tasks)))
    sums = np.zeros((U, S))
    with mp.get_context("fork").Pool(procs) as pool:
        pool.map(_pool_score, [(u % U, 0, 1) for u in range(procs)], chunksize=1)     # DSO load + first model, untimed
        t0 = time.perf_counter()
        for u, s0, part in pool.imap_unordered(_pool_score, tasks, chunksize=1):
            sums[u, s0:s0 + len(part)] = part
        t_gmm = time.perf_counter() - t0
    return {"utterances": U, "models": S, "frames": n_frames, "processes": procs, "tasks": len(tasks), "models_per_task": gsz,
            "t_mfcc_pool_s": t_mfcc, "t_gmm_pool_s": t_gmm, "frames_per_s": n_frames / (t_mfcc + t_gmm),
            "gmm_frames_per_s_all_models": n_frames / t_gmm, "mfcc_frames_per_s": n_frames / t_mfcc,
            "sums": sums.tolist(), "argmax": np.argmax(sums, axis=1).tolist()}


def main():
    global _SPEC
    spec = _SPEC = json.loads(sys.argv[1])
    from oracle import gmm_oracle as go
    from speaker_recognition_amd import synth

    fs = spec["fs"]
    n_utt, seconds = spec["n_utt"], spec["seconds"]
    cores = hostinfo.effective_cores()            # what this container may use (cgroup quota / affinity), not what the machine has
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

    # the Pool leg first: this process has not loaded the reference DSO yet (its -ffast-math start-up code switches FTZ / DAZ on
    # for the thread, and forked workers would inherit that into their MFCC arithmetic)
    pool = None
    if spec.get("pool_utts", 0) > 0 and spec.get("pool_model_files"):
        try:
            pool = pool_leg(spec, cores)
        except Exception as e:                                    # the single-call leg still stands
            pool = {"error": "%s: %s" % (type(e).__name__, e)}

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
        "host": hostinfo.describe(),
        "pool": pool,
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
