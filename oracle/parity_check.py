"""oracle/parity_check.py -- TEST / MEASUREMENT INFRASTRUCTURE ONLY.

Checker leg of bench.py.  Reads a pickle {name: spec} with

    "models"            [(w, mean, sigma), ...]            or
    "models_recipe"     {"kind": "cfg3", ...} / {"kind": "cfg2", ...}   (regenerated here with the product's own
                        speaker_recognition_amd.synth, seed for seed -- 1001 x 2048 x 39 doubles do not travel well)
    "X"                 float[n, D]  the frames the DEVICE scored (its own MFCC output, or drawn features)
    "offsets"           int[U + 1]
    "device_sums"       float64[U, S]   (optional)   "device_frame_ll"  float32[S, n]  (optional)

or, for the FEATURE stage (spec key "kind": "mfcc"),

    "pcm"               int16[n_samples]   concatenated utterances     "sample_offsets"  int[U + 1]
    "fs", "mfcc_kw", "nd"                  the extractor's parameters
    "device_feats"      float32[n_frames, D]  what the device produced (MFCC -> CMVN -> deltas)   "offsets" int[U + 1]

runs the float64 numpy restatement of the reference's MFCC.py / utils.py (oracle/mfcc_oracle.py) on every utterance -- Pool over
utterances, as src/test/test-gmm.py:207-212 -- and reports max / mean |device - oracle| (SURVEY.md 8d gates: 1e-3 / 1e-5).

END TO END FROM PCM (spec key "kind": "pcm_ll"): "pcm" / "sample_offsets" / "fs" / "mfcc_kw" / "nd" as above, the models (optionally
"model_index": the subset that was scored), and the device's per-frame log-likelihoods "device_frame_ll" float32[S', n] computed from the
PCM by its own feature stage.  The checker builds the features with the float64 oracle (never rounded to float32), scores them with the
reference's arithmetic, and compares per frame: north_star's criterion, |d| <= 1e-4 max(1, |LL|) on identical INPUTS (MFCC.py:49-79 ->
utils.py:24-31 -> gmm.cc:237-244), clamp decisions, per-utterance sums and argmax.

The GMM specs: scores every frame under every model with the C restatement of the reference's arithmetic (oracle/gmm_oracle.c mode 3 =
mode 0, what the reference's C ABI computes, to remez5's 1.2e-6: score_batch_fast) on ALL host cores --
multiprocessing.Pool over (model, frame range) tasks, the shape of the reference's own test driver
(src/test/test-gmm.py:128-133) -- and prints one JSON line {name: {...}}: worst per-frame and per-utterance relative
differences, frames whose clamp decision differs, argmax mismatches.  Runs as a subprocess: the benchmark process itself
never imports oracle/."""
import json
import multiprocessing as mp
import os
import pickle
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import hostinfo  # noqa: E402

hostinfo.single_thread_blas()

import numpy as np  # noqa: E402

_REQ = {}          # name -> spec with "models" materialised; inherited by the forked workers
FRAME_CHUNK = 20000


def _models_from_recipe(rc):
    from speaker_recognition_amd import synth
    if rc["kind"] == "cfg3":       # bench.py block_cfg3
        ubm = synth.synth_gmm(rc["K"], rc["dim"], rc["ubm_seed"])
        w, mean, sigma = ubm
        alpha = ((w * 40.0 * rc["K"]) / (w * 40.0 * rc["K"] + 16.0))[:, None]
        out = [ubm]
        for s in range(rc["S"]):
            rng = np.random.default_rng(rc["spk_seed"] + s)
            out.append((w, mean + alpha * 0.3 * rng.standard_normal(mean.shape), sigma))
        return out
    if rc["kind"] == "cfg2":       # bench.py headline: synth.synth_map_speaker
        ubm = synth.synth_gmm(rc["K"], rc["dim"], rc["ubm_seed"])
        return [ubm] + [synth.synth_map_speaker(ubm, rc["spk_seed"] + s) for s in range(rc["S"])]
    raise ValueError(rc["kind"])


def _task(t):
    """(name, model, frame range) -> worst per-frame relative difference, clamp mismatches, per-utterance partial sums"""
    from oracle import gmm_oracle as go
    name, s, f0, f1 = t
    r = _REQ[name]
    p = go.GMMParams(*[np.asarray(a, dtype=np.float64) for a in r["models"][s]])
    ll = go.score_batch(p, r["X"][f0:f1], go.MODE_FAST)
    worst, clamp_bad = 0.0, 0
    dev = r.get("device_frame_ll")
    if dev is not None:
        d = dev[s, f0:f1].astype(np.float64)
        worst = float(np.max(np.abs(d - ll) / np.maximum(1.0, np.abs(ll)))) if len(ll) else 0.0
        clamp_bad = int(np.sum((dev[s, f0:f1] == np.float32(go.LN_1E_15)) != (ll == go.LN_1E_15)))
    off = r["offsets"]
    u0 = int(np.searchsorted(off, f0, side="right") - 1)
    parts = []
    u = u0
    while u < len(off) - 1 and off[u] < f1:
        a, b = max(off[u], f0), min(off[u + 1], f1)
        if b > a:
            parts.append((u, float(ll[a - f0:b - f0].sum())))
        u += 1
    return name, s, worst, clamp_bad, parts


def _mfcc_task(t):
    """(name, utterance) -> the device's features against mfcc_oracle (float64), and -- for scale -- what the SAME chain makes of the
    utterance when only its FFT runs in float32 (scipy's pocketfft): the floor any fp32 implementation has on this input.  Also the
    utterance's largest mel-band dynamic range within a frame: above ~90 dB the weak bands sit below the fp32 spectrum's resolution
    relative to the strong ones, and their logarithms -- divided by small column deviations in the CMVN -- carry the error."""
    import scipy.fft
    from oracle import mfcc_oracle as mo
    name, u = t
    r = _REQ[name]
    so, fo = r["sample_offsets"], r["offsets"]
    p = r["pcm"][so[u]:so[u + 1]].astype(np.float64)
    nd = r["nd"]
    ex = mo.get_mfcc_extractor(r["fs"], **r["mfcc_kw"])
    frames = ex.n_frames(len(p))
    idx = np.arange(frames)[:, None] * ex.FRAME_SHIFT + np.arange(ex.FRAME_LEN)[None, :]
    fr = p[idx] * ex.window[None, :]                                   # MFCC.py:61-64, as mfcc_oracle.raw_cepstra
    fr[:, 1:] = fr[:, 1:] - fr[:, :-1] * ex.PRE_EMPH

    def chain(power):
        power = np.maximum(power, mo.POWER_SPECTRUM_FLOOR)
        mel = np.dot(power, ex.M.T)
        c = np.dot(np.log(mel), ex.D.T)
        c = (c - c.mean(axis=0)) / c.std(axis=0)
        return (mo.diff_feature(c, nd) if nd > 0 else c), mel
    ref, mel = chain(np.abs(np.fft.fft(fr, ex.FFT_SIZE, axis=1)[:, :ex.FFT_SIZE // 2 + 1]) ** 2)
    f32, _ = chain(np.abs(scipy.fft.fft(fr.astype(np.float32), ex.FFT_SIZE, axis=1)[:, :ex.FFT_SIZE // 2 + 1]).astype(np.float64) ** 2)
    dyn_db = float(np.max(10.0 * np.log10(mel.max(axis=1) / mel.min(axis=1))))
    dev = r["device_feats"][fo[u]:fo[u + 1]].astype(np.float64)
    if ref.shape != dev.shape:
        return name, float("inf"), float("inf"), 1, float("inf"), dyn_db
    d = np.abs(dev - ref)
    return name, float(d.max()) if d.size else 0.0, float(d.sum()), int(d.size), float(np.abs(f32 - ref).max()), dyn_db


def _feat_task(t):
    """(name, utterance) -> float64 features of the oracle chain (MFCC -> CMVN -> deltas), never rounded to float32"""
    from oracle import mfcc_oracle as mo
    name, u = t
    r = _REQ[name]
    so = r["sample_offsets"]
    return name, u, mo.extract(r["fs"], r["pcm"][so[u]:so[u + 1]], diff=r["nd"] > 0, nd=max(1, r["nd"]), **r["mfcc_kw"])


def main():
    from oracle import gmm_oracle as go
    if not os.path.exists(go.ORACLE_SO):
        go.build(ref=False)
    req_all = pickle.load(open(sys.argv[1], "rb"))
    mfcc_req = {k: v for k, v in req_all.items() if v.get("kind") == "mfcc"}
    req = {k: v for k, v in req_all.items() if v.get("kind") != "mfcc"}
    mfcc_tasks = []
    for name, r in mfcc_req.items():
        r["pcm"] = np.asarray(r["pcm"])
        r["sample_offsets"] = np.asarray(r["sample_offsets"], dtype=np.int64)
        r["offsets"] = np.asarray(r["offsets"], dtype=np.int64)
        r["device_feats"] = np.asarray(r["device_feats"], dtype=np.float32)
        _REQ[name] = r
        mfcc_tasks += [(name, u) for u in range(len(r["sample_offsets"]) - 1)]
    cores = hostinfo.effective_cores()
    # end-to-end specs: their frames come from the float64 feature oracle, one utterance per task
    pcm_ll = [k for k, v in req.items() if v.get("kind") == "pcm_ll"]
    feat_tasks = []
    for name in pcm_ll:
        r = req[name]
        r["pcm"] = np.asarray(r["pcm"])
        r["sample_offsets"] = np.asarray(r["sample_offsets"], dtype=np.int64)
        _REQ[name] = r
        feat_tasks += [(name, u) for u in range(len(r["sample_offsets"]) - 1)]
    if feat_tasks:
        if cores > 1:
            with mp.get_context("fork").Pool(min(cores, len(feat_tasks))) as pool:
                feats = pool.map(_feat_task, feat_tasks, chunksize=1)
        else:
            feats = [_feat_task(t) for t in feat_tasks]
        for name in pcm_ll:
            mine = sorted((x for x in feats if x[0] == name), key=lambda x: x[1])
            r = req[name]
            r["X"] = np.concatenate([x[2] for x in mine], axis=0)
            own_off = np.concatenate([[0], np.cumsum([len(x[2]) for x in mine])])
            if not np.array_equal(own_off, np.asarray(r["offsets"], dtype=np.int64)):
                raise SystemExit("pcm_ll spec %s: the device's frame offsets differ from the oracle's" % name)
            del r["pcm"]
    tasks = []
    for name, r in req.items():
        if "models" not in r:
            r["models"] = _models_from_recipe(r["models_recipe"])
        if r.get("model_index") is not None:
            r["models"] = [r["models"][i] for i in r["model_index"]]
        r["X"] = np.ascontiguousarray(r["X"], dtype=np.float64)
        r["offsets"] = np.asarray(r["offsets"], dtype=np.int64)
        if r.get("device_frame_ll") is not None:
            r["device_frame_ll"] = np.asarray(r["device_frame_ll"], dtype=np.float32)
        n = len(r["X"])
        for s in range(len(r["models"])):
            for f0 in range(0, n, FRAME_CHUNK):
                tasks.append((name, s, f0, min(n, f0 + FRAME_CHUNK)))
        _REQ[name] = r
    procs = max(1, min(cores, len(tasks) + len(mfcc_tasks)))
    t0 = time.perf_counter()
    mfcc_results = []
    if procs > 1:
        with mp.get_context("fork").Pool(procs) as pool:
            results = pool.map(_task, tasks, chunksize=max(1, len(tasks) // (8 * procs))) if tasks else []
            if mfcc_tasks:
                mfcc_results = pool.map(_mfcc_task, mfcc_tasks, chunksize=1)
    else:
        results = [_task(t) for t in tasks]
        mfcc_results = [_mfcc_task(t) for t in mfcc_tasks]
    elapsed = time.perf_counter() - t0
    out = {}
    for name, r in mfcc_req.items():
        mine = [x for x in mfcc_results if x[0] == name]
        n = sum(x[3] for x in mine)
        tame = [x for x in mine if x[5] <= 90.0]
        nt = sum(x[3] for x in tame)
        out[name] = {"utterances": len(mine), "frames": int(len(r["device_feats"])), "dims": int(r["device_feats"].shape[1]) if r["device_feats"].ndim == 2 else 0,
                     "max_abs_diff_vs_oracle": max(x[1] for x in mine) if mine else None,
                     "mean_abs_diff_vs_oracle": (sum(x[2] for x in mine) / n) if n else None,
                     "fp32_fft_floor_max_abs_diff": max(x[4] for x in mine) if mine else None,
                     "worst_ratio_device_over_fp32_floor": max(x[1] / max(x[4], 1e-12) for x in mine) if mine else None,
                     "mel_band_dynamic_range_dB": {"median": float(np.median([x[5] for x in mine])) if mine else None, "max": max(x[5] for x in mine) if mine else None},
                     "utterances_within_90_dB": {"count": len(tame), "max_abs_diff_vs_oracle": max(x[1] for x in tame) if tame else None,
                                                 "mean_abs_diff_vs_oracle": (sum(x[2] for x in tame) / nt) if nt else None},
                     "gates": "SURVEY.md 8d: max <= 1e-3, mean <= 1e-5 after CMVN (the deltas, up to 4 x a term's error, are in the figures).  The synthetic "
                              "speakers of SURVEY 8d with high formants have mel bands up to 120 dB apart inside a frame; there the float32 spectrum itself "
                              "is the limit: fp32_fft_floor = the same chain in float64 with ONLY the FFT in float32 (scipy pocketfft), same input",
                     "oracle": "oracle/mfcc_oracle.py (float64 numpy restatement of MFCC.py:49-79, utils.py:24-31), Pool over utterances"}
    for name, r in req.items():
        U, S = len(r["offsets"]) - 1, len(r["models"])
        want = np.zeros((U, S))
        worst, clamp_bad = 0.0, 0
        for nm, s, w, cb, parts in results:
            if nm != name:
                continue
            worst = max(worst, w)
            clamp_bad += cb
            for u, v in parts:
                want[u, s] += v
        o = {"utterances": U, "models": S, "frames": int(len(r["X"])), "features": "float64 oracle from PCM" if r.get("kind") == "pcm_ll" else "the device's own",
             "mixture_evaluations": int(len(r["X"])) * int(sum(len(m[0]) for m in r["models"]))}
        if r.get("device_frame_ll") is not None:
            o["max_rel_frame_ll_diff_vs_oracle"] = worst
            o["frames_with_different_clamp_decision"] = clamp_bad
        if r.get("device_sums") is not None:
            dev = np.asarray(r["device_sums"], dtype=np.float64)
            o["max_rel_sum_diff_vs_oracle"] = float(np.max(np.abs(dev - want) / np.maximum(1.0, np.abs(want))))
            if S > 1:
                o["argmax_mismatches"] = int(np.sum(np.argmax(dev, axis=1) != np.argmax(want, axis=1)))
        out[name] = o
    out["_checker"] = {"processes": procs, "host_cores": cores, "host": hostinfo.describe(), "seconds": elapsed, "tasks": len(tasks),
                       "oracle": "oracle/gmm_oracle.c mode 3 (= the reference's C ABI arithmetic to remez5's 1.2e-6), float64"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
