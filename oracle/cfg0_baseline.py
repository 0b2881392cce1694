"""oracle/cfg0_baseline.py -- MEASUREMENT INFRASTRUCTURE ONLY (bench.py's cpu_baseline leg for BASELINE configs[0]).

`cfg0_baseline.py DIR`: the reference's enroll + predict on the WAV files bench.py wrote into DIR (enroll_<s>.wav, test_<s>.wav),
restating src/speaker-recognition.py:52-90 / src/gui/interface.py:55-109 on this host's cores: features = the float64 numpy
restatement of MFCC.py (oracle/mfcc_oracle.py; the reference's LPC half needs the absent scikits.talkbox and is left out, as on
the device side), GMMs = the reference's own compiled C++ (oracle/_ref/pygmm_ref.so: train_model with the reference's defaults --
32... here 16 mixtures, 200 iterations, threshold 0.01, k-means start -- and score_all / len as GMMSetPyGMM.predict_one,
gmmset.py:95-99), concurrency = the cores this container may use.  Prints one JSON line."""
import ctypes as C
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import gmm_oracle as go, hostinfo, mfcc_oracle as mo  # noqa: E402
from speaker_recognition_amd._lib import Parameter  # noqa: E402
from speaker_recognition_amd.cli import read_wav  # noqa: E402

KW, K = dict(win_length_ms=25, win_shift_ms=10), 16


def main():
    d = sys.argv[1]
    if not os.path.exists(go.REF_SO):
        print(json.dumps({"error": "oracle/_ref/pygmm_ref.so absent (built from /root/reference by `make -C oracle ref`)"}))
        return
    enroll = sorted(glob.glob(os.path.join(d, "enroll_*.wav")))
    test = sorted(glob.glob(os.path.join(d, "test_*.wav")))
    label = lambda f: os.path.basename(f).split("_")[1].split(".")[0]
    cores = hostinfo.effective_cores()
    ref = go.RefLib()
    os.chdir(d)                                        # the reference drops gmm-training-intermediate-dump.model into the cwd
    t0 = time.perf_counter()
    feats = {label(f): mo.extract(*read_wav(f), **KW) for f in enroll}
    t1 = time.perf_counter()
    handles = {}
    devnull = os.open(os.devnull, os.O_WRONLY)
    saved = os.dup(1)
    os.dup2(devnull, 1)                                # the reference prints its parameter block
    try:
        for l, X in feats.items():
            h = ref.lib.new_gmm(K, 1)
            p = Parameter(nr_instance=len(X), nr_dim=X.shape[1], nr_mixture=K, min_covar=1e-3, threshold=0.01,
                          nr_iteration=200, init_with_kmeans=0, concurrency=cores, verbosity=0)
            rows, keep = ref.rows(X)
            ref.lib.train_model(h, rows, C.byref(p))
            handles[l] = h
    finally:
        os.dup2(saved, 1)
    t2 = time.perf_counter()
    ok, frames = 0, 0
    for f in test:
        X = mo.extract(*read_wav(f), **KW)
        frames += len(X)
        scores = {k: ref.score_all(h, X, cores) / len(X) for k, h in handles.items()}
        ok += int(max(scores, key=scores.get) == label(f))
    t3 = time.perf_counter()
    print(json.dumps({"enroll_features_s": t1 - t0, "train_s": t2 - t1, "predict_s": t3 - t2, "total_s": t3 - t0, "correct": ok, "of": len(test),
                      "predict_frames": frames, "cores": cores, "host": hostinfo.describe(),
                      "kind": "reference C++ (train_model / score_all of oracle/_ref/pygmm_ref.so) + float64 numpy restatement of MFCC.py"}))


if __name__ == "__main__":
    main()
