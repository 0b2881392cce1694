#!/usr/bin/env python3
"""Timeline of ONE sr_multi_predict_pcm call from page-locked host PCM (round 6: where the exposed H2D time is).

  run     CFG   -> the workload (CFG = 1: configs[1], 2: configs[2]); for `rocprofv3 --kernel-trace --memory-copy-trace`
  report  DIR   -> the last call of the trace: copies and kernels merged in time order, gaps, busy time per engine
"""
import csv
import glob
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def run(cfg):
    import numpy as np
    import bench
    from speaker_recognition_amd import _lib, synth
    from speaker_recognition_amd.core import MultiPredictor
    from speaker_recognition_amd.pygmm import GMM
    n_samples = (bench.FRAMES_PER_UTT + bench.ND - 1) * 160 + 400
    if cfg == 1:
        base = bench.base_clips(bench.CFG1_MODELS, n_samples)
        gm = [GMM.from_arrays(*synth.synth_gmm(bench.CFG1_MIX, bench.DIM, bench.MODEL_SEED + s)) for s in range(bench.CFG1_MODELS)]
        cat, off = bench.make_pcm(base, bench.CFG1_UTTS, 0)
    else:
        base = bench.base_clips(bench.CFG2_SPEAKERS, n_samples)
        ubm = synth.synth_gmm(bench.CFG2_MIX, bench.DIM, 99)
        gm = [GMM.from_arrays(*m) for m in [ubm] + [synth.synth_map_speaker(ubm, 500 + s) for s in range(bench.CFG2_SPEAKERS)]]
        cat, off = bench.make_pcm(base, int(os.environ.get("UTTS", bench.CFG2_UTTS)), 0)
    from speaker_recognition_amd.core import Batch, MfccExtractor, ModelSet
    ex, ms, pcm = MfccExtractor(bench.FS, **bench.MFCC_KW), ModelSet(gm), Batch.from_pcm((cat, off))
    out = (np.zeros((len(off) - 1, len(gm))), np.full(len(off) - 1, -1, np.int32))
    _lib.host_register(out[0])
    tr = []
    for i in range(5):
        t0 = time.perf_counter()
        ex.predict_batch(ms, pcm, nd=bench.ND, out=out)
        tr.append((time.perf_counter() - t0) * 1e3)
    print("resident ms per step:", ["%.2f" % t for t in tr])
    del pcm
    _lib.host_register(cat)
    mp_ = MultiPredictor(gm, bench.FS, n_slots=1, **bench.MFCC_KW)
    ts = []
    for i in range(int(os.environ.get("CALLS", 9))):
        t0 = time.perf_counter()
        mp_.predict_concat(cat, off, nd=bench.ND)
        ts.append((time.perf_counter() - t0) * 1e3)
    print("ms per call:", ["%.2f" % t for t in ts], "pcie floor %.2f ms" % (cat.nbytes / 55e9 * 1e3))


def report(d):
    ev = []
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K", r["Kernel_Name"].replace("void sr::", "").replace("(anonymous namespace)::", "")[:56]))
    for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C", r.get("Direction", "copy")))
    ev.sort()
    # the last call: from the last H2D copy group backwards -- take the events after the last gap > 2 ms before the final event burst
    ends = [i for i in range(1, len(ev)) if ev[i][0] - max(e[1] for e in ev[max(0, i - 50):i]) > 1_000_000]
    start = ends[-1] if ends else 0
    call = ev[start:]
    t0 = call[0][0]
    busy = {"K": 0, "C": 0}
    last_end = {"K": None, "C": None}
    big = [e for e in call if e[1] - e[0] > int(os.environ.get("MIN_NS", 20000))]
    for s, e, k, name in call:
        busy[k] += e - s
    print("call: %d events, wall %.3f ms, kernel busy %.3f ms, copy busy %.3f ms" % (len(call), (max(e[1] for e in call) - t0) / 1e6, busy["K"] / 1e6, busy["C"] / 1e6))
    for s, e, k, name in big:
        gap = (s - last_end[k]) / 1e3 if last_end[k] else 0.0
        print("%9.3f ms  %s  dur %9.1f us  gap(same engine) %8.1f us  %s" % ((s - t0) / 1e6, k, (e - s) / 1e3, gap, name))
        last_end[k] = e


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]))
    else:
        report(sys.argv[2])
