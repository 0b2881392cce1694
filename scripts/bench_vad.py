#!/usr/bin/env python3
"""LTSD front end throughput: U signals of T seconds at 16 kHz (window 743, hop 371, order 5) in one
sr_ltsd_compute call (H2D of the PCM, both kernels, D2H of the LTSD values), and the numpy
restatement on one signal for scale."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ltsd_oracle as lo  # noqa: E402
from speaker_recognition_amd import synth  # noqa: E402
from speaker_recognition_amd.filters import ltsd as L  # noqa: E402


def main():
    fs, U, T = 16000, int(os.environ.get("VAD_U", 500)), 10.0
    N = lo.window_size(fs)
    rng = np.random.default_rng(0)
    noise = rng.normal(0, 120, 3 * fs).astype(np.int16)
    base = [(synth.synth_speech(s, T, fs, seed=s) // 4 + rng.normal(0, 120, int(T * fs)).astype(np.int16)).astype(np.int16) for s in range(10)]
    sigs = [base[u % 10] for u in range(U)]
    na = L.noise_spectrum(noise, N)
    L.ltsd_values(sigs[:2], na, N)
    ts = []
    for r in range(3):
        t0 = time.perf_counter()
        out = L.ltsd_values(sigs, na, N)
        ts.append(time.perf_counter() - t0)
    t0 = time.perf_counter()
    want = lo.ltsd(sigs[0], lo.noise_spectrum(noise, N), N)
    t_cpu = time.perf_counter() - t0
    n_win = sum(len(o) for o in out)
    print(json.dumps({"signals": U, "seconds_each": T, "fs": fs, "window": N, "windows": n_win,
                      "gpu_call_s": min(ts), "audio_seconds_per_second": U * T / min(ts),
                      "numpy_one_signal_s": t_cpu, "numpy_audio_seconds_per_second": T / t_cpu,
                      "max_abs_db_diff_signal0": float(np.max(np.abs(out[0] - want)))}))


if __name__ == "__main__":
    main()
