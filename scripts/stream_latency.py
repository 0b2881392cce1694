#!/usr/bin/env python3
"""BASELINE configs[4] (streaming): 8 kHz audio in 1 s windows -> MFCC (reference defaults: 32/16 ms,
FFT 2048, 13 ceps) -> CMVN -> 256-mixture speaker GMMs -> decision.  Measures the host-observed
decision latency per window (H2D of the window + 4 kernels + D2H of the result, synchronous) and the
throughput when many independent streams are batched.  VAD (third-party LTSD in the reference) is not
part of this path."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from speaker_recognition_amd import _lib, synth  # noqa: E402
from speaker_recognition_amd.core import Batch, MfccExtractor, ModelSet, ServingStream  # noqa: E402
from speaker_recognition_amd.pygmm import GMM  # noqa: E402


def main():
    fs, n_speakers, K = 8000, int(os.environ.get("STREAM_S", 20)), 256
    ex = MfccExtractor(fs)
    models = ModelSet([GMM.from_arrays(*synth.synth_gmm(K, 13, 7 + s)) for s in range(n_speakers)])
    audio = synth.synth_speech(3, 40.0, fs)
    out = {"fs": fs, "window_s": 1.0, "speakers": n_speakers, "mixtures": K, "dim": 13,
           "frames_per_window": ex.num_frames(fs)}
    # single stream, one window at a time
    win = Batch.from_pcm([audio[:fs]])
    lat = []
    for i in range(230):
        chunk = audio[(i % 39) * fs // 2:(i % 39) * fs // 2 + fs]
        t0 = time.perf_counter()
        win.update_pcm(chunk)
        sums, arg = ex.predict_batch(models, win, nd=0)
        lat.append((time.perf_counter() - t0) * 1e3)
    lat = np.array(lat[30:])
    out["single_stream_latency_ms"] = {"p50": float(np.percentile(lat, 50)), "p90": float(np.percentile(lat, 90)),
                                       "p99": float(np.percentile(lat, 99)), "min": float(lat.min())}
    # the whole configs[4] chain: LTSD VAD in front (filters.VAD surface: noise calibration once, then per
    # window LTSD -> voiced intervals -> the reference's one-third rule, gui/interface.py:43-53) and
    # MFCC + GMM on the voiced samples only; all device buffers reused, host-observed latency per window
    from speaker_recognition_amd.filters import VAD
    rng = np.random.default_rng(5)
    floor = rng.normal(0, 60, len(audio)).astype(np.int16)
    gate = (np.arange(len(audio)) // (fs * 3 // 2)) % 2 == 0            # 1.5 s speech, 1.5 s background
    scene = (np.where(gate, audio // 2, 0) + floor).astype(np.int16)
    vad = VAD()
    vad.init_noise(fs, rng.normal(0, 60, 3 * fs).astype(np.int16))
    vwin = Batch.from_pcm([scene[:fs]])
    lat_v, n_voiced, n_dec = [], 0, 0
    for i in range(260):
        chunk = scene[(i % 70) * fs // 2:(i % 70) * fs // 2 + fs]
        t0 = time.perf_counter()
        voiced, intervals = vad.filter(fs, chunk)
        if len(voiced) > len(chunk) / 3 and ex.num_frames(len(voiced)) > 0:
            vwin.reset_pcm([voiced])
            sums, arg = ex.predict_batch(models, vwin, nd=0)
            n_dec += 1
        lat_v.append((time.perf_counter() - t0) * 1e3)
        n_voiced += len(voiced)
    lat_v = np.array(lat_v[30:])
    out["single_stream_with_ltsd_vad_latency_ms"] = {
        "p50": float(np.percentile(lat_v, 50)), "p90": float(np.percentile(lat_v, 90)),
        "p99": float(np.percentile(lat_v, 99)), "windows_decided": n_dec, "windows": 260,
        "voiced_fraction": n_voiced / (260.0 * fs)}
    # many concurrent streams batched per tick
    for n_streams in (64, 1024):
        batch = Batch.from_pcm([audio[(j % 39) * fs // 2:(j % 39) * fs // 2 + fs] for j in range(n_streams)])
        cat = np.concatenate([audio[(j % 39) * fs // 2:(j % 39) * fs // 2 + fs] for j in range(n_streams)])
        ts = []
        for i in range(30):
            t0 = time.perf_counter()
            batch.update_pcm(cat)
            sums, arg = ex.predict_batch(models, batch, nd=0)
            ts.append((time.perf_counter() - t0) * 1e3)
        ts = np.array(ts[5:])
        out["batched_%d_streams" % n_streams] = {"tick_ms_p50": float(np.percentile(ts, 50)),
                                                 "windows_per_s": n_streams / (np.percentile(ts, 50) * 1e-3),
                                                 "realtime_factor": n_streams * 1.0 / (np.percentile(ts, 50) * 1e-3)}
    # double-buffered session: H2D of tick i+1 (own HIP stream, pinned memory) under the kernels of tick i
    for n_streams, graph in ((1, False), (1, True), (64, False), (64, True), (1024, False), (1024, True)):
        st = ServingStream(ex, models, n_streams, fs, graph=graph)
        cat = np.stack([audio[(j % 39) * fs // 2:(j % 39) * fs // 2 + fs] for j in range(n_streams)])
        n_ticks = 60
        st.submit(cat)
        dev = []
        t0 = time.perf_counter()
        for i in range(n_ticks):
            st.submit(cat)
            dev.append(st.collect()[2])
        st.collect()
        dt = (time.perf_counter() - t0) / n_ticks
        # one tick at a time: submit -> collect, host-observed
        one = []
        for i in range(200):
            t1 = time.perf_counter()
            st.submit(cat)
            st.collect()
            one.append((time.perf_counter() - t1) * 1e3)
        one = np.array(one[20:])
        out["double_buffered_%d_streams%s" % (n_streams, "_hipgraph" if graph else "")] = {
            "sync_tick_latency_ms_p50": float(np.percentile(one, 50)), "sync_tick_latency_ms_p99": float(np.percentile(one, 99)),
            "tick_ms": dt * 1e3, "windows_per_s": n_streams / dt,
                                                         "device_ms_per_tick_p50": float(np.percentile(dev, 50))}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
