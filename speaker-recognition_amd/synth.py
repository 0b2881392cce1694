"""Deterministic synthetic workloads: speaker-like audio, diagonal
GMMs in the reference's text format, and frames drawn from those models.

Used by tests/, bench.py and the golden-vector generator; nothing here touches the GPU.
"""
from __future__ import annotations

import numpy as np


def synth_speech(speaker: int, seconds: float, fs: int = 16000, seed: int | None = None,
                 amplitude: float = 12000.0) -> np.ndarray:
    """int16 mono 'speech' for one synthetic speaker.

    Impulse train at f0 = 90 + 12*s Hz (1 % jitter) + 0.3 white noise, shaped by three
    two-pole resonators at (500+40s, 1500+60s, 2500+45s) Hz with 80 Hz bandwidth (formants are
    folded below 0.45 fs), amplitude-modulated by a 4 Hz syllable envelope.
    """
    from scipy.signal import lfilter

    rng = np.random.default_rng(1000 + speaker if seed is None else seed)
    n = int(round(seconds * fs))
    f0 = 90.0 + 12.0 * (speaker % 40)
    period = fs / f0
    src = np.zeros(n)
    t = 0.0
    while t < n:
        src[int(t)] = 1.0
        t += period * (1.0 + 0.01 * rng.standard_normal())
    src += 0.3 * rng.standard_normal(n) / np.sqrt(period)
    y = src
    for base, step in ((500.0, 40.0), (1500.0, 60.0), (2500.0, 45.0)):
        fc = base + step * speaker
        fc = fc % (0.45 * fs)
        r = np.exp(-np.pi * 80.0 / fs)
        a = [1.0, -2.0 * r * np.cos(2 * np.pi * fc / fs), r * r]
        y = lfilter([1.0 - r], a, y)
    env = 0.55 + 0.45 * np.sin(2 * np.pi * 4.0 * np.arange(n) / fs + 0.37 * speaker)
    y = y * env
    y = y / (np.max(np.abs(y)) + 1e-12) * amplitude
    return np.round(y).astype(np.int16)


def synth_gmm(K: int, D: int, seed: int):
    """(weights[K], mean[K,D], sigma[K,D]) float64, already rounded through the reference's
    6-significant-digit text format (gmm.cc:101-123, :655-662) so every consumer sees the
    same parameters.  sigma ~ U(0.2, 1.5) floored at sqrt(1e-3) (gmm.hh:58)."""
    rng = np.random.default_rng(seed)
    mean = rng.standard_normal((K, D))
    sigma = np.maximum(rng.uniform(0.2, 1.5, (K, D)), np.sqrt(1e-3))
    w = rng.dirichlet(np.ones(K))
    r6 = np.vectorize(lambda v: float("%g" % v))
    return r6(w), r6(mean), r6(sigma)


def synth_map_speaker(ubm, seed: int, n_frames_per_mix: float = 40.0, relevance: float = 16.0):
    """A MAP-adapted-looking speaker model: UBM with means shifted by alpha_k * N(0, 0.3^2),
    alpha_k = n_k/(n_k+16) (mimics gmmubm.cc:57-69); weights and sigmas shared (gmmubm.cc:40-51,76-81)."""
    w, mean, sigma = ubm
    rng = np.random.default_rng(seed)
    nk = w * n_frames_per_mix * len(w)
    alpha = nk / (nk + relevance)
    shifted = mean + alpha[:, None] * 0.3 * rng.standard_normal(mean.shape)
    r6 = np.vectorize(lambda v: float("%g" % v))
    return w, r6(shifted), sigma


def draw_frames(model, n: int, seed: int, outlier_frac: float = 0.0) -> np.ndarray:
    """float32[n, D] frames drawn from the model (mixture ~ w, x = mu + sigma*N(0,1)); a fraction
    is pushed +60 on every dimension to exercise the reference's underflow clamp (gmm.cc:34-38)."""
    w, mean, sigma = model
    rng = np.random.default_rng(seed)
    k = rng.choice(len(w), size=n, p=w / w.sum())
    x = mean[k] + sigma[k] * rng.standard_normal((n, mean.shape[1]))
    if outlier_frac > 0:
        m = rng.random(n) < outlier_frac
        x[m] += 60.0
    return x.astype(np.float32)
