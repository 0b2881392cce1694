"""oracle/mfcc_oracle.py -- TEST INFRASTRUCTURE ONLY.

Float64 numpy restatement of the reference's own MFCC chain
(/root/reference/src/feature/MFCC.py and src/feature/utils.py; Python 2 there, so it
cannot be imported here -- tests/golden/make_golden.py executes a mechanically patched copy
of the reference text in memory to pin this restatement, and the resulting known-answer
vectors live in tests/golden/mfcc_golden.npz).

Only tests/, ``__graft_entry__.smoke()`` and bench.py's ``cpu_baseline`` leg may import this.
"""
from __future__ import annotations

import numpy as np

POWER_SPECTRUM_FLOOR = 1e-100  # MFCC.py:8


def hamming(n: int) -> np.ndarray:
    """MFCC.py:14-16 -- half-sample-shifted Hamming window."""
    return 0.54 - 0.46 * np.cos(2 * np.pi / n * (np.arange(n) + 0.5))


def dctmtx(n: int) -> np.ndarray:
    """MFCC.py:107-113 -- orthonormal DCT-II matrix of order n."""
    x, y = np.meshgrid(range(n), range(n))
    D = np.sqrt(2.0 / n) * np.cos(np.pi * (2 * x + 1) * y / (2 * n))
    D[0] /= np.sqrt(2)
    return D


def mel_filterbank(fs: float, fft_size: int, n_bands: int):
    """MFCC.py:81-105 -- melfb.m-style bank, returns (M[n_bands, fft/2+1], CF)."""
    f0 = 700.0 / fs
    fn2 = int(np.floor(fft_size / 2))
    lr = np.log(1 + 0.5 / f0) / (n_bands + 1)
    CF = fs * f0 * (np.exp(np.arange(1, n_bands + 1) * lr) - 1)
    bl = fft_size * f0 * (np.exp(np.array([0, 1, n_bands, n_bands + 1]) * lr) - 1)
    b1 = int(np.floor(bl[0])) + 1
    b2 = int(np.ceil(bl[1]))
    b3 = int(np.floor(bl[2]))
    b4 = min(fn2, int(np.ceil(bl[3]))) - 1
    pf = np.log(1 + np.arange(b1, b4 + 1) / f0 / fft_size) / lr
    fp = np.floor(pf)
    pm = pf - fp
    M = np.zeros((n_bands, 1 + fn2))
    for c in range(b2 - 1, b4):
        r = int(fp[c] - 1)
        M[r, c + 1] += 2 * (1 - pm[c])
    for c in range(b3):
        r = int(fp[c])
        M[r, c + 1] += 2 * pm[c]
    return M, CF


class MFCCExtractor:
    """MFCC.py:18-79."""

    def __init__(self, fs, win_length_ms=32, win_shift_ms=16, FFT_SIZE=2048, n_bands=50,
                 n_coefs=13, PRE_EMPH=0.95):
        self.PRE_EMPH = PRE_EMPH
        self.fs = fs
        self.n_bands = n_bands
        self.coefs = n_coefs
        self.FFT_SIZE = FFT_SIZE
        self.FRAME_LEN = int(float(win_length_ms) / 1000 * fs)      # MFCC.py:28
        self.FRAME_SHIFT = int(float(win_shift_ms) / 1000 * fs)     # MFCC.py:29
        self.window = hamming(self.FRAME_LEN)
        self.M, self.CF = mel_filterbank(fs, FFT_SIZE, n_bands)
        self.D = dctmtx(n_bands)[1:n_coefs + 1]                      # MFCC.py:36-37, c0 dropped

    def n_frames(self, n_samples: int) -> int:
        return (n_samples - self.FRAME_LEN) // self.FRAME_SHIFT + 1  # MFCC.py:57 (py2 int division)

    def raw_cepstra(self, signal: np.ndarray) -> np.ndarray:
        """MFCC.py:53-71 -- everything before CMVN, float64[T, n_coefs]."""
        signal = np.asarray(signal, dtype=np.float64)
        if signal.ndim > 1:
            signal = np.mean(signal, axis=1)                         # MFCC.py:53-55
        assert len(signal) > 5 * self.FRAME_LEN, "Signal too short!"  # MFCC.py:56
        frames = self.n_frames(len(signal))
        idx = np.arange(frames)[:, None] * self.FRAME_SHIFT + np.arange(self.FRAME_LEN)[None, :]
        fr = signal[idx] * self.window[None, :]                      # MFCC.py:61-62
        fr[:, 1:] = fr[:, 1:] - fr[:, :-1] * self.PRE_EMPH           # MFCC.py:64 (RHS is a temporary)
        X = np.abs(np.fft.fft(fr, self.FFT_SIZE, axis=1)[:, :self.FFT_SIZE // 2 + 1]) ** 2  # :66
        X[X < POWER_SPECTRUM_FLOOR] = POWER_SPECTRUM_FLOOR           # MFCC.py:67
        return np.dot(np.log(np.dot(X, self.M.T)), self.D.T)         # MFCC.py:69

    def extract(self, signal: np.ndarray) -> np.ndarray:
        feature = self.raw_cepstra(signal)
        if feature.shape[0] > 1:                                      # MFCC.py:74-77
            mu = np.mean(feature, axis=0)
            sigma = np.std(feature, axis=0)
            feature = (feature - mu) / sigma
        return feature


def diff_feature(feat: np.ndarray, nd: int = 1) -> np.ndarray:
    """utils.py:24-31."""
    diff = feat[1:] - feat[:-1]
    feat = feat[1:]
    if nd == 1:
        return np.concatenate((feat, diff), axis=1)
    elif nd == 2:
        d2 = diff[1:] - diff[:-1]
        return np.concatenate((feat[1:], diff[1:], d2), axis=1)
    raise ValueError("nd must be 1 or 2")


_cache: dict = {}


def get_mfcc_extractor(fs, win_length_ms=32, win_shift_ms=16, FFT_SIZE=2048, n_filters=50,
                       n_ceps=13, pre_emphasis_coef=0.95) -> MFCCExtractor:
    """MFCC.py:115-121 (memoised as utils.cached_func does, utils.py:11-21)."""
    key = (fs, win_length_ms, win_shift_ms, FFT_SIZE, n_filters, n_ceps, pre_emphasis_coef)
    if key not in _cache:
        _cache[key] = MFCCExtractor(fs, win_length_ms, win_shift_ms, FFT_SIZE, n_filters,
                                    n_ceps, pre_emphasis_coef)
    return _cache[key]


def extract(fs, signal=None, diff=False, nd=1, **kwargs) -> np.ndarray:
    """MFCC.py:123-132; ``nd`` (delta order handed to diff_feature) is the one addition."""
    if signal is None:
        assert type(fs) == tuple
        fs, signal = fs[0], fs[1]
    signal = np.asarray(signal).astype(float)                         # MFCC.py:128
    ret = get_mfcc_extractor(fs, **kwargs).extract(signal)
    if diff:
        return diff_feature(ret, nd)
    return ret
