"""oracle/lpc_oracle.py -- TEST INFRASTRUCTURE ONLY.  **Parity unpinned.**

Float64 restatement of the reference's LPC features (/root/reference/src/feature/LPC.py:14-75).
The arithmetic of ``levinson_lpc.lpc`` lives in the third-party package ``scikits.talkbox``
(imported at LPC.py:9; not vendored, not installed here, version unpinned in the reference's
README.md:20 / Dockerfile -- the last release is 0.2.5), so it is restated from that package's
published algorithm:

  * ``acorr_lpc``: biased autocorrelation through the FFT,
        nfft = 2**nextpow2(2n - 1);  r = real(ifft(|fft(x, nfft)|^2))[:n+1] / n
  * ``levinson_1d``: the Levinson-Durbin recursion on r[0..order] -> (a, e, k), a[0] = 1.

and anchored on the reference's own call site: ``lpcc(frame) = lpc(frame, n_lpc)[0][1:]``
(LPC.py:40-42), frames windowed with MFCC.hamming and pre-emphasised exactly as in MFCC.py
(LPC.py:46-55), NaN -> 0 (LPC.py:56), defaults 32 ms / 16 ms / order 15 / 0.95 (LPC.py:59-63).
No golden vectors can be generated (talkbox is absent); tests cross-check the recursion against
``scipy.linalg.solve_toeplitz`` on the same autocorrelation.
"""
from __future__ import annotations

import numpy as np

from .mfcc_oracle import hamming


def acorr_lpc(x: np.ndarray) -> np.ndarray:
    n = x.shape[-1]
    nfft = 1 << int(np.ceil(np.log2(2 * n - 1)))
    a = np.real(np.fft.ifft(np.abs(np.fft.fft(x, n=nfft)) ** 2))
    return a[..., :n + 1] / n


def levinson_1d(r: np.ndarray, order: int):
    r = np.atleast_1d(np.asarray(r, dtype=np.float64))
    a = np.empty(order + 1)
    t = np.empty(order + 1)
    k = np.empty(order)
    a[0] = 1.0
    e = r[0]
    with np.errstate(all="ignore"):
        for i in range(1, order + 1):
            acc = r[i]
            for j in range(1, i):
                acc += a[j] * r[i - j]
            k[i - 1] = -acc / e
            a[i] = k[i - 1]
            t[:order] = a[:order]
            for j in range(1, i):
                a[j] += k[i - 1] * t[i - j]
            e *= 1 - k[i - 1] * k[i - 1]
    return a, e, k


class LPCExtractor:
    """LPC.py:14-57."""

    def __init__(self, fs, win_length_ms=32, win_shift_ms=16, n_lpc=15, pre_emphasis_coef=0.95):
        self.PRE_EMPH = pre_emphasis_coef
        self.n_lpc = n_lpc
        self.FRAME_LEN = int(float(win_length_ms) / 1000 * fs)
        self.FRAME_SHIFT = int(float(win_shift_ms) / 1000 * fs)
        self.window = hamming(self.FRAME_LEN)

    def extract(self, signal: np.ndarray) -> np.ndarray:
        signal = np.asarray(signal, dtype=np.float64)
        frames = (len(signal) - self.FRAME_LEN) // self.FRAME_SHIFT + 1      # LPC.py:47
        feature = []
        for f in range(frames):
            frame = signal[f * self.FRAME_SHIFT: f * self.FRAME_SHIFT + self.FRAME_LEN] * self.window
            frame[1:] -= frame[:-1] * self.PRE_EMPH
            a, _, _ = levinson_1d(acorr_lpc(frame), self.n_lpc)
            feature.append(a[1:])
        feature = np.array(feature)
        feature[np.isnan(feature)] = 0                                        # LPC.py:56
        return feature


def extract(fs, signal=None, **kwargs) -> np.ndarray:
    if signal is None:
        fs, signal = fs[0], fs[1]
    return LPCExtractor(fs, **kwargs).extract(np.asarray(signal).astype(float))
