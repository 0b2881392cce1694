"""LTSD voice-activity detector -- same surface as the reference's ``src/filters/ltsd.py``
(``init_params_by_noise``, ``filter``; ltsd.py:32-64), LTSD values from the GPU (csrc/ltsd.hip).

What the reference fixes and this keeps: window = ``int(0.04644 * fs)`` Hann (ltsd.py:17,66-69),
order 5 (:21), hop = window/2 (:56-57), first channel of multi-channel input (:79-82),
``lambda0 = 1.1 * max LTSD(noise vs itself)``, ``lambda1 = 2 * lambda0`` (:39-41), intervals in
samples = ``(start * window/2, (finish + 1) * window/2)`` over window indices (:56-57), voiced
signal = concatenation of the intervals (:59-64).

What the reference leaves to its (unavailable) fork of pyssp and this defines: the decision rule
that turns LTSD values and the two thresholds into window intervals -- here a double-threshold
(Schmitt) rule: a voiced interval is a maximal run of windows with LTSD > lambda0 that contains at
least one window with LTSD > lambda1.  Parity unpinned."""
import numpy as np

from .. import _lib
from .._lib import check, lib
from ..core import Batch

MAGIC_NUMBER = 0.04644


def ltsd_values(signals, noise_amp, window_size, order=5, batch=None):
    """LTSD (dB) of every window of every signal -> list of float32 arrays (one per signal).
    ``batch``: an int16 PCM ``Batch`` to reuse (its device buffers are rewritten in place)."""
    if batch is not None and all(np.asarray(s).dtype == np.int16 for s in signals):
        batch.reset_pcm(signals)
        b = batch
    else:
        b = Batch.from_pcm([np.asarray(s) for s in signals])
    n_win = [max(0, len(s) // (window_size // 2) - 1) for s in signals]
    out = np.zeros(max(1, sum(n_win)), dtype=np.float32)
    off = np.zeros(len(signals) + 1, dtype=np.int64)
    na = np.ascontiguousarray(noise_amp, dtype=np.float32)
    check(lib().sr_ltsd_compute(b._h, int(window_size), int(order), _lib.as_fp(na), _lib.as_fp(out),
                                _lib.as_i64p(off)), "sr_ltsd_compute")
    return [out[off[u]:off[u + 1]].copy() for u in range(len(signals))]


def noise_spectrum(noise_signal, window_size):
    """Mean amplitude spectrum (bins 0..window/2) of the noise recording."""
    b = Batch.from_pcm([np.asarray(noise_signal)])
    out = np.zeros(window_size // 2 + 1, dtype=np.float32)
    check(lib().sr_ltsd_noise_spectrum(b._h, int(window_size), _lib.as_fp(out)), "sr_ltsd_noise_spectrum")
    return out


def voiced_runs(ltsds, lambda0, lambda1):
    """[(start, finish)] window indices, finish inclusive (the reference's ``res``)."""
    above = np.asarray(ltsds) > lambda0
    res, i, n = [], 0, len(above)
    while i < n:
        if not above[i]:
            i += 1
            continue
        j = i
        while j + 1 < n and above[j + 1]:
            j += 1
        if np.max(ltsds[i:j + 1]) > lambda1:
            res.append((i, j))
        i = j + 1
    return res


class LTSD_VAD(object):
    order = 5

    def __getstate__(self):           # device handles do not pickle (ModelInterface.dump)
        d = dict(self.__dict__)
        d["_batch"] = None
        return d

    def __init__(self):
        self.fs = 0
        self.window_size = 0
        self.lambda0 = 0.0
        self.lambda1 = 0.0
        self.noise_signal = None
        self.noise_amp = None
        self._batch = None            # reused across filter() calls: a serving loop allocates nothing

    def _mononize_signal(self, signal):
        signal = np.asarray(signal)
        if signal.ndim > 1:
            signal = signal[:, 0]           # ltsd.py:79-82: the first channel, not the mean
        return signal

    def _init_window(self, fs):
        self.fs = fs
        self.window_size = int(MAGIC_NUMBER * fs)

    def init_params_by_noise(self, fs, noise_signal):
        noise_signal = self._mononize_signal(noise_signal)
        self.noise_signal = np.array(noise_signal)
        self._init_window(fs)
        self.noise_amp = noise_spectrum(self.noise_signal, self.window_size)
        ltsds = ltsd_values([self.noise_signal], self.noise_amp, self.window_size, self.order)[0]
        max_ltsd = float(np.max(ltsds)) if len(ltsds) else 0.0
        self.lambda0 = max_ltsd * 1.1
        self.lambda1 = self.lambda0 * 2.0

    def ltsd(self, signal):
        signal = self._mononize_signal(signal)
        if signal.dtype == np.int16:
            if self._batch is None:
                self._batch = Batch.from_pcm([signal])
            return ltsd_values([signal], self.noise_amp, self.window_size, self.order, batch=self._batch)[0]
        return ltsd_values([signal], self.noise_amp, self.window_size, self.order)[0]

    def filter(self, signal):
        if self.noise_amp is None:
            raise RuntimeError("LTSD_VAD: init_params_by_noise first")
        signal = self._mononize_signal(signal)
        ltsds = self.ltsd(signal)
        half = self.window_size // 2
        res = [(s * half, (f + 1) * half) for s, f in voiced_runs(ltsds, self.lambda0, self.lambda1)]
        if not res:
            return np.array([]), []
        return np.concatenate([signal[s:f] for s, f in res]), res

    def filter_many(self, signals):
        """One launch for many signals of the same sampling rate -> [(voiced, intervals)]."""
        sigs = [self._mononize_signal(s) for s in signals]
        half = self.window_size // 2
        out = []
        for sig, l in zip(sigs, ltsd_values(sigs, self.noise_amp, self.window_size, self.order)):
            res = [(s * half, (f + 1) * half) for s, f in voiced_runs(l, self.lambda0, self.lambda1)]
            out.append((np.concatenate([sig[s:f] for s, f in res]) if res else np.array([]), res))
        return out
