"""Handle wrappers over the sr_* part of the C ABI: device batches, speaker sets, the MFCC
extractor object.  Thin by design -- marshalling only, all arithmetic is in lib/pygmm.so."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import SRError, check, lib


class Batch:
    """An utterance batch resident in HBM (PCM or features); frees its device memory on GC."""

    def __init__(self, handle, owner=True):
        if not handle:
            raise SRError("batch creation failed: %s" % _lib.last_error())
        self._h = C.c_void_p(handle)
        self._owner = owner

    @classmethod
    def from_pcm(cls, signals) -> "Batch":
        """``signals``: list of 1-D arrays (int16 -> stays int16 on the device; anything else is
        sent as float32), or a tuple ``(concatenated, offsets)``."""
        if isinstance(signals, tuple):
            cat, offsets = signals
            cat = np.ascontiguousarray(cat)
            offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        else:
            sigs = [np.asarray(s) for s in signals]
            for s in sigs:
                if s.ndim != 1:
                    raise ValueError("Only Support Mono Wav File!")   # gui/utils.py:12
            offsets = np.zeros(len(sigs) + 1, dtype=np.int64)
            offsets[1:] = np.cumsum([len(s) for s in sigs])
            all_i16 = all(s.dtype == np.int16 for s in sigs)
            cat = (np.concatenate(sigs) if sigs else np.zeros(0, np.int16))
            cat = np.ascontiguousarray(cat if all_i16 else cat.astype(np.float32))
        if cat.dtype == np.int16:
            h = lib().sr_batch_from_pcm(cat.ctypes.data_as(C.POINTER(C.c_int16)),
                                        _lib.as_i64p(offsets), len(offsets) - 1)
        else:
            cat = np.ascontiguousarray(cat, dtype=np.float32)
            h = lib().sr_batch_from_pcm_f32(_lib.as_fp(cat), _lib.as_i64p(offsets), len(offsets) - 1)
        return cls(h)

    @classmethod
    def from_features(cls, X, offsets=None) -> "Batch":
        """``X``: [n, dim] matrix with ``offsets`` [U+1], or a list of [T_u, dim] matrices."""
        if offsets is None and isinstance(X, (list, tuple)):
            if X and all(isinstance(x, np.ndarray) and x.ndim == 2 for x in X):
                # what feature extraction hands over -- float64 matrices (MFCC.py:69-79) -- converted WHILE they are gathered:
                # one pass over the frames instead of a float32 copy per utterance and a second pass to join them
                mats = X
                X = np.concatenate(mats, axis=0, dtype=np.float32, casting="unsafe")
            else:
                mats = [_lib.f32_matrix(x) for x in X]
                X = np.concatenate(mats, axis=0) if mats else np.zeros((0, 1), np.float32)
            offsets = np.zeros(len(mats) + 1, dtype=np.int64)
            offsets[1:] = np.cumsum([m.shape[0] for m in mats])
        X = _lib.f32_matrix(X)
        if offsets is None:
            offsets = np.array([0, X.shape[0]], dtype=np.int64)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        h = lib().sr_batch_from_features(_lib.as_fp(X), X.shape[0], X.shape[1],
                                         _lib.as_i64p(offsets), len(offsets) - 1)
        return cls(h)

    def update_pcm(self, samples) -> None:
        """Overwrite the batch's int16 samples in place (same layout): the serving loop's H2D."""
        a = np.ascontiguousarray(samples, dtype=np.int16)
        check(lib().sr_batch_update_pcm(self._h, a.ctypes.data_as(C.POINTER(C.c_int16)), a.size),
              "sr_batch_update_pcm")

    def reset_pcm(self, signals) -> None:
        """New int16 signals with a new layout into the same device buffers (they only grow)."""
        sigs = [np.ascontiguousarray(x, dtype=np.int16) for x in signals]
        offsets = np.zeros(len(sigs) + 1, dtype=np.int64)
        offsets[1:] = np.cumsum([len(x) for x in sigs])
        cat = np.ascontiguousarray(np.concatenate(sigs)) if sigs else np.zeros(0, np.int16)
        check(lib().sr_batch_reset_pcm(self._h, cat.ctypes.data_as(C.POINTER(C.c_int16)), _lib.as_i64p(offsets),
                                       len(sigs)), "sr_batch_reset_pcm")

    def reset_features(self, X) -> None:
        """One [T, dim] matrix as the batch's single utterance, into the same device buffers (they only grow): what the
        reference's per-utterance loop (gmmset.py:62-64) costs nothing but the upload with."""
        X = _lib.f32_matrix(X)
        offsets = np.array([0, X.shape[0]], dtype=np.int64)
        check(lib().sr_batch_reset_features(self._h, _lib.as_fp(X), X.shape[0], X.shape[1], _lib.as_i64p(offsets), 1),
              "sr_batch_reset_features")

    @property
    def n_utt(self) -> int:
        return lib().sr_batch_num_utterances(self._h)

    @property
    def n_rows(self) -> int:
        return int(lib().sr_batch_num_rows(self._h))

    @property
    def dim(self) -> int:
        return lib().sr_batch_dim(self._h)

    def offsets(self) -> np.ndarray:
        out = np.zeros(self.n_utt + 1, dtype=np.int64)
        check(lib().sr_batch_offsets(self._h, _lib.as_i64p(out)), "sr_batch_offsets")
        return out

    def download(self) -> np.ndarray:
        out = np.empty((self.n_rows, self.dim), dtype=np.float32)
        check(lib().sr_batch_download(self._h, _lib.as_fp(out)), "sr_batch_download")
        return out

    def __del__(self):
        try:
            if self._owner and self._h:
                lib().sr_batch_free(self._h)
                self._h = None
        except Exception:
            pass


class ModelSet:
    """S models packed once and resident on the device (one fused scoring launch for all)."""

    def __init__(self, gmms):
        self._keep = list(gmms)
        arr = (C.c_void_p * len(self._keep))(*[g.gmm for g in self._keep])
        h = lib().sr_modelset_create(arr, len(self._keep))
        if not h:
            raise SRError("sr_modelset_create failed: %s" % _lib.last_error())
        self._h = C.c_void_p(h)

    def __len__(self):
        return lib().sr_modelset_size(self._h)

    @property
    def dim(self) -> int:
        return lib().sr_modelset_dim(self._h)

    def info(self) -> dict:
        """Conditioning of the packed set as the engine dispatcher sees it (sr_modelset_info)."""
        out = np.zeros(8)
        check(lib().sr_modelset_info(self._h, _lib.as_dp(out)), "sr_modelset_info")
        return {"amp": out[0], "pad_waste": out[1], "sigma_ratio": out[2], "coef_max": out[3],
                "shared_sigma": bool(out[4]), "models": int(out[5]), "device": int(out[6]),
                "hybrid_vector_mixtures": int(out[7])}

    def score(self, feats: Batch, frame_ll: bool = False, clamp_compat: bool = True):
        """-> (sums[U, S] float64, argmax[U] int32[, frame_ll[S, n] float32])."""
        U, S = feats.n_utt, len(self)
        sums = np.zeros((U, S), dtype=np.float64)
        arg = np.full(U, -1, dtype=np.int32)
        fll = np.empty((S, feats.n_rows), dtype=np.float32) if frame_ll else None
        check(lib().sr_score_batch_set(self._h, feats._h, _lib.as_dp(sums), _lib.as_i32p(arg),
                                       _lib.as_fp(fll) if frame_ll else None,
                                       _lib.SR_CLAMP_COMPAT if clamp_compat else 0),
              "sr_score_batch_set")
        return (sums, arg, fll) if frame_ll else (sums, arg)

    def __del__(self):
        try:
            if self._h:
                lib().sr_modelset_free(self._h)
                self._h = None
        except Exception:
            pass


class MfccExtractor:
    """Device MFCC extractor; constants as MFCCExtractor.__init__ (src/feature/MFCC.py:20-41)."""

    def __init__(self, fs, win_length_ms=32, win_shift_ms=16, FFT_SIZE=2048, n_filters=50,
                 n_ceps=13, pre_emphasis_coef=0.95, n_lpc=0):
        h = lib().sr_mfcc_create(float(fs), float(win_length_ms), float(win_shift_ms),
                                 int(FFT_SIZE), int(n_filters), int(n_ceps), float(pre_emphasis_coef))
        if not h:
            raise SRError("sr_mfcc_create failed: %s" % _lib.last_error())
        self._h = C.c_void_p(h)
        self.n_lpc = int(n_lpc)        # > 0: every frame also carries LPC-n_lpc columns (mix_feature)
        if self.n_lpc:
            check(lib().sr_mfcc_set_lpc(self._h, self.n_lpc), "sr_mfcc_set_lpc")
        self.fs, self.FFT_SIZE, self.n_bands, self.coefs = fs, FFT_SIZE, n_filters, n_ceps
        self.PRE_EMPH = pre_emphasis_coef
        self.FRAME_LEN = lib().sr_mfcc_frame_len(self._h)
        self.FRAME_SHIFT = lib().sr_mfcc_frame_shift(self._h)

    def tables(self):
        """Host float64 constants (window, mel bank M, DCT rows D) -- for tests."""
        win = np.empty(self.FRAME_LEN)
        M = np.empty((self.n_bands, self.FFT_SIZE // 2 + 1))
        D = np.empty((self.coefs, self.n_bands))
        check(lib().sr_mfcc_tables(self._h, _lib.as_dp(win), _lib.as_dp(M), _lib.as_dp(D)), "sr_mfcc_tables")
        return win, M, D

    def num_frames(self, n_samples: int) -> int:
        return int(lib().sr_mfcc_num_frames(self._h, int(n_samples)))

    def extract_batch(self, pcm: Batch, nd: int = 0, cmvn: bool = True) -> Batch:
        h = lib().sr_mfcc_extract_batch(self._h, pcm._h, int(nd), 1 if cmvn else 0)
        if not h:
            raise SRError("sr_mfcc_extract_batch failed: %s" % _lib.last_error())
        return Batch(h)

    def extract(self, signal, nd: int = 0, cmvn: bool = True) -> np.ndarray:
        """One utterance -> float64 [T - nd, n_ceps*(nd+1)] (what MFCCExtractor.extract returns,
        MFCC.py:49-79, plus optional deltas)."""
        signal = np.asarray(signal)
        if signal.ndim > 1:
            signal = np.mean(signal, axis=1)                      # MFCC.py:53-55
        assert len(signal) > 5 * self.FRAME_LEN, "Signal too short!"  # MFCC.py:56
        out = self.extract_batch(Batch.from_pcm([signal]), nd, cmvn)
        return out.download().astype(np.float64)

    def predict_batch(self, models: ModelSet, pcm: Batch, nd: int = 0, clamp_compat: bool = True, out=None):
        """Fused serving step on resident PCM: MFCC -> CMVN/deltas -> scoring -> argmax.
        ``out`` = (float64[U, S], int32[U]) C-contiguous arrays to fill instead of fresh ones: a serving loop that keeps its result
        buffers spares the page faults of 8 U S new bytes per call (and may page-lock them once with ``_lib.host_register``, so that
        the results leave the device by DMA)."""
        U, S = pcm.n_utt, len(models)
        if out is None:
            sums = np.zeros((U, S), dtype=np.float64)
            arg = np.full(U, -1, dtype=np.int32)
        else:
            sums, arg = out
            if sums.shape != (U, S) or sums.dtype != np.float64 or not sums.flags.c_contiguous or arg.shape != (U,) or \
                    arg.dtype != np.int32 or not arg.flags.c_contiguous:
                raise ValueError("out must be (float64[%d, %d], int32[%d]), C-contiguous" % (U, S, U))
        check(lib().sr_predict_pcm_batch(self._h, models._h, pcm._h, int(nd), _lib.as_dp(sums),
                                         _lib.as_i32p(arg), _lib.SR_CLAMP_COMPAT if clamp_compat else 0),
              "sr_predict_pcm_batch")
        return sums, arg

    def __del__(self):
        try:
            if self._h:
                lib().sr_mfcc_free(self._h)
                self._h = None
        except Exception:
            pass


class ServingStream:
    """Double-buffered fixed-shape serving session (sr_stream_*): ``n_windows`` windows of
    ``window_samples`` int16 samples per tick; ``submit`` queues a tick (H2D on its own HIP stream,
    overlapping the previous tick's kernels), ``collect`` returns the oldest tick's decisions.
    ``graph=True`` replays each tick's kernels and result copies as one captured hipGraph."""

    def __init__(self, extractor: MfccExtractor, models: ModelSet, n_windows: int, window_samples: int,
                 nd: int = 0, clamp_compat: bool = True, graph: bool = False):
        self._keep = (extractor, models)
        self.n_windows, self.window_samples, self.n_models = int(n_windows), int(window_samples), len(models)
        h = lib().sr_stream_create(extractor._h, models._h, self.n_windows, self.window_samples, int(nd),
                                   (_lib.SR_CLAMP_COMPAT if clamp_compat else 0) | (_lib.SR_STREAM_GRAPH if graph else 0))
        if not h:
            raise SRError("sr_stream_create failed: %s" % _lib.last_error())
        self._h = C.c_void_p(h)

    def submit(self, pcm) -> None:
        a = np.ascontiguousarray(pcm, dtype=np.int16)
        if a.size != self.n_windows * self.window_samples:
            raise ValueError("expected %d x %d samples" % (self.n_windows, self.window_samples))
        check(lib().sr_stream_submit(self._h, a.ctypes.data_as(C.POINTER(C.c_int16))), "sr_stream_submit")

    def collect(self):
        sums = np.empty((self.n_windows, self.n_models), dtype=np.float64)
        arg = np.empty(self.n_windows, dtype=np.int32)
        ms = C.c_double(0)
        check(lib().sr_stream_collect(self._h, _lib.as_dp(sums), _lib.as_i32p(arg), C.byref(ms)), "sr_stream_collect")
        return sums, arg, ms.value

    def __del__(self):
        try:
            if self._h:
                lib().sr_stream_free(self._h)
                self._h = None
        except Exception:
            pass


class MultiPredictor:
    """Every GPU of the node from one process: utterances are dealt to ``n_slots`` slots by length
    (slot i on device i % device_count), each slot has its own replica of the models and a host thread
    that runs MFCC -> CMVN/deltas -> all models -> sums + argmax on its GPU; rows are gathered on the
    host (``sr_multi_*``; the reference: Threadpool in gmm.cc:533-560, Pool in test-gmm.py:128-133)."""

    def __init__(self, gmms, fs, n_slots=0, win_length_ms=32, win_shift_ms=16, FFT_SIZE=2048, n_filters=50,
                 n_ceps=13, pre_emphasis_coef=0.95):
        self._keep = list(gmms)
        arr = (C.c_void_p * len(self._keep))(*[g.gmm for g in self._keep])
        h = lib().sr_multi_create(arr, len(self._keep), float(fs), float(win_length_ms), float(win_shift_ms),
                                  int(FFT_SIZE), int(n_filters), int(n_ceps), float(pre_emphasis_coef), int(n_slots))
        if not h:
            raise SRError("sr_multi_create failed: %s" % _lib.last_error())
        self._h = C.c_void_p(h)
        self.n_models = len(self._keep)
        self.slot_seconds = None

    @property
    def n_slots(self) -> int:
        return lib().sr_multi_slots(self._h)

    def slot_devices(self):
        return [lib().sr_multi_slot_device(self._h, i) for i in range(self.n_slots)]

    def slot_numa_nodes(self):
        """NUMA node every slot's host thread pinned itself to in the last call (-1: the platform does not say)."""
        return [lib().sr_multi_slot_numa_node(self._h, i) for i in range(self.n_slots)]

    def predict(self, signals, nd=0, clamp_compat=True):
        """``signals``: list of int16 arrays.  -> (sums[U, S], argmax[U])."""
        sigs = [np.ascontiguousarray(s, dtype=np.int16) for s in signals]
        offsets = np.zeros(len(sigs) + 1, dtype=np.int64)
        offsets[1:] = np.cumsum([len(s) for s in sigs])
        cat = np.ascontiguousarray(np.concatenate(sigs)) if sigs else np.zeros(0, np.int16)
        return self.predict_concat(cat, offsets, nd, clamp_compat)

    def predict_concat(self, cat, offsets, nd=0, clamp_compat=True):
        U = len(offsets) - 1
        sums = np.zeros((U, self.n_models), dtype=np.float64)
        arg = np.full(U, -1, dtype=np.int32)
        secs = np.zeros(self.n_slots, dtype=np.float64)
        check(lib().sr_multi_predict_pcm(self._h, cat.ctypes.data_as(C.POINTER(C.c_int16)), _lib.as_i64p(offsets), U,
                                         int(nd), _lib.as_dp(sums), _lib.as_i32p(arg), _lib.as_dp(secs),
                                         _lib.SR_CLAMP_COMPAT if clamp_compat else 0), "sr_multi_predict_pcm")
        self.slot_seconds = secs
        return sums, arg

    def __del__(self):
        try:
            if self._h:
                lib().sr_multi_free(self._h)
                self._h = None
        except Exception:
            pass
