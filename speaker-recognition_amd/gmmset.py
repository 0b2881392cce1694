"""Speaker set -- API mirror of the reference's ``src/testbench/gmmset.py`` (``GMMSet`` :15-91,
``GMMSetPyGMM`` :94-105): one GMM per label, prediction = arg max over speakers of the summed
per-frame log-likelihood, optional open-set rejection against a UBM.  Same method names, arguments
and results; the implementation is this package's.

The reference scores speaker by speaker through the ABI (gmmset.py:59-64, :95-99).  Here every
speaker model sits in one device-resident ``ModelSet`` and an utterance -- or a whole list of them
-- is scored against all of them in ONE fused launch: each frame tile leaves HBM once.
"""
from __future__ import annotations

import ctypes as C
import threading
from collections import OrderedDict

import numpy as np

from . import _lib
from .core import Batch, ModelSet
from .pygmm import GMM


class GMMSet(object):
    def __init__(self, gmm_order=32, ubm=None, reject_threshold=10, **kwargs):
        self.ubm, self.reject_threshold, self.kwargs = ubm, reject_threshold, kwargs
        # speakers adapted from a UBM inherit its size (gmmset.py:24-27)
        self.gmm_order = gmm_order if ubm is None else ubm.get_nr_mixtures()
        self.gmms, self.y = [], []
        self._set = None
        self._set_key = None            # packed device copy of self.gmms, rebuilt when the list changes

    # ---- enrolment ----
    def _append(self, label, gmm):
        self.gmms.append(gmm)
        self.y.append(label)
        self._set = None

    def fit_new(self, x, label):
        """Train one model on the frames ``x`` (EM, or MAP adaptation when a UBM was given)."""
        model = GMM(self.gmm_order, **self.kwargs)
        model.fit(x, self.ubm)
        self._append(label, model)

    def cluster_by_label(self, X, y):
        """Pool the frame lists of equal labels -> (tuple of frame lists, tuple of labels)."""
        pooled = OrderedDict()
        for frames, label in zip(X, y):
            pooled.setdefault(label, []).extend(frames)
        return tuple(pooled.values()), tuple(pooled.keys())

    def auto_tune_parameter(self, X, y):
        return None                  # unimplemented in the reference as well (gmmset.py:44-47)

    def fit(self, X, y):
        frames, labels = self.cluster_by_label(X, y)
        for f, lab in zip(frames, labels):
            self.fit_new(f, lab)
        self.auto_tune_parameter(frames, labels)

    def load_gmm(self, label, fname):
        model = GMM.load(fname)
        for name, value in self.kwargs.items():
            setattr(model, name, value)
        self._append(label, model)

    # ---- scoring ----
    def _model_set(self):
        # the packed device copy is valid for exactly these model objects in this state: refitting
        # a model in place, replacing an element or reloading one must not leave a stale copy
        key = tuple((id(g), getattr(g, "_version", 0)) for g in self.gmms)
        if self._set is None or self._set_key != key:
            self._set = ModelSet(self.gmms)
            self._set_key = key
        return self._set

    def gmm_score(self, gmm, x):
        return float(np.sum(gmm.score(x)))

    def predict_one_scores(self, x):
        """Summed log-likelihood of utterance ``x`` under every speaker model (one launch)."""
        if _lib.gpu_runtime_lost():
            # a worker forked after the parent used the GPU (the reference's fit-then-Pool drivers, test-nperson.py:126-139):
            # device-resident sets cannot exist here; the reference's own speaker-by-speaker loop (gmmset.py:59-64) can --
            # each call is served by this process's helper (csrc/fork_proxy.cpp)
            # -- in ONE conversation and one fused pass in the helper for the whole set (sr_score_models_f32; the per-speaker loop was
            # a conversation, a launch chain and a reply per speaker: 80 per utterance in the reference's logged run)
            X = _lib.f32_matrix(x)
            handles = (C.c_void_p * len(self.gmms))(*[g.gmm for g in self.gmms])
            sums = np.zeros(len(self.gmms))
            _lib.check(_lib.lib().sr_score_models_f32(handles, len(self.gmms), _lib.as_fp(X), X.shape[0], X.shape[1],
                                                      _lib.as_dp(sums), _lib.SR_CLAMP_COMPAT), "sr_score_models_f32")
            return sums.tolist()
        # one utterance at a time is how the reference's drivers call (gmmset.py:62-64, gui.py:179-214): the device batch is kept and
        # refilled, so such a loop allocates nothing
        # (a batch per calling thread: the calls below release the GIL)
        mine = self.__dict__.setdefault("_scratch", {})
        scratch = mine.get(threading.get_ident())
        if scratch is None:
            scratch = mine[threading.get_ident()] = Batch.from_features([x])
        else:
            scratch.reset_features(x)
        totals, _ = self._model_set().score(scratch)
        return totals[0].tolist()

    def _label_of_best(self, scores):
        return self.y[int(np.argmax(scores))]          # numpy's argmax keeps the first maximum, as
                                                        # max(enumerate(...)) does (gmmset.py:62-64)

    def predict_one(self, x):
        return self._label_of_best(self.predict_one_scores(x))

    def predict(self, X):
        """All utterances in one batch; the arg max comes back from the device."""
        utterances = list(X)
        if not utterances:
            return []
        if _lib.gpu_runtime_lost():
            return [self.predict_one(x) for x in utterances]
        _, winners = self._model_set().score(Batch.from_features(utterances))
        return [None if w < 0 else self.y[w] for w in winners]

    def predict_one_with_rejection(self, x):
        """Open-set decision (gmmset.py:69-81): per-frame margin of the best speaker over the UBM
        below ``reject_threshold`` -> None."""
        if self.ubm is None:
            raise AssertionError("UBM must be given prior to conduct reject prediction.")
        n = float(len(x))
        per_frame = np.asarray(self.predict_one_scores(x)) / n
        best = int(np.argmax(per_frame))
        margin = per_frame[best] - self.gmm_score(self.ubm, x) / n
        return self.y[best] if margin >= self.reject_threshold else None

    def predict_with_reject(self, X):
        return [self.predict_one_with_rejection(x) for x in X]


class GMMSetPyGMM(GMMSet):
    def predict_one(self, x):
        # the reference divides every total by the frame count first (gmmset.py:96); same winner
        return self._label_of_best(np.asarray(self.predict_one_scores(x)) / float(len(x)))

    # models travel through pickle as their text dumps (gmmset.py:101-105)
    def before_pickle(self):
        self._set = None
        self.__dict__.pop("_scratch", None)
        self.gmms = [m.dumps() for m in self.gmms]

    def after_pickle(self):
        self._set = None
        self.gmms = [GMM.loads(text) for text in self.gmms]

    def __getstate__(self):
        state = dict(self.__dict__)
        state["_set"] = None
        state.pop("_scratch", None)
        return state
