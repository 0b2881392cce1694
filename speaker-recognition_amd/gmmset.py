"""Speaker set -- mirrors the reference's ``src/testbench/gmmset.py`` (GMMSet :15-91,
GMMSetPyGMM :94-105): one GMM per label, prediction = argmax over speakers of the summed
per-frame log-likelihood, optional UBM rejection.

The reference scores speaker by speaker through the ABI (gmmset.py:59-64, :95-99).  Here all
speaker models are packed into one device-resident set and scored in ONE fused launch: every
frame tile is read from HBM once and walked over all S models.
"""
from __future__ import annotations

import operator
from collections import defaultdict

import numpy as np

from .core import Batch, ModelSet
from .pygmm import GMM


class GMMSet(object):
    def __init__(self, gmm_order=32, ubm=None, reject_threshold=10, **kwargs):
        self.kwargs = kwargs
        self.gmms = []
        self.ubm = ubm
        self.reject_threshold = reject_threshold
        if ubm is not None:
            self.gmm_order = ubm.get_nr_mixtures()
        else:
            self.gmm_order = gmm_order
        self.y = []
        self._set = None            # packed device copy of self.gmms (rebuilt when the list changes)

    # ---- enrolment ----
    def fit_new(self, x, label):
        self.y.append(label)
        gmm = GMM(self.gmm_order, **self.kwargs)
        gmm.fit(x, self.ubm)
        self.gmms.append(gmm)
        self._set = None

    def cluster_by_label(self, X, y):
        Xtmp = defaultdict(list)
        for ind, x in enumerate(X):
            Xtmp[y[ind]].extend(x)
        yp, Xp = zip(*Xtmp.items())
        return Xp, yp

    def auto_tune_parameter(self, X, y):
        return                       # a TODO in the reference as well (gmmset.py:45-48)

    def fit(self, X, y):
        X, y = self.cluster_by_label(X, y)
        for ind, x in enumerate(X):
            self.fit_new(x, y[ind])
        self.auto_tune_parameter(X, y)

    def load_gmm(self, label, fname):
        self.y.append(label)
        gmm = GMM.load(fname)
        for key, val in self.kwargs.items():
            setattr(gmm, key, val)
        self.gmms.append(gmm)
        self._set = None

    # ---- scoring ----
    def _model_set(self):
        if self._set is None or len(self._set) != len(self.gmms):
            self._set = ModelSet(self.gmms)
        return self._set

    def gmm_score(self, gmm, x):
        return np.sum(gmm.score(x))

    def predict_one_scores(self, x):
        """Summed log-likelihood of utterance x under every speaker model (one launch)."""
        sums, _ = self._model_set().score(Batch.from_features([x]))
        return list(sums[0])

    def predict_one(self, x):
        scores = self.predict_one_scores(x)
        return self.y[max(enumerate(scores), key=operator.itemgetter(1))[0]]   # first maximum wins

    def predict(self, X):
        """All utterances in one batch; the argmax comes back from the device."""
        X = list(X)
        if not X:
            return []
        _, arg = self._model_set().score(Batch.from_features(X))
        return [self.y[i] if i >= 0 else None for i in arg]

    def predict_one_with_rejection(self, x):
        assert self.ubm is not None, "UBM must be given prior to conduct reject prediction."
        scores = self.predict_one_scores(x)
        x_len = len(x)               # normalize score
        scores = [v / x_len for v in scores]
        max_tup = max(enumerate(scores), key=operator.itemgetter(1))
        ubm_score = self.gmm_score(self.ubm, x) / x_len
        if max_tup[1] - ubm_score < self.reject_threshold:
            return None
        return self.y[max_tup[0]]

    def predict_with_reject(self, X):
        return [self.predict_one_with_rejection(x) for x in X]


class GMMSetPyGMM(GMMSet):
    def predict_one(self, x):
        scores = [s / len(x) for s in self.predict_one_scores(x)]              # gmmset.py:96
        return self.y[max(enumerate(scores), key=operator.itemgetter(1))[0]]

    def before_pickle(self):
        self._set = None
        self.gmms = [x.dumps() for x in self.gmms]

    def after_pickle(self):
        self.gmms = [GMM.loads(x) for x in self.gmms]
        self._set = None

    def __getstate__(self):
        st = dict(self.__dict__)
        st["_set"] = None
        return st
