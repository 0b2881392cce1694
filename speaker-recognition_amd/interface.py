"""Model facade -- mirrors the reference's ``src/gui/interface.py`` (ModelInterface :26-109):
enroll / train / predict / dump / load with the same meaning.  The speaker set is the
device-backed ``GMMSetPyGMM`` (the reference's C++ back-end, interface.py:19-23,63-75, is the
drop-in boundary of this repo; its scikit-learn default is third-party code).

Extensions (all keyword-only, defaults = the reference's behaviour): ``gmm_order``,
``feature_kwargs`` (forwarded to the extractor), ``lpc`` (False: MFCC half of mix_feature only),
``diff``/``nd`` (append deltas to the MFCC half; excludes the LPC columns),
``gmm_kwargs`` (forwarded to ``pygmm.GMM``), ``ubm`` (MAP-adapt speakers from a UBM).
VAD (``init_noise`` / ``filter``) is the LTSD detector of ``filters`` (third-party pyssp in the
reference: restated, parity unpinned).
"""
from __future__ import annotations

import pickle
import sys
import time
import traceback as tb
from collections import defaultdict

import numpy as np

from .feature import mix_feature
from .gmmset import GMMSetPyGMM as GMMSet
from .pygmm import GMM


class ModelInterface(object):

    UBM_MODEL_FILE = None

    def __init__(self, *, gmm_order=32, feature_kwargs=None, diff=False, nd=1, lpc=True, gmm_kwargs=None,
                 verbose=True):
        self.features = defaultdict(list)
        self.gmm_order = gmm_order
        self.feature_kwargs = dict(feature_kwargs or {})
        self.diff, self.nd, self.lpc = diff, nd, lpc
        self.gmm_kwargs = dict(gmm_kwargs or {})
        self.verbose = verbose
        self.gmmset = GMMSet(gmm_order=gmm_order, **self.gmm_kwargs)

    def init_noise(self, fs, signal):
        """init vad from environment noise (gui/interface.py:37-41)"""
        from .filters import VAD
        if getattr(self, "vad", None) is None:
            self.vad = VAD()
        self.vad.init_noise(fs, signal)

    def filter(self, fs, signal):
        """use VAD to filter out the silent part of a signal; empty if less than a third of it is
        voiced (gui/interface.py:43-53)"""
        if getattr(self, "vad", None) is None:
            raise RuntimeError("NoiseFilter Not Initialized")
        ret, intervals = self.vad.filter(fs, signal)
        if len(ret) > len(signal) / 3:
            return ret
        return np.array([])

    def _features(self, fs, signal):
        return mix_feature((fs, signal), lpc=self.lpc, diff=self.diff, nd=self.nd, **self.feature_kwargs)

    def enroll(self, name, fs, signal):
        """add the signal to this person's training dataset"""
        feat = self._features(fs, signal)
        self.features[name].extend(feat)

    def _get_gmm_set(self):
        import os
        if self.UBM_MODEL_FILE and os.path.isfile(self.UBM_MODEL_FILE):
            return GMMSet(ubm=GMM.load(self.UBM_MODEL_FILE), **self.gmm_kwargs)
        return GMMSet(gmm_order=self.gmm_order, **self.gmm_kwargs)

    def train(self):
        self.gmmset = self._get_gmm_set()
        start = time.time()
        if self.verbose:
            print("Start training...")
        for name, feats in self.features.items():
            self.gmmset.fit_new(np.asarray(feats), name)
        if self.verbose:
            print(time.time() - start, " seconds")

    def predict(self, fs, signal):
        """return a label (name)"""
        try:
            feat = self._features(fs, signal)
        except Exception:
            print(tb.format_exc(), file=sys.stderr)
            return None
        return self.gmmset.predict_one(feat)

    def predict_many(self, items, gpus=1):
        """Extension: [(fs, signal), ...] -> labels, every utterance scored in one batch.  ``gpus`` != 1
        (0 = every visible GPU) shards the utterances over the GPUs of the node from this one process
        (core.MultiPredictor: a host thread and a model replica per GPU, no collective) -- for the MFCC-only
        feature (``lpc=False``) on int16 audio of one sampling rate; anything else takes the one-GPU path."""
        items = list(items)
        rates = {fs for fs, _ in items}
        if gpus != 1 and not self.lpc and len(rates) == 1 and items and \
                all(np.asarray(sig).dtype == np.int16 and np.asarray(sig).ndim == 1 for _, sig in items):
            from .core import MultiPredictor
            kw = dict(self.feature_kwargs)
            fs = rates.pop()
            # the per-GPU model replicas are packed and uploaded once and kept until the model set changes (re-training
            # replaces the GMM objects), not rebuilt on every call
            key = (tuple(id(g) for g in self.gmmset.gmms), fs, int(gpus), tuple(sorted(kw.items())))
            cached = getattr(self, "_multi", None)
            if cached is None or cached[0] != key:
                cached = self._multi = (key, MultiPredictor(self.gmmset.gmms, fs, n_slots=int(gpus), **kw))
            mp = cached[1]
            _, winners = mp.predict([sig for _, sig in items], nd=self.nd if self.diff else 0)
            return [None if w < 0 else self.gmmset.y[w] for w in winners]
        feats = [self._features(fs, sig) for fs, sig in items]
        return self.gmmset.predict(feats)

    def dump(self, fname):
        """ dump all models to file"""
        self.gmmset.before_pickle()
        multi, self._multi = getattr(self, "_multi", None), None        # (device handles do not pickle)
        try:
            with open(fname, "wb") as f:
                pickle.dump(self, f, -1)
        finally:
            self._multi = multi
            self.gmmset.after_pickle()

    @staticmethod
    def load(fname):
        """ load from a dumped model file"""
        with open(fname, "rb") as f:
            R = pickle.load(f)
            R.gmmset.after_pickle()
            return R
