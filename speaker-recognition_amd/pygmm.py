"""Python surface of the GMM engine -- mirrors the reference's ``src/gmm/python/pygmm.py``
(class GMM :38-141, GMMParameter :18-27) on top of lib/pygmm.so's HIP path.

Same constructor keywords and methods (fit / score / score_all / dump / load / dumps / loads /
get_dim / get_nr_mixtures).  Differences: Python 3; frames go to the device as one contiguous
fp32 matrix instead of one ctypes array per frame (pygmm.py:89-95, the dominant cost in the
reference's logs); ``dumps``/``loads`` do not round-trip through fixed /tmp files
(pygmm.py:71-86); handles are freed with the object.
"""
from __future__ import annotations

import ctypes as C
from multiprocessing import cpu_count

import numpy as np

from . import _lib
from ._lib import Parameter as GMMParameter  # noqa: F401  (name kept from pygmm.py:18)
from ._lib import SRError, check, lib

COVTYPE_SPHEREICAL, COVTYPE_DIAGONAL, COVTYPE_FULL = 0, 1, 2   # pygmm.py:33-35 (sic)


class GMM(object):
    def __init__(self, nr_mixture=10, covariance_type=COVTYPE_DIAGONAL, min_covar=1e-3,
                 threshold=0.01, nr_iteration=200, init_with_kmeans=0, concurrency=cpu_count(),
                 verbosity=0, seed=-1, _handle=None):
        self.nr_mixture = nr_mixture
        self.covariance_type = covariance_type
        self.min_covar = min_covar
        self.threshold = threshold
        self.nr_iteration = nr_iteration
        self.init_with_kmeans = init_with_kmeans
        self.concurrency = concurrency          # accepted for compatibility; the HIP grid ignores it
        self.verbosity = verbosity
        self.seed = seed                        # extension: reproducible initialisation
        if _handle is None:
            _handle = lib().new_gmm(int(nr_mixture), int(covariance_type))
            if not _handle:
                raise SRError("new_gmm failed: %s" % _lib.last_error())
        self.gmm = C.c_void_p(_handle)

    # ---- persistence (text format of gmm.cc:655-682) ----
    @staticmethod
    def load(model_file):
        h = lib().load(str(model_file).encode())
        if not h:
            raise SRError("load failed: %s" % _lib.last_error())
        g = GMM(_handle=h)
        g.nr_mixture = lib().get_nr_mixtures(g.gmm)
        return g

    def dump(self, model_file):
        # through dumps(): the legacy `dump` symbol returns void (pygmm.hh:32), so a failed write
        # would only reach stderr; here it raises
        text = self.dumps()
        with open(str(model_file), "w") as f:
            f.write(text)

    def dumps(self):
        need = C.c_long(0)
        check(lib().sr_gmm_dumps(self.gmm, None, 0, C.byref(need)), "sr_gmm_dumps")
        buf = C.create_string_buffer(need.value)
        check(lib().sr_gmm_dumps(self.gmm, buf, need.value, None), "sr_gmm_dumps")
        return buf.value.decode()

    @staticmethod
    def loads(s):
        if isinstance(s, str):
            s = s.encode()
        h = lib().sr_gmm_loads(s)
        if not h:
            raise SRError("loads failed: %s" % _lib.last_error())
        g = GMM(_handle=h)
        g.nr_mixture = lib().get_nr_mixtures(g.gmm)
        return g

    @staticmethod
    def from_arrays(weights, mean, sigma):
        """Extension: build a model from float64 arrays (sigma = standard deviations)."""
        w = np.ascontiguousarray(weights, dtype=np.float64)
        mu = np.ascontiguousarray(mean, dtype=np.float64)
        sg = np.ascontiguousarray(sigma, dtype=np.float64)
        h = lib().sr_gmm_from_arrays(mu.shape[0], mu.shape[1], _lib.as_dp(w), _lib.as_dp(mu), _lib.as_dp(sg))
        if not h:
            raise SRError("sr_gmm_from_arrays failed: %s" % _lib.last_error())
        g = GMM(_handle=h)
        g.nr_mixture = mu.shape[0]
        return g

    def params(self):
        K, D = self.get_nr_mixtures(), self.get_dim()
        w, mu, sg = np.empty(K), np.empty((K, D)), np.empty((K, D))
        check(lib().sr_gmm_get_params(self.gmm, _lib.as_dp(w), _lib.as_dp(mu), _lib.as_dp(sg)), "sr_gmm_get_params")
        return w, mu, sg

    # ---- training (pygmm.py:97-117) ----
    def _gen_param(self, X):
        p = GMMParameter()
        p.nr_instance, p.nr_dim = int(X.shape[0]), int(X.shape[1])
        p.nr_mixture = int(self.nr_mixture)
        p.min_covar, p.threshold = float(self.min_covar), float(self.threshold)
        p.nr_iteration, p.init_with_kmeans = int(self.nr_iteration), int(self.init_with_kmeans)
        p.concurrency, p.verbosity = int(self.concurrency), int(self.verbosity)
        return p

    def fit(self, X, ubm=None):
        """:param ubm: None or a GMM instance (MAP adaptation of the means, gmmubm.cc)."""
        X = _lib.f32_matrix(X)
        p = self._gen_param(X)
        n_iter = lib().sr_train_f32(self.gmm, ubm.gmm if ubm is not None else None, _lib.as_fp(X),
                                    X.shape[0], X.shape[1], C.byref(p), int(self.seed))
        check(n_iter, "train")
        self.nr_mixture = lib().get_nr_mixtures(self.gmm)
        self._version = getattr(self, "_version", 0) + 1      # invalidates packed copies (GMMSet._model_set)
        return n_iter

    # ---- scoring (pygmm.py:120-132) ----
    def score(self, X):
        """Per-frame log-likelihoods (what score_batch fills), float64[n]."""
        X = _lib.f32_matrix(X)
        out = np.empty(X.shape[0], dtype=np.float32)
        check(lib().sr_score_frames_f32(self.gmm, _lib.as_fp(X), X.shape[0], X.shape[1],
                                        _lib.as_fp(out), None, _lib.SR_CLAMP_COMPAT), "score")
        return out.astype(np.float64)

    def score_all(self, X):
        X = _lib.f32_matrix(X)
        s = C.c_double(0)
        check(lib().sr_score_frames_f32(self.gmm, _lib.as_fp(X), X.shape[0], X.shape[1], None,
                                        C.byref(s), _lib.SR_CLAMP_COMPAT), "score_all")
        return s.value

    def get_dim(self):
        return lib().get_dim(self.gmm)

    def get_nr_mixtures(self):
        return lib().get_nr_mixtures(self.gmm)

    def __del__(self):
        try:
            if self.gmm:
                lib().sr_free_gmm(self.gmm)
                self.gmm = None
        except Exception:
            pass

    # pickling support: the handle travels as the text model (what before_pickle does by hand,
    # gmmset.py:101-105)
    def __getstate__(self):
        st = dict(self.__dict__)
        st["gmm"] = self.dumps() if lib().get_dim(self.gmm) > 0 else None
        return st

    def __setstate__(self, st):
        text = st.pop("gmm")
        self.__dict__.update(st)
        if text is None:
            h = lib().new_gmm(int(self.nr_mixture), int(self.covariance_type))
        else:
            h = lib().sr_gmm_loads(text.encode())
        if not h:
            raise SRError("cannot restore a pickled GMM: %s" % _lib.last_error())
        self.gmm = C.c_void_p(h)
