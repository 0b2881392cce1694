"""oracle/gmm_oracle.py -- TEST INFRASTRUCTURE ONLY.

ctypes front-ends for the two CPU checkers of the GMM scoring path:

* ``libgmm_oracle.so``  -- our plain-C restatement (oracle/gmm_oracle.c);
* ``_ref/pygmm_ref.so`` -- the reference's own C++ (``/root/reference/src/gmm/src``),
  compiled by ``oracle/Makefile`` from the sources where they lie.

Only tests/, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of bench.py may import
this module.  The product package never does.

Model text format follows ``GMM::load`` / ``Gaussian::load``
(/root/reference/src/gmm/src/gmm.cc:664-682, :125-150): ``K``, then K weights, then per
Gaussian ``dim covtype``, ``dim`` means, ``dim`` sigmas (standard deviations).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "libgmm_oracle.so")
REF_SO = os.path.join(HERE, "_ref", "pygmm_ref.so")

MODE_FASTEXP = 0   # what score_batch/score_all compute (gmm.cc:237-244 + fastexp.cc:99-212)
MODE_LIBM = 1      # GMM::log_probability_of (gmm.cc:229-235)
MODE_LOGSUMEXP = 2 # float64 log-sum-exp (the HIP kernel's formulation)
MODE_FAST = 3      # mode 0's result (to remez5's 1.2e-6) at a fraction of its cost: bulk parity checks (gmm_oracle.c score_batch_fast)

LN_1E_15 = float(np.log(1e-15))          # safe_log floor, gmm.cc:34-38
MINLOG = -7.08396418532264106224e2        # fastexp.cc:93 / :105


def build(ref: bool = True) -> None:
    """Compile the checkers (gcc only; the reference DSO only when /root/reference exists)."""
    subprocess.check_call(["make", "-s", "-C", HERE, "libgmm_oracle.so"])
    if ref and os.path.isdir("/root/reference/src/gmm/src"):
        subprocess.check_call(["make", "-s", "-C", HERE, "ref"])


@dataclass
class GMMParams:
    """Plain container: weights[K], mean[K,D], sigma[K,D] (float64)."""
    weights: np.ndarray
    mean: np.ndarray
    sigma: np.ndarray

    @property
    def K(self) -> int:
        return int(self.weights.shape[0])

    @property
    def D(self) -> int:
        return int(self.mean.shape[1])


def parse_model_text(text: str) -> GMMParams:
    """Token-stream parse, exactly as ``istream >>`` would (gmm.cc:664-682)."""
    tok = text.split()
    pos = 0
    K = int(tok[pos]); pos += 1
    w = np.array([float(t) for t in tok[pos:pos + K]], dtype=np.float64); pos += K
    means, sigmas = [], []
    for _ in range(K):
        dim = int(tok[pos]); cov = int(tok[pos + 1]); pos += 2
        if cov != 1:
            raise ValueError("only COVTYPE_DIAGONAL (1) models exist (gmm.hh:18-22)")
        means.append([float(t) for t in tok[pos:pos + dim]]); pos += dim
        sigmas.append([float(t) for t in tok[pos:pos + dim]]); pos += dim
    return GMMParams(w, np.array(means, dtype=np.float64), np.array(sigmas, dtype=np.float64))


def _fmt(v: float) -> str:
    # default ostream precision = 6 significant digits (%g), gmm.cc:101-123, :655-662
    return "%g" % v


def format_model_text(p: GMMParams) -> str:
    """What ``GMM::dump`` writes (gmm.cc:655-662 + Gaussian::dump :101-123)."""
    out = ["%d\n" % p.K, "".join(_fmt(w) + " " for w in p.weights) + "\n"]
    for k in range(p.K):
        out.append("%d 1\n" % p.D)
        out.append("".join(_fmt(v) + " " for v in p.mean[k]) + "\n")
        out.append("".join(_fmt(v) + " " for v in p.sigma[k]) + "\n")
    return "".join(out)


_oracle = None


def _lib():
    global _oracle
    if _oracle is None:
        if not os.path.exists(ORACLE_SO):
            build(ref=False)
        lib = C.CDLL(ORACLE_SO)
        dp = C.POINTER(C.c_double)
        lib.oracle_gmm_score_batch.argtypes = [dp, dp, dp, C.c_int, C.c_int, dp, C.c_long, dp,
                                               C.c_int, C.c_int, C.c_int]
        lib.oracle_gmm_score_batch.restype = None
        lib.oracle_gmm_score_all.argtypes = [dp, dp, dp, C.c_int, C.c_int, dp, C.c_long,
                                             C.c_int, C.c_int, C.c_int]
        lib.oracle_gmm_score_all.restype = C.c_double
        lib.oracle_gmm_em_iteration.argtypes = [dp, dp, dp, C.c_int, C.c_int, dp, C.c_long,
                                                C.c_double, C.c_double, dp, dp, C.c_int]
        lib.oracle_gmm_em_iteration.restype = None
        lib.oracle_set_flush_order.argtypes = [C.c_int]
        lib.oracle_set_flush_order.restype = None
        _oracle = lib
    return _oracle


def _dp(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def score_batch(p: GMMParams, X: np.ndarray, mode: int = MODE_FASTEXP, ftz: bool = True,
                clamp_compat: bool = True) -> np.ndarray:
    """Per-frame log-likelihoods, float64[n]."""
    X = np.ascontiguousarray(X, dtype=np.float64)
    n = X.shape[0]
    assert X.ndim == 2 and X.shape[1] == p.D
    w = np.ascontiguousarray(p.weights, dtype=np.float64)
    mu = np.ascontiguousarray(p.mean, dtype=np.float64)
    sg = np.ascontiguousarray(p.sigma, dtype=np.float64)
    out = np.empty(n, dtype=np.float64)
    _lib().oracle_gmm_score_batch(_dp(w), _dp(mu), _dp(sg), p.K, p.D, _dp(X), n, _dp(out),
                                  mode, int(ftz), int(clamp_compat))
    return out


def set_flush_order(order: int) -> None:
    """Which partial products mode 0 forms (and flushes): 2 = as the reference DSO's compiler does (default), 1 = the
    source's order; oracle/gmm_oracle.c gaussian_prob_fastexp."""
    _lib().oracle_set_flush_order(int(order))


def score_all(p: GMMParams, X: np.ndarray, mode: int = MODE_FASTEXP, ftz: bool = True,
              clamp_compat: bool = True) -> float:
    return float(np.sum(score_batch(p, X, mode, ftz, clamp_compat)))  # frame-order sum, gmm.cc:562-569


def em_iteration(p: GMMParams, X: np.ndarray, min_covar: float = 1e-3,
                 map_relevance: float = 0.0, ubm: GMMParams | None = None) -> GMMParams:
    """One EM (or MAP, when map_relevance>0) iteration; returns updated parameters."""
    X = np.ascontiguousarray(X, dtype=np.float64)
    n = X.shape[0]
    w = np.array(p.weights, dtype=np.float64, copy=True)
    mu = np.array(p.mean, dtype=np.float64, copy=True, order="C")
    sg = np.array(p.sigma, dtype=np.float64, copy=True, order="C")
    scratch = np.empty(p.K * n, dtype=np.float64)
    ubm_mean = np.ascontiguousarray(ubm.mean, dtype=np.float64) if ubm is not None else None
    _lib().oracle_gmm_em_iteration(_dp(w), _dp(mu), _dp(sg), p.K, p.D, _dp(X), n,
                                   float(min_covar), float(map_relevance),
                                   _dp(ubm_mean) if ubm_mean is not None else None,
                                   _dp(scratch), 1)
    return GMMParams(w, mu, sg)


# --------------------------------------------------------------------------------------
# The reference DSO itself (kind "reference").  Its -ffast-math start-up code flips the
# process to FTZ/DAZ on load, so callers that care load it in a subprocess.
# --------------------------------------------------------------------------------------

class RefLib:
    """Thin binding of the reference C ABI (/root/reference/src/gmm/src/pygmm.hh:28-41)."""

    def __init__(self, path: str = REF_SO):
        if not os.path.exists(path):
            raise FileNotFoundError(path + " (run `make -C oracle ref` where /root/reference exists)")
        lib = C.CDLL(path)
        pp = C.POINTER(C.POINTER(C.c_double))
        lib.load.restype = C.c_void_p
        lib.load.argtypes = [C.c_char_p]
        lib.new_gmm.restype = C.c_void_p
        lib.new_gmm.argtypes = [C.c_int, C.c_int]
        lib.dump.restype = None
        lib.dump.argtypes = [C.c_void_p, C.c_char_p]
        lib.score_all.restype = C.c_double
        lib.score_all.argtypes = [C.c_void_p, pp, C.c_int, C.c_int, C.c_int]
        lib.score_batch.restype = None
        lib.score_batch.argtypes = [C.c_void_p, pp, C.POINTER(C.c_double), C.c_int, C.c_int, C.c_int]
        lib.get_dim.restype = C.c_int
        lib.get_dim.argtypes = [C.c_void_p]
        lib.get_nr_mixtures.restype = C.c_int
        lib.get_nr_mixtures.argtypes = [C.c_void_p]
        lib.train_model.restype = None
        lib.train_model.argtypes = [C.c_void_p, pp, C.c_void_p]
        lib.train_model_from_ubm.restype = None
        lib.train_model_from_ubm.argtypes = [C.c_void_p, C.c_void_p, pp, C.c_void_p]
        self.lib = lib

    @staticmethod
    def rows(X: np.ndarray):
        """double** over the rows of a C-contiguous float64 matrix (pygmm.py:89-95 builds the same)."""
        X = np.ascontiguousarray(X, dtype=np.float64)
        n, d = X.shape
        base = X.ctypes.data
        arr = (C.POINTER(C.c_double) * n)()
        for i in range(n):
            arr[i] = C.cast(base + i * d * 8, C.POINTER(C.c_double))
        return arr, X

    def load(self, model_file: str):
        return self.lib.load(model_file.encode())

    def score_batch(self, handle, X: np.ndarray, concurrency: int = 1) -> np.ndarray:
        rows, keep = self.rows(X)
        out = np.empty(keep.shape[0], dtype=np.float64)
        self.lib.score_batch(handle, rows, _dp(out), keep.shape[0], keep.shape[1], concurrency)
        return out

    def score_all(self, handle, X: np.ndarray, concurrency: int = 1) -> float:
        rows, keep = self.rows(X)
        return float(self.lib.score_all(handle, rows, keep.shape[0], keep.shape[1], concurrency))
