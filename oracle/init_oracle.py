"""CPU restatement (numpy, float64) of the reference's GMM initialisation in front of EM -- TEST INFRASTRUCTURE: only
tests/ may import this; the product path is speaker-recognition_amd/csrc/kmeans_init.hip.

Follows, decision for decision:
  * GMMTrainerBaseline::init_gaussians      /root/reference/src/gmm/src/gmm.cc:306-361
  * KMeansIISolver::cluster (k-means||)     src/gmm/src/kmeansII.cc:82-171
  * KMeansppSolver::cluster_weighted        src/gmm/src/kmeans++.cc:161-212 (+ Vector2Instance :43-51)
  * KMeansSolver::Lloyds_iteration[_weighted]  src/gmm/src/kmeans.cc:150-246, :249-342
  * Random                                   src/gmm/src/random.hh:17-56
with the reference's random numbers: libc rand() (glibc's TYPE_3 generator from its default seed, restated here so
that nothing else in the process can disturb it) and libstdc++'s std::default_random_engine (minstd_rand0) +
std::uniform_real_distribution<double> (generate_canonical, two draws per double) seeded from it.

Pinned by tests/test_oracle_golden.py::test_init_oracle_matches_reference_trainer against models the reference's
compiled trainer produced FROM SCRATCH in fresh processes (tests/golden/make_init_golden.py), and the glibc restatement
against the C library's own rand()."""
from __future__ import annotations

import numpy as np

RAND_MAX = 2147483647
INT_MAX = 2147483647
DBL_MAX = np.finfo(np.float64).max


class GlibcRand:
    """glibc rand() / random_r.c TYPE_3: r[i] = r[i-3] + r[i-31] over 31 words filled by 16807 x mod 2^31 - 1, 310 outputs dropped."""

    def __init__(self, seed: int = 1):
        seed = seed or 1
        st = [0] * 31
        st[0] = seed
        for i in range(1, 31):
            hi, lo = divmod(st[i - 1], 127773)          # (values stay positive: C's truncating / and % agree with divmod)
            word = 16807 * lo - 2836 * hi
            if word < 0:
                word += 2147483647
            st[i] = word
        self.st, self.f, self.b = st, 3, 0
        for _ in range(310):
            self()

    def __call__(self) -> int:
        val = (self.st[self.f] + self.st[self.b]) & 0xFFFFFFFF
        self.st[self.f] = val
        self.f = (self.f + 1) % 31
        self.b = (self.b + 1) % 31
        return val >> 1


class RefRandom:
    """random.hh:17-56 on libstdc++: minstd_rand0 (x <- 16807 x mod 2^31 - 1) and generate_canonical<double, 53>."""
    M = 2147483647

    def __init__(self, seed: int):
        x = seed % self.M
        self.x = x if x else 1

    def _next(self) -> int:
        self.x = (16807 * self.x) % self.M
        return self.x

    def rand_real(self) -> float:
        r = 2147483646.0                                 # max - min + 1
        s = (self._next() - 1) + (self._next() - 1) * r  # k = 2 draws cover 53 bits
        v = s / (r * r)
        return v if v < 1.0 else np.nextafter(1.0, 0.0)

    def rand_int(self, max_val: int = INT_MAX) -> int:
        return int(self.rand_real() * max_val)


def _blocked_sum(v: np.ndarray, concurrency: int) -> float:
    """Sequential inside worker blocks of ceil(n / concurrency), then over the blocks (kmeansII.cc:103-129)."""
    n = len(v)
    block = int(np.ceil(n / concurrency))
    total = 0.0
    for b in range(0, n, block):
        s = 0.0
        for x in v[b:b + block].tolist():
            s += x
        total += s
    return total


def _dist_rows(X: np.ndarray, c: np.ndarray) -> np.ndarray:
    """Squared distances of every row to one centre, summed dimension by dimension as the reference does."""
    d = np.zeros(len(X))
    for j in range(X.shape[1]):
        delta = X[:, j] - c[j]
        d += delta * delta
    return d


def _lloyd(X, centroids, concurrency, weight=None, max_iter=200):
    """kmeans.cc:150-246 (weight None) / :249-342.  Empty clusters divide by zero as the reference does."""
    n, dim = X.shape
    K = len(centroids)
    block = int(np.ceil(n / concurrency))
    best, last, best_c = DBL_MAX, DBL_MAX, None
    keep = np.abs(X) >= 1e-15 if weight is not None else None      # Vector2Instance drops |x| < 1e-15 (candidates only)
    for _ in range(max_iter):
        D = np.empty((n, K))
        for k in range(K):
            if weight is None:
                D[:, k] = _dist_rows(X, centroids[k])
            else:
                d = np.zeros(n)
                for j in range(dim):
                    delta = np.where(keep[:, j], X[:, j] - centroids[k, j], 0.0)
                    d += delta * delta
                D[:, k] = d * weight
        with np.errstate(invalid="ignore"):
            cmp = np.where(np.isnan(D), np.inf, D)
        belong = np.argmin(cmp, axis=1)                            # first minimum wins (strict <)
        mind = cmp[np.arange(n), belong]
        none = ~np.isfinite(mind)
        mind = np.where(none, DBL_MAX, mind)
        total = 0.0
        sums = np.zeros((K, dim))
        size = np.zeros(K)
        for b in range(0, n, block):
            e = min(n, b + block)
            bs = np.zeros((K, dim))
            bc = np.zeros(K)
            ok = ~none[b:e]
            idx = belong[b:e][ok]
            if weight is None:
                np.add.at(bc, idx, 1.0)                            # ufunc.at: unbuffered, in index order = the reference's point order
                np.add.at(bs, idx, X[b:e][ok])
            else:
                np.add.at(bc, idx, weight[b:e][ok])
                np.add.at(bs, idx, np.where(keep[b:e][ok], X[b:e][ok] * weight[b:e][ok, None], 0.0))
            s = 0.0
            for v in mind[b:e].tolist():
                s += v
            total += s
            sums += bs
            size += bc
        if total < best:
            best, best_c = total, centroids.copy()
        if abs(last - total) < 1e-6:
            break
        if weight is None and total > best * 1.5:
            break
        if weight is not None:
            size = np.trunc(size)                                  # an int accumulator, kmeans.cc:303-308
        with np.errstate(invalid="ignore", divide="ignore"):
            centroids = sums / size[:, None]
        last = total
    return best_c


def kmeans_parallel(X: np.ndarray, K: int, concurrency: int, rand: GlibcRand) -> np.ndarray:
    """KMeansIISolver::cluster with its defaults (oversampling_factor = size_factor = 2)."""
    n, dim = X.shape
    solver_random = RefRandom(rand())
    cand = [X[rand() % n].copy()]
    dist = np.full(n, DBL_MAX)
    belong = np.zeros(n, dtype=np.int64)
    last_size = 0
    while True:
        for j in range(last_size, len(cand)):
            d = _dist_rows(X, cand[j])
            closer = d < dist
            dist[closer] = d[closer]
            belong[closer] = j
        if len(cand) > 2.0 * K:
            break
        total = _blocked_sum(dist, concurrency)
        last_size = len(cand)
        for i in range(n):
            if rand() / RAND_MAX * total < dist[i] * 2.0 * K:
                cand.append(X[i].copy())
        if len(cand) == last_size:
            break
    while len(cand) <= 2.0 * K:
        cand.append(X[solver_random.rand_int(n)].copy())
    P = np.array(cand)
    weight = np.bincount(belong, minlength=len(P)).astype(np.float64)
    pp = RefRandom(rand())
    Ps = np.where(np.abs(P) < 1e-15, 0.0, P)
    cent = np.zeros((K, dim))
    cent[0] = Ps[pp.rand_int() % len(P)]
    keep = np.abs(P) >= 1e-15
    pd = np.full(len(P), DBL_MAX)
    for k in range(1, K):
        d = np.zeros(len(P))
        for j in range(dim):
            delta = np.where(keep[:, j], P[:, j] - cent[k - 1, j], 0.0)
            d += delta * delta
        pd = np.minimum(pd, d * weight)
        total = _blocked_sum(pd, concurrency)
        rw = pp.rand_int() / RAND_MAX * total
        for i in range(len(P)):
            rw -= pd[i]
            if rw <= 0:
                cent[k] = Ps[i]
                break
    cent = _lloyd(P, cent, concurrency, weight=weight)
    return _lloyd(X, cent, concurrency)


def init_gaussians(X: np.ndarray, K: int, init_with_kmeans: int, concurrency: int, rand: GlibcRand):
    """-> (weights, mean, sigma) as GMMTrainerBaseline::init_gaussians leaves them.  `rand` is the process's rand() stream;
    the trainer's own Random is seeded first (pygmm.cc:65), then one draw per `new Gaussian` (gmm.hh:44, gmm.cc:327-329)."""
    X = np.ascontiguousarray(X, dtype=np.float64)
    n, dim = X.shape
    trainer_random = RefRandom(rand())
    mean = np.zeros(dim)
    for x in X:
        mean += x
    mean /= n
    var = np.zeros(dim)
    for x in X:
        var += (x - mean) ** 2
    sigma = np.sqrt(var * (1.0 / (n - 1)))
    for _ in range(K):
        rand()
    if init_with_kmeans:
        mu = kmeans_parallel(X, K, concurrency, rand)
    else:
        mu = np.array([X[trainer_random.rand_int(n)] for _ in range(K)])
    return np.full(K, 1.0 / K), mu, np.tile(sigma, (K, 1))
