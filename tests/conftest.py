import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def gmm_golden():
    return np.load(os.path.join(ROOT, "tests", "golden", "gmm_golden.npz"))


@pytest.fixture(scope="session")
def clamp_golden():
    """Reference-DSO answers around its underflow clamp (tests/golden/make_clamp_golden.py)."""
    return np.load(os.path.join(ROOT, "tests", "golden", "clamp_golden.npz"))


@pytest.fixture(scope="session")
def flush_golden():
    """Reference-DSO answers where its flush-to-zero arithmetic decides on PARTIAL products
    (tests/golden/make_flush_golden.py)."""
    return np.load(os.path.join(ROOT, "tests", "golden", "flush_golden.npz"))


def flush_models(g, c):
    """the case's model(s) as (weights, mean, sigma): the UBM first, then the speakers that share its sigma / weights"""
    out = [(g[c + "_w"], g[c + "_mean"], g[c + "_sigma"])]
    if c + "_spk_mean" in g.files:
        out += [(g[c + "_w"], m, g[c + "_sigma"]) for m in g[c + "_spk_mean"]]
    return out


@pytest.fixture(scope="session")
def mfcc_golden():
    return np.load(os.path.join(ROOT, "tests", "golden", "mfcc_golden.npz"))


@pytest.fixture(scope="session")
def built_lib():
    """The C-ABI library, built in-tree (hipcc cross-compiles without a GPU)."""
    import __graft_entry__ as ge
    from speaker_recognition_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        ge.build()
    return _lib.lib()


@pytest.fixture(scope="session")
def oracle_built():
    from oracle import gmm_oracle
    if not os.path.exists(gmm_oracle.ORACLE_SO):
        gmm_oracle.build(ref=False)
    return gmm_oracle


def ll_close(a, ref, tol=1e-4):
    """|a - ref| <= tol * max(1, |ref|) elementwise (SURVEY.md 8d parity gate); returns worst ratio."""
    a = np.asarray(a, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    return float(np.max(np.abs(a - ref) / np.maximum(1.0, np.abs(ref)))) if a.size else 0.0
