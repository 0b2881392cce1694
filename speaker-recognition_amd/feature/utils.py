"""API mirror of the reference's ``src/feature/utils.py`` (``cached_func`` :11-21, ``diff_feature``
:24-31) -- same names and results, own implementation.  ``diff_feature`` works on an already
extracted feature matrix on the host; the extraction path (``MFCC.extract(..., diff=True)``)
computes its deltas on the device, fused with the CMVN write-out (csrc/mfcc.hip)."""
import functools

import numpy as np


def cached_func(function):
    """Memoise on the call signature (positional values + keyword items), as the reference does to
    build one extractor per parameter tuple."""
    memo = {}

    @functools.wraps(function)
    def lookup(*args, **kwargs):
        signature = (args, tuple(sorted(kwargs.items())))
        try:
            return memo[signature]
        except KeyError:
            value = memo[signature] = function(*args, **kwargs)
            return value
    return lookup


def diff_feature(feat, nd=1):
    """Rows t >= nd of ``feat`` with their causal finite differences of order 1..nd appended:
    nd=1 -> [c_t, c_t - c_{t-1}]; nd=2 -> [c_t, c_t - c_{t-1}, c_t - 2 c_{t-1} + c_{t-2}]."""
    feat = np.asarray(feat)
    if nd not in (1, 2):
        return None                      # the reference falls off the end of its if/elif
    blocks = [feat[nd:]]
    level = feat
    for order in range(1, nd + 1):
        level = np.diff(level, axis=0)   # order-th difference, one row shorter each time
        blocks.append(level[nd - order:])
    return np.concatenate(blocks, axis=1)
