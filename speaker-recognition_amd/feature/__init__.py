"""Mirror of the reference's ``src/feature/__init__.py``.

``mix_feature`` concatenates MFCC (the reference's own MFCC.py when bob is absent, :11-16) with
LPC-15 (:25-30): 13 + 15 = 28 dims at the defaults.  Both halves come out of ONE device pass here
(same frames, same window / pre-emphasis).  ``lpc=False`` returns the MFCC half only (what the
GMM path of BASELINE.json's configs uses); ``diff``/``nd`` append deltas to the MFCC half and are
exclusive with the LPC columns (the reference's mix_feature has no deltas either).
"""
import numpy as np

from . import LPC, MFCC
from ..core import MfccExtractor
from .utils import cached_func


def get_extractor(extract_func, **kwargs):
    def f(tup):
        return extract_func(*tup, **kwargs)
    return f


@cached_func
def _mix_extractor(fs, n_lpc, **kwargs):
    return MfccExtractor(fs, n_lpc=n_lpc, **kwargs)


def mix_feature(tup, lpc=True, diff=False, nd=1, n_lpc=15, **kwargs):
    if not lpc or diff:
        return MFCC.extract(tup, diff=diff, nd=nd, **kwargs)
    fs, signal = tup
    return _mix_extractor(fs, n_lpc, **kwargs).extract(np.asarray(signal), nd=0)
