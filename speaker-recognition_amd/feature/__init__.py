"""Mirror of the reference's ``src/feature/__init__.py``.

``mix_feature`` there concatenates MFCC with LPC-15 (:25-30).  LPC (a per-frame Levinson
recursion from the absent scikits.talkbox) is outside this path's scope, so
``mix_feature`` returns the MFCC half only -- stated here rather than silently substituted.
"""
from . import MFCC


def get_extractor(extract_func, **kwargs):
    def f(tup):
        return extract_func(*tup, **kwargs)
    return f


def mix_feature(tup, **kwargs):
    return MFCC.extract(tup, **kwargs)
