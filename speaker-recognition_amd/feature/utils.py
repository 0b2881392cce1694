"""Mirror of the reference's ``src/feature/utils.py``: ``cached_func`` (:11-21) and
``diff_feature`` (:24-31).  ``diff_feature`` is pure array slicing on an already extracted
feature matrix; the extraction path itself (``MFCC.extract(..., diff=True)``) computes its
deltas on the device, fused with the CMVN write-out."""
import numpy

kwd_mark = object()


def cached_func(function):
    cache = {}

    def wrapper(*args, **kwargs):
        key = args + (kwd_mark,) + tuple(sorted(kwargs.items()))
        if key not in cache:
            cache[key] = function(*args, **kwargs)
        return cache[key]
    return wrapper


def diff_feature(feat, nd=1):
    diff = feat[1:] - feat[:-1]
    feat = feat[1:]
    if nd == 1:
        return numpy.concatenate((feat, diff), axis=1)
    elif nd == 2:
        d2 = diff[1:] - diff[:-1]
        return numpy.concatenate((feat[1:], diff[1:], d2), axis=1)
