"""Mirror of the reference's ``src/feature/MFCC.py`` API on the device extractor.

``get_mfcc_extractor`` keeps the reference's keyword defaults (:115-121: 32 ms / 16 ms /
FFT 2048 / 50 filters / 13 ceps / pre-emphasis 0.95) and is memoised the same way;
``extract(fs, signal=None, diff=False, **kwargs)`` keeps the tuple form (:123-132).
"""
import numpy as np

from ..core import MfccExtractor as MFCCExtractor  # noqa: F401  (class name kept from MFCC.py:18)
from .utils import cached_func

POWER_SPECTRUM_FLOOR = 1e-100


@cached_func
def get_mfcc_extractor(fs, win_length_ms=32, win_shift_ms=16, FFT_SIZE=2048, n_filters=50,
                       n_ceps=13, pre_emphasis_coef=0.95):
    return MFCCExtractor(fs, win_length_ms, win_shift_ms, FFT_SIZE, n_filters, n_ceps,
                         pre_emphasis_coef)


def extract(fs, signal=None, diff=False, nd=1, **kwargs):
    """accept two argument, or one as a tuple.  ``nd`` (1 or 2: delta order when diff=True) is
    the one extension; the reference's ``diff=True`` is nd=1 (utils.py:24)."""
    if signal is None:
        assert type(fs) == tuple
        fs, signal = fs[0], fs[1]
    signal = np.asarray(signal)
    return get_mfcc_extractor(fs, **kwargs).extract(signal, nd=nd if diff else 0)
