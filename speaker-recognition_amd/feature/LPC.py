"""Mirror of the reference's ``src/feature/LPC.py`` API (LPC-15 per frame: Hamming window and
pre-emphasis as MFCC.py, then talkbox's ``lpc(frame, n_lpc)[0][1:]``, NaN -> 0) on the device
kernel (csrc/lpc.hip).  Keyword defaults as LPC.py:59-63.  The arithmetic is third-party
(scikits.talkbox, absent here): restated from its published algorithm -- see oracle/lpc_oracle.py.
"""
import numpy as np

from ..core import MfccExtractor
from .utils import cached_func, diff_feature


@cached_func
def get_lpc_extractor(fs, win_length_ms=32, win_shift_ms=16, n_lpc=15, pre_emphasis_coef=0.95):
    # one device extractor yields [cepstra | LPC]; the LPC module hands back its columns only
    return MfccExtractor(fs, win_length_ms, win_shift_ms, pre_emphasis_coef=pre_emphasis_coef, n_lpc=n_lpc)


def extract(fs, signal=None, diff=False, **kwargs):
    """accept two argument, or one as a tuple"""
    if signal is None:
        assert type(fs) == tuple
        fs, signal = fs[0], fs[1]
    ex = get_lpc_extractor(fs, **kwargs)
    ret = ex.extract(np.asarray(signal), nd=0, cmvn=True)[:, ex.coefs:]
    if diff:
        return diff_feature(ret)
    return ret
