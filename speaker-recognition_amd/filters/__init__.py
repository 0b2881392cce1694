"""Voice-activity front end -- mirrors the reference's ``src/filters`` (``VAD``, ``LTSD_VAD``).
The reference delegates the arithmetic to third-party ``pyssp.vad.ltsd`` (absent from its tree):
the LTSD measure is restated from its published form and computed on the GPU (csrc/ltsd.hip);
parity is unpinned (no reference vectors exist).  The noise-reduction stage (``noisered.py``, a
``sox`` subprocess) is not part of this package."""
from .VAD import VAD
from .ltsd import LTSD_VAD

__all__ = ["VAD", "LTSD_VAD"]
