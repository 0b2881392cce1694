"""``VAD`` -- same surface as the reference's ``src/filters/VAD.py`` (``init_noise``, ``filter``):
the LTSD detector only (the reference's noise-reduction and energy-silence stages are commented
out there too, VAD.py:24-33)."""
from .ltsd import LTSD_VAD


class VAD(object):
    def __init__(self):
        self.initted = False
        self.ltsd = LTSD_VAD()

    def init_noise(self, fs, signal):
        self.initted = True
        self.ltsd.init_params_by_noise(fs, signal)

    def filter(self, fs, signal):
        if not self.initted:
            raise RuntimeError("NoiseFilter Not Initialized")      # VAD.py:29 raises a string
        return self.ltsd.filter(signal)
