"""Import shim: the product package lives in ``speaker-recognition_amd/`` (a directory name
Python cannot import directly); this module aliases it as ``speaker_recognition_amd``."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                      "speaker-recognition_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
