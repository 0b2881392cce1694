"""speaker-recognition_amd -- MI355X-native MFCC + diagonal-GMM scoring hot path.

The compute lives in ``lib/pygmm.so`` (HIP kernels for gfx950 behind the C ABI declared in
``include/pygmm_hip.h``); the modules here mirror the reference's Python surface for that
path: ``pygmm`` (src/gmm/python/pygmm.py), ``gmmset`` (src/testbench/gmmset.py),
``feature`` (src/feature/MFCC.py, utils.py, __init__.py), ``interface``
(src/gui/interface.py) and ``cli`` (src/speaker-recognition.py).
"""
__version__ = "0.1.0"
