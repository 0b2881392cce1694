#!/usr/bin/env python3
"""configs[1]'s resident-PCM step, a few times, for `rocprofv3 --kernel-trace`: where the 0.4 ms between the kernels' sum and the
step's wall time go (scripts/debug/kernel_trace_tail.py on the trace).  Prints the wall time per step."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from speaker_recognition_amd import _lib, synth  # noqa: E402
from speaker_recognition_amd.core import Batch, MfccExtractor, ModelSet  # noqa: E402
from speaker_recognition_amd.pygmm import GMM  # noqa: E402

ex = MfccExtractor(bench.FS, **bench.MFCC_KW)
n_samples = (bench.FRAMES_PER_UTT + bench.ND - 1) * ex.FRAME_SHIFT + ex.FRAME_LEN
base = bench.base_clips(bench.CFG1_MODELS, n_samples)
ms = ModelSet([GMM.from_arrays(*synth.synth_gmm(bench.CFG1_MIX, bench.DIM, bench.MODEL_SEED + s)) for s in range(bench.CFG1_MODELS)])
cat, off = bench.make_pcm(base, bench.CFG1_UTTS, 0)
pcm = Batch.from_pcm((cat, off))
for _ in range(3):
    ex.predict_batch(ms, pcm, nd=bench.ND)
_lib.synchronize()
t0 = time.perf_counter()
for _ in range(6):
    ex.predict_batch(ms, pcm, nd=bench.ND)
_lib.synchronize()
print("configs[1] step: %.3f ms wall" % ((time.perf_counter() - t0) / 6 * 1e3))
