#!/usr/bin/env python3
"""A/B of two BUILDS of the library on the configs[2]-shaped scoring pass, each in its own subprocess, alternating:
   ab_two_libs.py LIB_A LIB_B [UTTS] [ROUNDS]      (prints the scoring kernel's HIP-event ms per pass)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np
import bench
from speaker_recognition_amd import _lib, synth
from speaker_recognition_amd.core import Batch, MfccExtractor, ModelSet
from speaker_recognition_amd.pygmm import GMM
utts = int(sys.argv[1])
ex = MfccExtractor(bench.FS, **bench.MFCC_KW)
n_samples = (bench.FRAMES_PER_UTT + bench.ND - 1) * ex.FRAME_SHIFT + ex.FRAME_LEN
base = bench.base_clips(bench.CFG2_SPEAKERS, n_samples)
ubm = synth.synth_gmm(bench.CFG2_MIX, bench.DIM, 99)
ms = ModelSet([GMM.from_arrays(*m) for m in [ubm] + [synth.synth_map_speaker(ubm, 500 + s) for s in range(bench.CFG2_SPEAKERS)]])
cat, off = bench.make_pcm(base, utts, 0)
feats = ex.extract_batch(Batch.from_pcm((cat, off)), nd=bench.ND)
_lib.profile_enable(True)
ts = []
for r in range(5):
    _lib.profile_reset(); sums, arg = ms.score(feats); t, c = _lib.profile_get(_lib.T_SCORE)
    if r: ts.append(t)
print("%%.2f %%.2f %%.2f %%.2f ms  checksum %%.6f  %%s" %% (*ts, float(np.sum(sums)), _lib.last_score_kernel()[:40]))
''' % ROOT
a, b = sys.argv[1], sys.argv[2]
utts = sys.argv[3] if len(sys.argv) > 3 else "3000"
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 2
for r in range(rounds):
    for lib in (a, b):
        env = dict(os.environ, SR_PYGMM_LIB=os.path.abspath(lib))
        out = subprocess.run([sys.executable, "-c", CHILD, utts], env=env, capture_output=True, text=True, stdin=subprocess.DEVNULL, timeout=300)
        print(os.path.basename(lib), (out.stdout.strip().split("\n") or [""])[-1], out.stderr.strip()[-200:] if out.returncode else "", flush=True)
