#!/usr/bin/env python3
"""ONE call of train_model (pygmm.hh:33) through the legacy C ABI in a FRESH process -- of the reference's compiled
library (--lib ref: oracle/_ref/pygmm_ref.so) or of this repo's (--lib hip: speaker-recognition_amd/lib/pygmm.so).
The reference draws its initialisation from libc rand() (random.hh:22-25, gmm.hh:44, kmeansII.cc:94,133), which
starts from its default seed in a fresh process; so does the HIP library (csrc/kmeans_init.hip).  An optional model
file is load()ed first: every Gaussian the reference constructs consumes one rand().

usage: _train_proc.py --lib ref|hip X.npy K nr_iteration init_with_kmeans concurrency out.model [preload.model]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import gmm_oracle as go  # noqa: E402
from speaker_recognition_amd._lib import LIB_PATH, Parameter  # noqa: E402

assert sys.argv[1] == "--lib"
which = sys.argv[2]
X = np.load(sys.argv[3])
K, iters, km, conc, out = int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7]), sys.argv[8]
ref = go.RefLib() if which == "ref" else go.RefLib(LIB_PATH)     # the same ten symbols either way
if len(sys.argv) > 9:
    assert ref.lib.load(sys.argv[9].encode())
h = ref.lib.new_gmm(K, 1)
p = Parameter(nr_instance=len(X), nr_dim=X.shape[1], nr_mixture=K, min_covar=1e-3, threshold=0.01,
              nr_iteration=iters, init_with_kmeans=km, concurrency=conc, verbosity=0)
rows, keep = ref.rows(X)
os.chdir(os.path.dirname(os.path.abspath(out)))     # the reference's trainer drops gmm-training-intermediate-dump.model in cwd
ref.lib.train_model(h, rows, C.byref(p))
ref.lib.dump(h, out.encode())
