#!/usr/bin/env python3
"""Per-frame log-likelihood error of each scoring engine against the float64 oracle (exact
log-sum-exp, no clamp) on frames drawn from BASELINE configs[1]-shaped models: max and RMS of
|LL - LL64| / max(1, |LL64|).  Evidence for "the split-bf16 engine is fp32-grade" (HISTORY.md 2.1)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import gmm_oracle as go  # noqa: E402
from speaker_recognition_amd import _lib, synth  # noqa: E402
from speaker_recognition_amd.core import Batch, ModelSet  # noqa: E402
from speaker_recognition_amd.pygmm import GMM  # noqa: E402

S, K, D, N = 16, 64, 39, 20000
models = [synth.synth_gmm(K, D, 7 + s) for s in range(S)]
utts = [synth.draw_frames(models[u % S], N // 8, 42 + u) for u in range(8)]
X = np.concatenate(utts).astype(np.float64)
want = np.stack([go.score_batch(go.GMMParams(*m), X, go.MODE_LOGSUMEXP, clamp_compat=False) for m in models])
ms = ModelSet([GMM.from_arrays(*m) for m in models])
out = {"frames": int(X.shape[0]), "models": S, "mixtures": K, "dim": D}
for name, eng in (("vector_alu_direct_form", 1), ("fp32_mfma_expanded_form", 2), ("split_bf16_mfma_expanded_form", 3),
                  ("split_fp16_mfma_expanded_form", 5)):
    _lib.set_option("score_engine", eng)
    sums, arg, fll = ms.score(Batch.from_features(utts), frame_ll=True)
    rel = np.abs(fll - want) / np.maximum(1.0, np.abs(want))
    out[name] = {"max_rel": float(rel.max()), "rms_rel": float(np.sqrt(np.mean(rel ** 2))), "kernel": _lib.last_score_kernel()}
print(json.dumps(out, indent=1))
