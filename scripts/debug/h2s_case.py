import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import gmm_oracle as go
from speaker_recognition_amd import _lib, synth
from speaker_recognition_amd.core import Batch, ModelSet
from speaker_recognition_amd.pygmm import GMM
K, D, S = (int(v) for v in sys.argv[1:4])
frac = float(sys.argv[4]) if len(sys.argv) > 4 else 0.01
ubm = synth.synth_gmm(K, D, 1234 + K)
models = [ubm] + [synth.synth_map_speaker(ubm, 7000 + s) for s in range(S)]
lens = [0, 1, 127, 128, 129, 300, 33]
utts = [synth.draw_frames(models[1 + u % S], n, 11 + u, outlier_frac=frac) for u, n in enumerate(lens)]
X = np.concatenate(utts).astype(np.float64)
want = np.stack([go.score_batch(go.GMMParams(*m), X) for m in models])
ms = ModelSet([GMM.from_arrays(*m) for m in models])
for G in (1, 2):
    for force in (0, 1):
        _lib.set_option("score_engine", 6); _lib.set_option("score_model_groups", G); _lib.set_option("score_h2s_force_exc", force)
        sums, arg, fll = ms.score(Batch.from_features(utts), frame_ll=True)
        rel = np.abs(fll - want) / np.maximum(1, np.abs(want))
        bad = np.argwhere(rel > 1e-4)
        print("G", G, "force", force, "max rel", rel.max(), "n bad", len(bad), "models", sorted(set(bad[:, 0]))[:20], "frames", sorted(set(bad[:, 1]))[:40])
print("---- determinism / pattern")
_lib.set_option("score_engine", 6); _lib.set_option("score_model_groups", 1); _lib.set_option("score_h2s_force_exc", 0)
runs = [ms.score(Batch.from_features(utts), frame_ll=True)[2] for _ in range(3)]
print("run0==run1", np.array_equal(runs[0], runs[1]), "run1==run2", np.array_equal(runs[1], runs[2]))
rel = np.abs(runs[0] - want) / np.maximum(1, np.abs(want))
bad = np.argwhere(rel > 1e-4)
for m_, f_ in bad[:12]:
    print("model", m_, "frame", f_, "got", runs[0][m_, f_], "want", want[m_, f_], "ubm", want[0, f_])
# single block sets: S = 14 (one block), same shapes
ms1 = ModelSet([GMM.from_arrays(*m) for m in models[:15]])
f1 = ms1.score(Batch.from_features(utts), frame_ll=True)[2]
print("one-block set max rel", (np.abs(f1 - want[:15]) / np.maximum(1, np.abs(want[:15]))).max())
