import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from speaker_recognition_amd import _lib
from speaker_recognition_amd.core import Batch, MfccExtractor, ModelSet
ex = MfccExtractor(bench.FS, **bench.MFCC_KW)
n_samples = (1000 + 1) * ex.FRAME_SHIFT + ex.FRAME_LEN
base = bench.base_clips(40, n_samples)
for K in (16, 64, 512):
    ubm, spk = bench.train_cfg2_models(ex, base, K)
    ms = ModelSet([ubm] + spk)
    w, mu, sg = ubm.params()
    print("K", K, ms.info(), "sigma min/max", sg.min(), sg.max(), "mean abs mu", np.abs(mu).mean())
    a = np.sum(((mu - mu.mean(axis=0)) / sg) ** 2, axis=1)
    print("   amp percentiles", np.percentile(a, [50, 90, 99, 100]))
    fb = ex.extract_batch(Batch.from_pcm(base[:4]), nd=2)
    sums, arg = ms.score(fb)
    print("   kernel:", _lib.last_score_kernel(), "per-frame LL", sums[:, 0] / 1000)
