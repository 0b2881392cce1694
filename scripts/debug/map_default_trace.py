"""MAP enrolment as the drop-in surface runs it (GMMSet(ubm=...).fit_new: nr_iteration 200, threshold 0.01): wall time per speaker and the
phase times of the last iterations for a K-mixture UBM in 39 dims on 3000 frames: map_default_trace.py [K]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from speaker_recognition_amd import synth
from speaker_recognition_amd.pygmm import GMM
K = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ubm_raw = synth.synth_gmm(K, 39, 99)
ubm = GMM.from_arrays(*ubm_raw)
spk = [synth.draw_frames(synth.synth_map_speaker(ubm_raw, 500 + s), 3000, 10 + s) for s in range(4)]
GMM(K).fit(spk[0], ubm=ubm)
for s in (1, 2):
    m = GMM(K)
    t0 = time.perf_counter(); it = m.fit(spk[s], ubm=ubm); dt = time.perf_counter() - t0
    print("MAP K=%d 3000 frames, defaults: %d iterations, %.1f ms (%.3f ms per iteration)" % (K, it, dt * 1e3, dt * 1e3 / it))
m = GMM(K, verbosity=2)
m.fit(spk[3], ubm=ubm)
