import sys, time, os, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from speaker_recognition_amd import _lib
from speaker_recognition_amd.core import Batch, MfccExtractor, ModelSet
from speaker_recognition_amd.pygmm import GMM
clips, models = bench.build_workload(0, 1000, 1000)
pcm = Batch.from_pcm(clips); ex = MfccExtractor(16000, **bench.MFCC_KW)
ms = ModelSet([GMM.from_arrays(*m) for m in models])
def run(n, tag):
    ts=[]
    for i in range(n):
        t0=time.perf_counter(); s,a = ex.predict_batch(ms, pcm, nd=2); ts.append((time.perf_counter()-t0)*1e3)
    print(tag, np.round(ts,2), flush=True)
run(6, "no profiling, first 6 steps of the process:")
_lib.profile_enable(True)
run(6, "profiling on:")
_lib.profile_reset()
run(6, "after profile_reset:")
_lib.profile_enable(False)
run(4, "profiling off again:")
