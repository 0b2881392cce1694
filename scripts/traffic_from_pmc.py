#!/usr/bin/env python3
"""profiles/rNN_traffic.json from the stdout of scripts/pmc.sh (profiles/rNN_pmc.txt): HBM-side bytes per launch of the headline's
scoring and MFCC kernels from the FETCH_SIZE / WRITE_SIZE passes (each counter in its own run), corrected as MI355X_MICROARCH.md
prescribes -- counter unit KiB; gfx950's FETCH_SIZE reports half of the bytes -- beside the algorithmic bytes and the figure the bench
line carries:  traffic_from_pmc.py PMC_TXT BENCH_LINE_JSON OUT_JSON"""
import ast
import json
import re
import sys

pmc, line, out = sys.argv[1:4]
vals = {}
for l in open(pmc):
    m = re.match(r"\s+(.*?) (\{.*\})\s*$", l)
    if m:
        for k, v in ast.literal_eval(m.group(2)).items():
            vals.setdefault(m.group(1), {})[k] = v
rec = json.loads(open(line).read().strip().splitlines()[-1])
frames = rec["config"]["frames_per_gpu"]
res = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE, one counter per run (scripts/pmc.sh, %s), per-dispatch averages of "
                 "`bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-config-blocks --no-traffic`" % pmc,
       "units": "counter values are KiB; reads doubled per MI355X_MICROARCH.md (gfx950 FETCH_SIZE reports 1/2 of the bytes)"}
for key, pat in (("gmm_score_h2p_kernel", "gmm_score_h2p_kernel"), ("mfcc_frames_fft2048_f64_kernel", "mfcc_frames_fft2048_f64_kernel<short, 1>")):
    name = next(n for n in vals if pat in n and "FETCH_SIZE" in vals[n] and "WRITE_SIZE" in vals[n])
    f, w = vals[name]["FETCH_SIZE"], vals[name]["WRITE_SIZE"]
    res[key] = {"kernel": name, "FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w, "hbm_bytes_per_launch": (2 * f + w) * 1024}
res["algorithmic_bytes"] = {"gmm_score_h2p_kernel": frames * 4.0 * rec["config"]["dim"],
                            "mfcc_frames_fft2048_f64_kernel": frames * 372.0 * 1.002}
res["bench_line_traffic"] = rec["roofline"]["traffic"]
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
