#!/usr/bin/env python3
"""profiles/traffic.json from the stdout of scripts/pmc.sh (profiles/r01_pmc.txt): HBM bytes per
launch of the scoring and MFCC kernels from the FETCH_SIZE / WRITE_SIZE passes, corrected as
MI355X_MICROARCH.md prescribes (counter unit KiB; gfx950 FETCH_SIZE reports half of the bytes)."""
import ast
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def parse(path):
    vals = {}
    for line in open(path):
        m = re.match(r"\s+(.*?) (\{.*\})\s*$", line)
        if not m:
            continue
        name, d = m.group(1), ast.literal_eval(m.group(2))
        for k, v in d.items():
            vals.setdefault(name, {})[k] = v
    return vals


src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r01_pmc.txt")
vals = parse(src)
# the scoring kernel alone on a static feature batch (scripts/time_score_engine.py under pmc.sh)
alone = parse(os.path.join(ROOT, "profiles", "r01_pmc_score_only.txt"))
score_alone = next(n for n in alone if "gmm_score" in n and "FETCH_SIZE" in alone[n])
score = next(n for n in vals if "gmm_score" in n and "FETCH_SIZE" in vals[n])
mfcc = next(n for n in vals if "mfcc_frames" in n and "FETCH_SIZE" in vals[n])
out = {
    "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (scripts/pmc.sh), per-dispatch averages; "
              "profiles/r01_pmc.txt; written by scripts/make_traffic.py",
    "units": "counter values are KiB; reads doubled per MI355X_MICROARCH.md (gfx950 FETCH_SIZE reports 1/2 of the bytes); "
             "calibration: the MFCC kernel reads 1000 x 160560 int16 samples = 321.1 MB algorithmic",
    "gmm_score_kernel": score_alone,
    "note": "per-kernel attribution inside the pipeline is polluted by write-backs: the 156 MB of features the "
            "CMVN kernel has just written are still dirty in L2/MALL when the scoring kernel starts, and their "
            "eviction is counted against it (in_pipeline_* below); the scoring kernel's own traffic is measured on a "
            "static feature batch (profiles/r01_pmc_score_only.txt) and is what gmm_score_hbm_bytes_per_launch reports",
    "gmm_score_fetch_kib_raw": alone[score_alone]["FETCH_SIZE"],
    "gmm_score_write_kib": alone[score_alone]["WRITE_SIZE"],
    "gmm_score_hbm_bytes_per_launch": (2 * alone[score_alone]["FETCH_SIZE"] + alone[score_alone]["WRITE_SIZE"]) * 1024,
    "in_pipeline_fetch_kib_raw": vals[score]["FETCH_SIZE"],
    "in_pipeline_write_kib": vals[score]["WRITE_SIZE"],
    "gmm_score_algorithmic_bytes_per_launch": 156000000,
    "mfcc_fetch_kib_raw": vals[mfcc]["FETCH_SIZE"],
    "mfcc_hbm_bytes_per_launch": (2 * vals[mfcc]["FETCH_SIZE"] + vals[mfcc]["WRITE_SIZE"]) * 1024,
}
json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
