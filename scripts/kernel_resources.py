#!/usr/bin/env python3
"""profiles/rNN_kernel_resources.txt from the remarks the build keeps next to every object (speaker-recognition_amd/build/*.resources,
csrc/Makefile: -Rpass-analysis=kernel-resource-usage): registers, scratch, occupancy and LDS of every kernel.  kernel_resources.py OUT [ROUND]"""
import glob
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = open(sys.argv[1], "w")
out.write("# registers / scratch / occupancy of every kernel of the round-%s build (from speaker-recognition_amd/build/*.resources)\n" % (sys.argv[2] if len(sys.argv) > 2 else "6"))
for path in sorted(glob.glob(os.path.join(ROOT, "speaker-recognition_amd", "build", "*.resources"))):
    out.write("\n## %s\n" % os.path.basename(path))
    cur, rows = None, []
    for line in open(path):
        m = re.search(r" Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1)}
            rows.append(cur)
            continue
        if cur is None:
            continue
        for key, pat in (("vgpr", r"\bVGPRs: (\d+)"), ("agpr", r"\bAGPRs: (\d+)"), ("sgpr", r"TotalSGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                         ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
            m = re.search(pat, line)
            if m and key not in cur:
                cur[key] = int(m.group(1))
    seen = set()
    for r in rows:
        if r["name"] in seen:
            continue
        seen.add(r["name"])
        out.write("%-110s VGPR %3d AGPR %3d SGPR %3d scratch %4d occ %d LDS %d\n" % (r["name"], r.get("vgpr", -1), r.get("agpr", -1), r.get("sgpr", -1),
                                                                                    r.get("scratch", -1), r.get("occ", -1), r.get("lds", -1)))
out.close()
