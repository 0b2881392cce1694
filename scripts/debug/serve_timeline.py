#!/usr/bin/env python3
"""Merged timeline (HIP API calls on the host, kernels and copies on the device) of the LAST calls of a rocprofv3 run of
serving_one.py:   rocprofv3 --kernel-trace --memory-copy-trace --hip-runtime-trace --output-format csv -d DIR -- python serving_one.py
                  python serve_timeline.py DIR [events=60]"""
import csv, glob, sys
d = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
ev = []
def rd(pat):
    fs = glob.glob(d + "/**/*" + pat, recursive=True)
    return list(csv.DictReader(open(fs[0]))) if fs else []
for r in rd("kernel_trace.csv"):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "GPU ", r["Kernel_Name"][:60]))
for r in rd("memory_copy_trace.csv"):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY", r.get("Direction", "") + " " + r.get("Bytes", "")))
for r in rd("hip_api_trace.csv"):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "host", r["Function"]))
ev.sort()
ev = ev[-n:]
t0 = ev[0][0]
for s, e, k, name in ev:
    print("%9.1f .. %9.1f  (%7.1f us)  %s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, k, name))
