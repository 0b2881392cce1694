#!/usr/bin/env python3
"""Host-side API calls that can stall a pipeline (allocations, frees, synchronisations, registrations) from a rocprofv3
--hip-runtime-trace run, with the H2D copies > 1 MB beside them for orientation:  api_stalls.py DIR [min_us=50]"""
import csv, glob, sys
d = sys.argv[1]
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 50.0
ev = []
for f in glob.glob(d + "/**/*hip_api_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Function"]
        if any(k in n for k in ("Malloc", "Free", "Synchronize", "Register", "hipMemcpy ", "hipMemset ")) or n in ("hipMemcpy", "hipMemset"):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 300000:
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "    [copy " + r.get("Direction", "") + "]"))
ev.sort()
t0 = ev[0][0]
for s, e, n in ev:
    if (e - s) / 1e3 >= min_us:
        print("%10.3f ms  %9.1f us  %s" % ((s - t0) / 1e6, (e - s) / 1e3, n))
