import csv, sys, glob
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last call = the last group of kernels ending with gmm_finalize / flush...; print the last 14 kernels with gaps
last = rows[-14:]
t0 = int(last[0]["Start_Timestamp"])
prev_end = None
for r in last:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print("%8.1f us  +%6.1f gap  dur %7.1f us  %s" % ((s - t0) / 1e3, gap, (e - s) / 1e3, r["Kernel_Name"][:70]))
    prev_end = e
