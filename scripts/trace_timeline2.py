#!/usr/bin/env python3
"""Merged kernel + memory-copy timeline (rocprofv3 --kernel-trace --memory-copy-trace CSVs) around the last K1 launches.
usage: trace_timeline2.py <dir> [first_k1_from_end] [last_k1_from_end] [min_us]"""
import csv, glob, os, sys
d = sys.argv[1]
a = int(sys.argv[2]) if len(sys.argv) > 2 else 8
b = int(sys.argv[3]) if len(sys.argv) > 3 else 4
mn = float(sys.argv[4]) if len(sys.argv) > 4 else 20.0
def short(n):
    n = n.replace("void ", "").replace("thip::", "")
    return n.split("(")[0].split("<")[0]
ev = []
for f in glob.glob(os.path.join(d, "*kernel_trace.csv")):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r["Stream_Id"]))
for f in glob.glob(os.path.join(d, "*memory_copy_trace.csv")):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY_" + r["Direction"][12:], r.get("Stream_Id", "")))
ev.sort()
k1 = [e for e in ev if e[2] == "tim_graph_mfma_kernel"]
print("K1 launches", len(k1), "start-to-start us:", " ".join("%.0f" % ((k1[i + 1][0] - k1[i][0]) / 1e3) for i in range(len(k1) - 1)))
t0 = k1[-a][0] - 300000
for e in ev:
    if e[0] >= t0 and e[0] < k1[-b][1] + 100000 and ((e[1] - e[0]) / 1e3 > mn or "COPY" in e[2]):
        print("%10.1f %9.1f %-32s %s" % ((e[0] - t0) / 1e3, (e[1] - e[0]) / 1e3, e[2], e[3]))
