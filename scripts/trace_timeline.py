#!/usr/bin/env python3
"""Timeline summary of a rocprofv3 --kernel-trace CSV of bench.py: per step (one K1 launch = one step) the
K1 span, the span of the batch's tail kernels, and the distance between consecutive K1 starts.
usage: trace_timeline.py <kernel_trace.csv> [first_step]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 6
def short(n):
    n = n.replace("void ", "").replace("thip::", "")
    return n.split("(")[0].split("<")[0]
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), int(r["Stream_Id"])) for r in rows]
ev.sort()
t0 = ev[0][0]
k1 = [e for e in ev if e[2].startswith("tim_graph_mfma")]
print("K1 launches", len(k1))
per = [(k1[i + 1][0] - k1[i][0]) / 1e3 for i in range(len(k1) - 1)]
print("K1 start-to-start us:", " ".join("%.0f" % p for p in per))
print("K1 durations us     :", " ".join("%.0f" % ((e[1] - e[0]) / 1e3) for e in k1))
# per stream chain after each K1: list kernels with start offsets relative to that K1's start
for i in range(skip, min(skip + 3, len(k1))):
    s, e, _, st = k1[i]
    nxt = [x for x in k1 if x[3] == st and x[0] > s]
    lim = nxt[0][0] if nxt else ev[-1][1] + 1
    chain = [x for x in ev if x[3] == st and s - 200000 <= x[0] < lim]
    print("--- step %d stream %d" % (i, st))
    for x in chain:
        print("   %-28s start %8.1f us  dur %7.1f us" % (x[2], (x[0] - s) / 1e3, (x[1] - x[0]) / 1e3))
