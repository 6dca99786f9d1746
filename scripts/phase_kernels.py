#!/usr/bin/env python3
"""Average duration per kernel in the HBM-resident and the host-resident loop of one bench.py configuration (rocprofv3
--kernel-trace CSV): the launches are split at the largest pause between two K1 launches of the given grid size.
usage: phase_kernels.py <dir> <k1_grid_size>"""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
k1 = sorted((int(r["Start_Timestamp"]) for r in rows if "tim_graph_mfma" in r["Kernel_Name"] and r["Grid_Size_Y"] == sys.argv[2]))
gaps = sorted(((k1[i + 1] - k1[i], i) for i in range(len(k1) - 1)), reverse=True)[:4]
cut = sorted(k1[i + 1] for _, i in gaps)
print("K1 launches of that grid:", len(k1), "phase boundaries at", [round((c - k1[0]) / 1e6, 1) for c in cut], "ms")
def phase(t):
    return sum(1 for c in cut if t >= c)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    t = int(r["Start_Timestamp"])
    if t < k1[0] or t > k1[-1] + 2000000:
        continue
    name = r["Kernel_Name"].replace("void ", "").replace("thip::", "").split("(")[0].split("<")[0]
    agg[name][phase(t)].append((int(r["End_Timestamp"]) - t) / 1e3)
for name in sorted(agg):
    print("%-28s" % name, "  ".join("ph%d: n=%d avg %.1f max %.0f" % (p, len(v), sum(v) / len(v), max(v)) for p, v in sorted(agg[name].items())))
