#!/usr/bin/env python3
"""K1 launches of a rocprofv3 --kernel-trace CSV in time order: duration and start-to-start distance, in blocks of 12
(helper for comparing the HBM-resident and host-resident loops of bench.py).  usage: k1_gaps.py <dir> [grid_size]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "tim_graph_mfma" in r["Kernel_Name"]]
if len(sys.argv) > 2:
    rows = [r for r in rows if r["Grid_Size_Y"] == sys.argv[2]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
st = [int(r["Start_Timestamp"]) for r in rows]
du = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
for i in range(0, len(rows) - 1, 12):
    seg = range(i, min(i + 12, len(rows) - 1))
    print("launch %3d: dur us %s | start-to-start us %s" % (i, " ".join("%.0f" % du[k] for k in seg),
                                                          " ".join("%.0f" % ((st[k + 1] - st[k]) / 1e3) for k in seg)))
