#!/usr/bin/env python3
"""Per-launch durations of the kernels whose name contains one of the given substrings, from a rocprofv3 kernel trace
csv: the LAST solve's launches in stream order.   trace_kernels.py <kernel_trace.csv> <substr>[,<substr>...] [n_last]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
subs = sys.argv[2].split(",")
n_last = int(sys.argv[3]) if len(sys.argv) > 3 else 40
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sel = [r for r in rows if any(s in r["Kernel_Name"] for s in subs)][-n_last:]
t0 = int(sel[0]["Start_Timestamp"]) if sel else 0
prev_end = None
for r in sel:
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (st - prev_end) / 1e3 if prev_end is not None else 0.0
    print("%-40s start %9.1f us  dur %7.1f us  gap %6.1f us  grid %s wg %s" % (r["Kernel_Name"].split("(")[0][-40:], (st - t0) / 1e3, (en - st) / 1e3, gap,
          r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?"))))
    prev_end = en
if sel:
    print("span %.1f us, busy %.1f us" % ((int(sel[-1]["End_Timestamp"]) - t0) / 1e3, sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in sel) / 1e3))
