#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc SQ_* passes (one rocprofv3 run per counter set, as gpurun requires) into
per-kernel averages per launch.
usage: summarize_sq.py <out.json> <label> <kernel-substring> <counter_collection.csv> [more csv ...]
Appends/updates out.json[label] = {counter: mean over the launches of the matching kernel with the
largest grid}.  Units as the guide states: SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count
quad-cycles summed over waves, SQ_BUSY_CYCLES per SE, SQ_VALU_MFMA_BUSY_CYCLES cycles."""
import collections
import csv
import json
import os
import sys


def main():
    out, label, sub = sys.argv[1], sys.argv[2], sys.argv[3]
    doc = json.load(open(out)) if os.path.exists(out) else {}
    agg = collections.defaultdict(list)
    grids = collections.Counter()
    rows = []
    for path in sys.argv[4:]:
        for r in csv.DictReader(open(path)):
            if sub in r["Kernel_Name"]:
                rows.append(r)
                grids[int(r["Grid_Size"])] += 1
    if not rows:
        print("no rows for", sub)
        return
    g = max(grids)  # the batched launch
    for r in rows:
        if int(r["Grid_Size"]) == g:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    res = {k: sum(v) / len(v) for k, v in agg.items()}
    res["_launches"] = max(len(v) for v in agg.values())
    res["_grid_size"] = g
    if "SQ_WAVE_CYCLES" in res:
        wc = res["SQ_WAVE_CYCLES"]
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU",
                  "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA", "SQ_WAIT_INST_LDS"):
            if k in res:
                res["frac_of_wave_cycles:" + k] = res[k] / wc
    doc[label] = res
    json.dump(doc, open(out, "w"), indent=1, sort_keys=True)
    for k in sorted(res):
        print("%-40s %s" % (k, res[k]))


if __name__ == "__main__":
    main()
