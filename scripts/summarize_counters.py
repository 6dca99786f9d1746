#!/usr/bin/env python3
"""Per-kernel totals of rocprofv3 --pmc passes (one rocprofv3 run per counter set, as gpurun requires).
usage: summarize_counters.py <out.json> <label> <kernel-substring[,substring...]> <counter_collection.csv> [...]
out.json[label][kernel] = {counter: sum over the kernel's launches, "_launches": n}; derived LDS lines:
lds_bank_conflict_frac = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE (cycles), lds_per_vmem_rd = SQ_INSTS_LDS /
SQ_INSTS_VMEM_RD (the share of a kernel's data accesses served from LDS = lds / (lds + vmem_rd))."""
import collections
import csv
import json
import os
import re
import sys


def short(name):
    name = re.sub(r"^void\s+", "", name)
    name = name.split("(")[0]
    return name.replace("thip::", "")


def main():
    out, label, subs = sys.argv[1], sys.argv[2], sys.argv[3].split(",")
    doc = json.load(open(out)) if os.path.exists(out) else {}
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(lambda: collections.defaultdict(set))
    for path in sys.argv[4:]:
        for r in csv.DictReader(open(path)):
            k = short(r["Kernel_Name"])
            if not any(s in k for s in subs):
                continue
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
            launches[k][r["Counter_Name"]].add(r.get("Dispatch_Id", r.get("Correlation_Id", "")))
    res = {}
    for k, c in tot.items():
        e = dict(c)
        e["_launches"] = max(len(v) for v in launches[k].values())
        if e.get("SQ_LDS_IDX_ACTIVE"):
            e["lds_bank_conflict_frac"] = e.get("SQ_LDS_BANK_CONFLICT", 0.0) / e["SQ_LDS_IDX_ACTIVE"]
        if "SQ_INSTS_LDS" in e and "SQ_INSTS_VMEM_RD" in e:
            e["lds_share_of_reads"] = e["SQ_INSTS_LDS"] / max(e["SQ_INSTS_LDS"] + e["SQ_INSTS_VMEM_RD"], 1.0)
        res[k] = e
    doc[label] = res
    json.dump(doc, open(out, "w"), indent=1, sort_keys=True)
    for k in sorted(res):
        print(k, json.dumps({a: (round(b, 4) if isinstance(b, float) else b) for a, b in sorted(res[k].items())}))


if __name__ == "__main__":
    main()
