#!/usr/bin/env python3
"""K1 (tim_graph_mfma*_kernel) issue statistics from rocprofv3 --pmc SQ passes of scripts/probe/k1_probe
(one run per counter set; gpurun forbids trace domains beside --pmc) plus the kernel's duration from a separate
--kernel-trace run of the same command -> profiles/<round>/k1_sq_counters.json in the schema bench.py reads.
usage: summarize_k1.py <out.json> <batch> <n> <kernel_trace.csv> <counter_collection.csv> [more ...]
Units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves;
SQ_VALU_MFMA_BUSY_CYCLES counts cycles.  Fractions of SIMD time use duration x shader clock x 1024 SIMDs, the
clock taken as SQ_BUSY_CYCLES-free: 2.4 GHz nominal is NOT assumed -- GRBM_GUI_ACTIVE of the same launch is."""
import collections
import csv
import json
import sys


def main():
    out, batch, n, trace = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(trace)):
        if "tim_graph_mfma" in r["Kernel_Name"]:
            grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
            dur[(r["Kernel_Name"].split("(")[0].replace("void ", ""), grid)].append(
                int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    key = max(dur, key=lambda k: k[1])  # the batched launch
    ns = sorted(dur[key])[len(dur[key]) // 2]
    agg = collections.defaultdict(list)
    for path in sys.argv[5:]:
        for r in csv.DictReader(open(path)):
            if "tim_graph_mfma" in r["Kernel_Name"] and int(r["Grid_Size"]) == key[1]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    c = {k: sum(v) / len(v) for k, v in agg.items()}
    pairs = batch * n * (n - 1) / 2
    k = {"kernel": key[0], "grid_size": key[1], "batch": batch, "n": n, "duration_ns_kernel_trace": ns,
         "launches_per_pass": max(len(v) for v in agg.values()), "counters": c}
    if "GRBM_GUI_ACTIVE" in c:
        # one value per XCD summed by the tool on this stack: 8 XCDs active for the whole launch
        cyc = c["GRBM_GUI_ACTIVE"] / 8.0
        k["shader_cycles_per_launch"] = cyc
        k["shader_clock_ghz"] = cyc / ns
    else:
        cyc = ns * 2.4
    simd_cycles = cyc * 1024
    if "SQ_ACTIVE_INST_VALU" in c:
        k["valu_busy_frac"] = 4 * c["SQ_ACTIVE_INST_VALU"] / simd_cycles
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
        k["mfma_busy_frac"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles
    if "SQ_INSTS_VALU" in c:
        k["valu_insts_per_1024_pairs"] = c["SQ_INSTS_VALU"] / (pairs / 1024)
    if "SQ_INSTS_MFMA" in c:
        k["mfma_insts_per_1024_pairs"] = c["SQ_INSTS_MFMA"] / (pairs / 1024)
    if "SQ_INSTS_VMEM_RD" in c:
        k["vmem_insts_per_1024_pairs"] = (c["SQ_INSTS_VMEM_RD"] + c.get("SQ_INSTS_VMEM_WR", 0.0)) / (pairs / 1024)
    if "TA_TA_BUSY_sum" in c:  # summed over the 256 texture-address units (one per CU)
        k["ta_busy_frac"] = c["TA_TA_BUSY_sum"] / 256.0 / cyc
    if "TCP_TOTAL_CACHE_ACCESSES_sum" in c and "TCP_TCC_READ_REQ_sum" in c and c["TCP_TOTAL_CACHE_ACCESSES_sum"] > 0:
        k["l1_hit_rate"] = 1.0 - c["TCP_TCC_READ_REQ_sum"] / c["TCP_TOTAL_CACHE_ACCESSES_sum"]
    if "SQ_VMEM_TA_ADDR_FIFO_FULL" in c:  # summed over the CUs' sequencers
        k["ta_addr_fifo_full_frac"] = c["SQ_VMEM_TA_ADDR_FIFO_FULL"] / 256.0 / cyc
    if "SQ_WAVE_CYCLES" in c:
        wc = c["SQ_WAVE_CYCLES"]
        for name, src in (("wave_issue_frac", "SQ_ACTIVE_INST_ANY"), ("wave_wait_frac", "SQ_WAIT_ANY"),
                          ("wave_stall_frac", "SQ_WAIT_INST_ANY")):
            if src in c:
                k[name] = c[src] / wc
        k["resident_waves_per_simd"] = 4 * wc / simd_cycles
    json.dump({"note": __doc__.strip().split("usage")[0].strip(), "kernels": [k]}, open(out, "w"), indent=1)
    print(json.dumps({f: k[f] for f in k if f != "counters"}))


if __name__ == "__main__":
    main()
