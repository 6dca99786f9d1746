#!/bin/bash
cd /tmp; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $OUT/pmc_clk -o k1 -- python $GRAFT_REPO_ROOT/scripts/k1_bench.py > $OUT/pmc_clk.log 2>&1
python - <<EOF
import csv,collections
rows=list(csv.DictReader(open("$OUT/pmc_clk/k1_counter_collection.csv")))
for r in rows:
    if "mfma_kernel" in r["Kernel_Name"]:
        dur=int(r["End_Timestamp"])-int(r["Start_Timestamp"])
        print(r["Counter_Name"], r["Grid_Size"], float(r["Counter_Value"]), "dur_ns", dur, "GHz", float(r["Counter_Value"])/dur)
EOF
rocm-smi --showclocks 2>/dev/null | head -20
