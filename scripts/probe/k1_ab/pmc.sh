#!/bin/bash
# pmc.sh <tag>: counter passes on the product K1 (k1_probe "one" mode), one rocprofv3 run per set
cd $GRAFT_REPO_ROOT; TAG=$1; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
K="$GRAFT_REPO_ROOT/scripts/probe/k1_probe 64 10000 4 one"
i=0
while read -r set; do
  [ -z "$set" ] && continue
  i=$((i+1))
  (cd /tmp && timeout 120 rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -o t -- $K) > $OUT/p$i.log 2>&1; echo "p$i rc=$? : $set"
done <<SETS
SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_VALU_MFMA_COEXEC_CYCLES SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_WAVE_CYCLES SQ_BUSY_CYCLES
SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
TCP_TCP_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum
TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
TA_BUSY_avr TA_TA_BUSY_sum TCP_GATE_EN TCP_TA_TCP_STATE_READ_sum
SETS
python - <<PY
import csv,glob,collections
for f in sorted(glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True)):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "tim_graph_mfma3" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(f.split("/")[-3] if "/p" in f else f, {k: round(sum(v)/len(v)) for k,v in acc.items()}, "launches", {k: len(v) for k,v in acc.items()}.popitem()[1] if acc else 0)
PY
