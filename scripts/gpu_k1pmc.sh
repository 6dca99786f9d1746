#!/bin/bash
cd /tmp; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAVES"; do
  tag=$(echo $set | cut -c1-12 | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmc_k1_$tag -o k1 -- python $GRAFT_REPO_ROOT/scripts/k1_bench.py > $OUT/pmc_k1_$tag.log 2>&1
  python - <<EOF
import csv,collections
rows=list(csv.DictReader(open("$OUT/pmc_k1_$tag/k1_counter_collection.csv")))
agg=collections.defaultdict(list)
for r in rows:
    if "mfma_kernel" in r["Kernel_Name"] and r["Grid_Size"]=="3276800":
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in agg.items(): print(k, len(v), sum(v)/len(v))
EOF
done
