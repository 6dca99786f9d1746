#!/bin/bash
# K1 evidence for the new default kernel: bench under --kernel-trace --stats, then the probe (64 x 10 k, K1 only)
# under separate --pmc passes: FETCH_SIZE, WRITE_SIZE, two SQ sets; and a --kernel-trace pass for its duration.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3q2
export OUT=$GRAFT_REPO_ROOT/gpurun_out/r3q2
export TMPDIR=/tmp
cd /tmp
P="$GRAFT_REPO_ROOT/scripts/probe/k1_probe 64 10000 5 one"
timeout 100 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -o t -- $P > $OUT/kt.log 2>&1; echo "kt rc=$?"
timeout 100 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o t -- $P > $OUT/fetch.log 2>&1; echo "fetch rc=$?"
timeout 100 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o t -- $P > $OUT/write.log 2>&1; echo "write rc=$?"
timeout 100 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq1 -o t -- $P > $OUT/sq1.log 2>&1; echo "sq1 rc=$?"
timeout 100 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_WAVES --output-format csv -d $OUT/sq2 -o t -- $P > $OUT/sq2.log 2>&1; echo "sq2 rc=$?"
cd $GRAFT_REPO_ROOT
python scripts/summarize_pmc.py $(ls $OUT/fetch/*counter_collection.csv) $(ls $OUT/write/*counter_collection.csv) $OUT/pmc_traffic.json 64 10000 | grep -i "tim_\|degree" 
python scripts/summarize_k1.py $OUT/k1_sq_counters.json 64 10000 $(ls $OUT/kt/*kernel_trace.csv) $(ls $OUT/sq1/*counter_collection.csv) $(ls $OUT/sq2/*counter_collection.csv)
bash scripts/gpu_r3r.sh
