#!/bin/bash
# final evidence of the round: K1 SQ counters (final kernel), the default bench line, the same command under
# rocprofv3 --kernel-trace --stats, smoke()
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/final
export OUT=$GRAFT_REPO_ROOT/gpurun_out/final
export TMPDIR=/tmp
cd /tmp
P="$GRAFT_REPO_ROOT/scripts/probe/k1_probe 64 10000 5 one"
timeout 100 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -o t -- $P > $OUT/kt.log 2>&1; echo "kt rc=$?"
timeout 100 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq1 -o t -- $P > $OUT/sq1.log 2>&1; echo "sq1 rc=$?"
timeout 100 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_WAVES --output-format csv -d $OUT/sq2 -o t -- $P > $OUT/sq2.log 2>&1; echo "sq2 rc=$?"
timeout 100 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o t -- $P > $OUT/fetch.log 2>&1; echo "fetch rc=$?"
timeout 100 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o t -- $P > $OUT/write.log 2>&1; echo "write rc=$?"
cd $GRAFT_REPO_ROOT
python scripts/summarize_pmc.py $(ls $OUT/fetch/*counter_collection.csv) $(ls $OUT/write/*counter_collection.csv) $OUT/pmc_traffic.json 64 10000 | grep -i "tim_graph"
python scripts/summarize_k1.py $OUT/k1_sq_counters.json 64 10000 $(ls $OUT/kt/*kernel_trace.csv) $(ls $OUT/sq1/*counter_collection.csv) $(ls $OUT/sq2/*counter_collection.csv) | cut -c1-600
# the bench reads profiles/*/pmc_traffic.json and k1_sq_counters.json: make the fresh ones visible to this run
mkdir -p profiles/r3d; cp $OUT/pmc_traffic.json $OUT/k1_sq_counters.json profiles/r3d/
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cut -c1-400 $OUT/bench.json
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -o t -- python $GRAFT_REPO_ROOT/bench.py --configs '' --no-cpu-baseline > $OUT/bench_prof.log 2>&1; echo "bench prof rc=$?"
grep '^{' $OUT/bench_prof.log | tail -1 > $OUT/bench_under_rocprof.json
cp $OUT/bench/*kernel_stats.csv $OUT/bench_kernel_stats.csv
head -6 $OUT/bench_kernel_stats.csv | cut -c1-160
cd $GRAFT_REPO_ROOT
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
