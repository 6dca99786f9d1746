#!/bin/bash
# Round-2 GPU session A: probes (K1 variants, async pipeline), SQ counters of K1, new parity tests, bench.
set -x
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2a
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2a
P=$GRAFT_REPO_ROOT/scripts/probe/k1_probe
timeout 120 $P 64 10000 10 k1 > $OUT/probe_k1.jsonl 2> $OUT/probe_k1.err; echo "probe k1 rc=$?"; cat $OUT/probe_k1.jsonl; tail -3 $OUT/probe_k1.err
timeout 200 $P 64 10000 16 pipe > $OUT/probe_pipe.jsonl 2> $OUT/probe_pipe.err; echo "probe pipe rc=$?"; cat $OUT/probe_pipe.jsonl; tail -3 $OUT/probe_pipe.err
cd /tmp
rocprofv3 -L > $OUT/counters_list.txt 2>&1; grep -c SQ_ $OUT/counters_list.txt
SET1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES"
SET2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"
SET3="SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_WAVE_CYCLES"
for v in 0 1; do
  i=0
  for set in "$SET1" "$SET2" "$SET3"; do
    i=$((i+1))
    TEASER_K1_VARIANT=$v timeout 120 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/sq_v${v}_$i -o k1 -- $P 64 10000 4 one > $OUT/sq_v${v}_$i.log 2>&1; echo "sq v$v set$i rc=$?"
  done
  python $GRAFT_REPO_ROOT/scripts/summarize_sq.py $OUT/k1_sq_counters.json variant$v tim_graph_mfma_kernel $(find $OUT/sq_v${v}_1 $OUT/sq_v${v}_2 $OUT/sq_v${v}_3 -name "*counter_collection.csv")
done
cd $GRAFT_REPO_ROOT
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
timeout 900 python -m pytest tests -m gpu -q --timeout=300 -x > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -15 $OUT/tests.log
timeout 400 python bench.py > $OUT/bench.log 2>&1; echo "bench rc=$?"; tail -1 $OUT/bench.log | cut -c1-1500
timeout 200 python bench.py --depth 1 --no-cpu-baseline --no-host-resident --no-latency > $OUT/bench_d1.log 2>&1; tail -1 $OUT/bench_d1.log | cut -c1-400
