#!/bin/bash
# One GPU-box session: parity tests, bench, rocprofv3 kernel stats + PMC passes (FETCH / WRITE / SQ).
# Usage (from the repo root, on the GPU box via gpurun): bash scripts/gpu_session.sh <tag>
set -x
TAG=${1:-r2}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/$TAG
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
P=$GRAFT_REPO_ROOT/scripts/probe/k1_probe
BENCHQ="--steps 10 --warmup 2 --no-cpu-baseline --no-latency --no-host-resident"
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"
timeout 900 python -m pytest tests -m gpu -q --timeout=300 > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -4 $OUT/tests.log
timeout 400 python bench.py > $OUT/bench.log 2>&1; echo "bench rc=$?"; tail -1 $OUT/bench.log | cut -c1-600
for i in 1 2; do timeout 200 python bench.py --steps 30 --no-cpu-baseline --no-latency --no-host-resident > $OUT/bench_rep$i.log 2>&1; tail -1 $OUT/bench_rep$i.log | cut -c1-260; done
timeout 200 python bench.py --depth 1 --no-cpu-baseline --no-latency --no-host-resident > $OUT/bench_depth1.log 2>&1; tail -1 $OUT/bench_depth1.log | cut -c1-260
timeout 200 python bench.py --batch 16 --steps 40 --no-cpu-baseline --no-host-resident > $OUT/bench_batch16.log 2>&1; tail -1 $OUT/bench_batch16.log | cut -c1-260
timeout 120 $P 64 10000 10 k1 > $OUT/probe_k1.jsonl 2>/dev/null; cat $OUT/probe_k1.jsonl
timeout 120 python scripts/profile_stages.py > $OUT/stages.log 2>&1; grep '^{' $OUT/stages.log > $OUT/stages.jsonl; cut -c1-330 $OUT/stages.jsonl
timeout 100 python scripts/profile_stages.py big >> $OUT/stages.log 2>&1; grep '^{' $OUT/stages.log > $OUT/stages.jsonl
cd /tmp
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py $BENCHQ > $OUT/prof_bench.log 2>&1; echo "prof rc=$?"
timeout 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-latency --no-host-resident > $OUT/pmc_fetch.log 2>&1; echo "pmc fetch rc=$?"
timeout 240 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-latency --no-host-resident > $OUT/pmc_write.log 2>&1; echo "pmc write rc=$?"
python $GRAFT_REPO_ROOT/scripts/summarize_pmc.py $(find $OUT/pmc_fetch -name "*counter_collection.csv") $(find $OUT/pmc_write -name "*counter_collection.csv") $OUT/pmc_traffic.json 64 10000 | grep -i "tim_graph\|greedy\|peel" | head
SET1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES"
SET2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"
i=0
for set in "$SET1" "$SET2"; do
  i=$((i+1))
  timeout 120 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/sq_$i -o k1 -- $P 64 10000 4 one > $OUT/sq_$i.log 2>&1; echo "sq set$i rc=$?"
done
python $GRAFT_REPO_ROOT/scripts/summarize_sq.py $OUT/k1_sq_counters.json tim_graph_mfma_kernel tim_graph_mfma_kernel $(find $OUT/sq_1 $OUT/sq_2 -name "*counter_collection.csv") | grep -i "frac\|INSTS_VALU \|INSTS_MFMA"
python $GRAFT_REPO_ROOT/scripts/summarize_sq.py $OUT/k1_sq_counters.json greedy_clique_kernel greedy_clique_kernel $(find $OUT/sq_1 $OUT/sq_2 -name "*counter_collection.csv") | grep -i "frac\|LDS"
cd $GRAFT_REPO_ROOT
find $OUT/prof -name "*kernel_stats*" | head -1 | xargs -I{} head -16 {}
