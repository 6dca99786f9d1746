#!/bin/bash
# Round-2 GPU session C: coalesced K1 operands + degrees folded into K1; pipeline; tests; bench (+ HW queue knob).
set -x
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2c
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2c
P=$GRAFT_REPO_ROOT/scripts/probe/k1_probe
timeout 120 $P 64 10000 10 k1 > $OUT/probe_k1.jsonl 2> $OUT/probe_k1.err; echo "probe k1 rc=$?"; cat $OUT/probe_k1.jsonl; tail -3 $OUT/probe_k1.err
timeout 200 $P 64 10000 16 pipe > $OUT/probe_pipe.jsonl 2> $OUT/probe_pipe.err; echo "probe pipe rc=$?"; cat $OUT/probe_pipe.jsonl; tail -3 $OUT/probe_pipe.err
GPU_MAX_HW_QUEUES=8 timeout 200 $P 64 10000 16 pipe > $OUT/probe_pipe_q8.jsonl 2> $OUT/probe_pipe_q8.err; echo "probe pipe q8 rc=$?"; cat $OUT/probe_pipe_q8.jsonl
timeout 900 python -m pytest tests -m gpu -q --timeout=300 -x > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -25 $OUT/tests.log
timeout 400 python bench.py --no-cpu-baseline > $OUT/bench.log 2>&1; echo "bench rc=$?"; tail -1 $OUT/bench.log | cut -c1-700
GPU_MAX_HW_QUEUES=8 timeout 400 python bench.py --no-cpu-baseline > $OUT/bench_q8.log 2>&1; echo "bench q8 rc=$?"; tail -1 $OUT/bench_q8.log | cut -c1-400
timeout 400 python bench.py --no-cpu-baseline --depth 2 > $OUT/bench_d2.log 2>&1; echo "bench d2 rc=$?"; tail -1 $OUT/bench_d2.log | cut -c1-400
cd /tmp
SET1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES"
TEASER_K1_VARIANT=1 timeout 120 rocprofv3 --kernel-trace --pmc $SET1 --output-format csv -d $OUT/sq_v1_1 -o k1 -- $P 64 10000 4 one > $OUT/sq_v1_1.log 2>&1; echo "sq rc=$?"
python $GRAFT_REPO_ROOT/scripts/summarize_sq.py $OUT/k1_sq_counters.json variant1 tim_graph_mfma_kernel $(find $OUT/sq_v1_1 -name "*counter_collection.csv")
