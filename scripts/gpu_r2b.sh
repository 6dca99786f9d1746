#!/bin/bash
# Round-2 GPU session B: K1 variants 0..3, priority-stream pipeline, full parity tests, bench.
set -x
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2b
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2b
P=$GRAFT_REPO_ROOT/scripts/probe/k1_probe
timeout 120 $P 64 10000 10 k1 > $OUT/probe_k1.jsonl 2> $OUT/probe_k1.err; echo "probe k1 rc=$?"; cat $OUT/probe_k1.jsonl; tail -3 $OUT/probe_k1.err
timeout 200 $P 64 10000 16 pipe > $OUT/probe_pipe.jsonl 2> $OUT/probe_pipe.err; echo "probe pipe rc=$?"; cat $OUT/probe_pipe.jsonl; tail -3 $OUT/probe_pipe.err
timeout 900 python -m pytest tests -m gpu -q --timeout=300 > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -25 $OUT/tests.log
timeout 400 python bench.py > $OUT/bench.log 2>&1; echo "bench rc=$?"; tail -1 $OUT/bench.log | cut -c1-900
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-latency --no-host-resident > $OUT/prof_bench.log 2>&1; echo "prof rc=$?"
find $OUT/prof -name "*kernel_stats*" | head -1 | xargs -I{} head -24 {}
