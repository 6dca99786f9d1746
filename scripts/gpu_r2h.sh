#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2h
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2h
cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_d3 -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-host-resident --no-latency --steps 12 --warmup 3 --depth 3 > $OUT/trace_d3.log 2>&1; tail -1 $OUT/trace_d3.log | cut -c75-200
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_d2 -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-host-resident --no-latency --steps 12 --warmup 3 --depth 2 > $OUT/trace_d2.log 2>&1; tail -1 $OUT/trace_d2.log | cut -c75-200
