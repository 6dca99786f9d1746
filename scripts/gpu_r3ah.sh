#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3ah
export OUT=$GRAFT_REPO_ROOT/gpurun_out/r3ah
TEASER_HEU_TRACE=$OUT/heu_trace_depth1.txt timeout 150 python bench.py --configs '' --no-cpu-baseline --no-latency --no-host-resident --steps 80 --pool 8 --depth 1 > $OUT/b1.json 2> $OUT/b1.err; echo "rc=$?"
head -16 $OUT/heu_trace_depth1.txt | cut -c1-200
