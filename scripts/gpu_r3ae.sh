#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3ae
export OUT=$GRAFT_REPO_ROOT/gpurun_out/r3ae
TEASER_HEU_TRACE=$OUT/heu_trace_overlap.txt timeout 150 python bench.py --configs '' --no-cpu-baseline --no-latency --no-host-resident --steps 80 --pool 8 > $OUT/b.json 2> $OUT/b.err; echo "rc=$?"
sed -n 5,14p $OUT/heu_trace_overlap.txt
TEASER_HEU_TRACE=$OUT/heu_trace_depth1.txt timeout 150 python bench.py --configs '' --no-cpu-baseline --no-latency --no-host-resident --steps 80 --pool 8 --depth 1 > $OUT/b1.json 2> $OUT/b1.err; echo "rc=$?"
sed -n 5,10p $OUT/heu_trace_depth1.txt
