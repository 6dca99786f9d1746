#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3ag
export OUT=$GRAFT_REPO_ROOT/gpurun_out/r3ag
TEASER_HEU_TRACE=$OUT/heu_trace_depth1.txt timeout 150 python bench.py --configs '' --no-cpu-baseline --no-latency --no-host-resident --steps 80 --pool 8 --depth 1 > $OUT/b1.json 2> $OUT/b1.err; echo "rc=$?"
sed -n 5,7p $OUT/heu_trace_depth1.txt | cut -c1-260
TEASER_HEU_TRACE=$OUT/heu_trace_overlap.txt timeout 150 python bench.py --configs '' --no-cpu-baseline --no-latency --no-host-resident --steps 80 --pool 8 > $OUT/b.json 2> $OUT/b.err; echo "rc=$?"
sed -n 5,8p $OUT/heu_trace_overlap.txt | cut -c1-260
for i in 1 2; do timeout 150 python bench.py --configs '' --no-cpu-baseline --no-latency --no-host-resident --steps 60 2>/dev/null | grep '^{' | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print({'value':round(d['value']),'ms_per_step':round(d['ms_per_step'],4)})"; done
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "parity or config or clique or colouring or batch or async or fixtures or bunny" > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -2 $OUT/tests.log
