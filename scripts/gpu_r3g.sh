#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3g
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3g
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "clique or colouring or config3 or object_scene or benchmark or inlier_selection or kcore or edge_cases" > $OUT/tests_clique.log 2>&1; echo "clique rc=$?"; tail -5 $OUT/tests_clique.log
timeout 200 python -m pytest tests/test_gpu_features.py -m gpu -q -x > $OUT/tests_feat.log 2>&1; echo "feat rc=$?"; tail -3 $OUT/tests_feat.log
TEASER_K4_DEBUG=1 timeout 400 python bench.py --no-cpu-baseline --no-host-resident --steps 10 --configs 3,5 --repeats 1 > $OUT/bench.log 2> $OUT/bench.err; echo "bench rc=$?"
grep "exact search" $OUT/bench.err | sort -t' ' -k32 -n | awk '{print $5, $7, $12, $14, $33, $34, $NF, $(NF-1)}' | sort -k5 -n | tail -5
grep "exact search" $OUT/bench.err | head -2 | cut -c1-300
tail -1 $OUT/bench.log | python -c "
import sys,json
d=json.loads(sys.stdin.read())
for k,c in d['configs'].items():
    print(k, c['value'], c['ms_per_step'], c['stage_ms'])
"
