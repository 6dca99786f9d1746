#!/bin/bash
# round 3, session F: colouring work lists + batched device-side exact stage: clique tests, config tests, sharded test
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3f
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3f
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "clique or colouring or config3 or object_scene or benchmark or inlier_selection or kcore or edge_cases" > $OUT/tests_clique.log 2>&1; echo "clique rc=$?"; tail -5 $OUT/tests_clique.log
timeout 300 python -m pytest tests/test_gpu_features.py tests/test_gpu_sharded.py -m gpu -q -x > $OUT/tests_feat.log 2>&1; echo "feat rc=$?"; tail -5 $OUT/tests_feat.log
TEASER_K4_DEBUG=1 timeout 100 python scripts/profile_config5.py > $OUT/config5.jsonl 2> $OUT/config5.err; tail -1 $OUT/config5.jsonl; tail -3 $OUT/config5.err | cut -c1-300
timeout 400 python bench.py --no-cpu-baseline --steps 20 --configs 3,5 > $OUT/bench.log 2> $OUT/bench.err; echo "bench rc=$?"; tail -2 $OUT/bench.err
tail -1 $OUT/bench.log | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('top', d['value'], d['ms_per_step'])
for k,c in d['configs'].items():
    print(k, c['value'], c['ms_per_step'], c['ms_per_step_repeats'], c['stage_ms'])
"
