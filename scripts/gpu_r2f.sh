#!/bin/bash
set -x
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2f
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2f
timeout 600 python -m pytest tests/test_gpu_features.py tests/test_cxx_facade.py -m gpu -q --timeout=300 > $OUT/tests_feat.log 2>&1; echo "feat tests rc=$?"; tail -30 $OUT/tests_feat.log
timeout 900 python -m pytest tests -m gpu -q --timeout=300 --deselect tests/test_gpu_features.py > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -6 $OUT/tests.log
timeout 120 python scripts/profile_stages.py > $OUT/stages.log 2>&1; grep '^{' $OUT/stages.log | cut -c1-400
timeout 300 python bench.py --no-cpu-baseline > $OUT/bench.log 2>&1; tail -1 $OUT/bench.log | cut -c1-300
