#!/bin/bash
# quick GPU check: parity tests + per-stage profile.  bash scripts/gpu_quick.sh <tag> [pytest -k expr]
set -x
TAG=${1:-q}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
if [ -n "$2" ]; then
timeout 300 python -m pytest tests -m gpu -q -x -k "$2" > $OUT/tests_$TAG.log 2>&1; echo "tests rc=$?"
else
timeout 300 python -m pytest tests -m gpu -q -x > $OUT/tests_$TAG.log 2>&1; echo "tests rc=$?"
fi
tail -15 $OUT/tests_$TAG.log
timeout 600 python scripts/profile_stages.py > $OUT/stages_$TAG.log 2>&1; cat $OUT/stages_$TAG.log
timeout 600 python scripts/profile_stages.py big > $OUT/stages_big_$TAG.log 2>&1; cat $OUT/stages_big_$TAG.log
