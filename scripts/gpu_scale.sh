#!/bin/bash
# GPU check of the estimate_scaling=true path: parity tests + timing at N=10k.
set -x
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests -m gpu -q --timeout=600 -x -k "scal" > $OUT/tests_scale.log 2>&1; echo "tests rc=$?"
tail -30 $OUT/tests_scale.log
timeout 600 python scripts/scale_timing.py > $OUT/scale_timing.log 2>&1; cat $OUT/scale_timing.log
