#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2j
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2j
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout=300 -x -k scal > $OUT/tests_all.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/tests_all.log
timeout 300 python scripts/profile_scale.py all > $OUT/scale_batched2.jsonl 2>&1; cat $OUT/scale_batched2.jsonl
