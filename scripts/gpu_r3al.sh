#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3al
export OUT=$GRAFT_REPO_ROOT/gpurun_out/r3al
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "degenerate or float_key or batch_mid" > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -15 $OUT/tests.log
