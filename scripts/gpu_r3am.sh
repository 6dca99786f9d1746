#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3am
timeout 120 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "reference_no_max" > gpurun_out/r3am/tests.log 2>&1; echo "tests rc=$?"; tail -12 gpurun_out/r3am/tests.log
