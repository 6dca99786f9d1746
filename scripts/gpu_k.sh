#!/bin/bash
# bash scripts/gpu_k.sh "<pytest -k expr>"  -- run a subset of the GPU parity tests
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q --timeout=600 -x -k "$1" > gpurun_out/tests_k.log 2>&1; echo "tests rc=$?"
tail -40 gpurun_out/tests_k.log
