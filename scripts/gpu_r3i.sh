#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3i
export OUT=$GRAFT_REPO_ROOT/gpurun_out/r3i
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "clique or colouring or config3 or object_scene" > $OUT/tests_clique.log 2>&1; echo "clique rc=$?"; tail -3 $OUT/tests_clique.log
OUT=$OUT bash scripts/gpu_r3h.sh 2>&1 | grep -E "colour|root_prune|exact|greedy|tls|gnc|select|peel|mfma"
