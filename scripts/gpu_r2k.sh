#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2k
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2k
TEASER_K4_DEBUG=1 TEASER_CLIQUE_LIMIT=4 TEASER_PROFILE_WATCHDOG=70 timeout 90 python scripts/profile_config5.py > $OUT/config5.jsonl 2> $OUT/config5.err; echo rc=$?; cat $OUT/config5.jsonl | cut -c1-700; grep -v "^  File \"/usr" $OUT/config5.err | tail -3 | cut -c1-300
timeout 100 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout=100 -x -k "clique or exact or planted" > $OUT/tests_exact.log 2>&1; echo "exact tests rc=$?"; tail -2 $OUT/tests_exact.log | cut -c1-200
