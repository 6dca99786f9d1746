#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 60 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 200 python -m pytest tests -m gpu -q -x > gpurun_out/tests_final.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/tests_final.log
timeout 100 python bench.py 2>/dev/null | tail -1 > gpurun_out/bench_final.json; cut -c1-260 gpurun_out/bench_final.json
TEASER_HIP_PIPELINE=2 timeout 60 python -m pytest tests -m gpu -q -x -k "batch or config4" 2>&1 | tail -1
