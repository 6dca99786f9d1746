#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for i in 1 2 3 4; do
  timeout 60 python -m pytest tests -m gpu -q -x -k "filter_fallbacks or adversarial" > gpurun_out/dbg_$i.log 2>&1; echo "run $i rc=$?"; tail -1 gpurun_out/dbg_$i.log
done
