#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3w
export OUT=$GRAFT_REPO_ROOT/gpurun_out/r3w
for d in 0 1 2 3; do
  echo "dbg=$d"; TEASER_FIX_DEBUG=$d timeout 100 python scripts/profile_scale.py large 2>&1 | grep '^{' | grep 10000
done
