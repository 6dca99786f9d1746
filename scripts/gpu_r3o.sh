#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3o
export OUT=$GRAFT_REPO_ROOT/gpurun_out/r3o
timeout 200 python scripts/frontend_probe.py 2>&1 | tail -1 | tee $OUT/frontend.json
