#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3d
for v in 11 1; do
TEASER_K1_VARIANT=$v timeout 120 python scripts/host_loop_probe.py 40 64 10000 dev,host1,host0 2>&1 | grep "ms/step" | tr '\n' ' '; echo " variant=$v"
done
