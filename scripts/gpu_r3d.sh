#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3d
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3d
for m in 0 1 2; do
TEASER_HIP_COPY_STREAM=$m timeout 120 python scripts/host_loop_probe.py 40 64 10000 host1,host0,dev 2>&1 | grep "ms/step" | tr '\n' ' '; echo " copy_stream=$m"
done
GPU_MAX_HW_QUEUES=8 TEASER_HIP_COPY_STREAM=1 timeout 120 python scripts/host_loop_probe.py 40 64 10000 host1,host0,dev 2>&1 | grep "ms/step" | tr '\n' ' '; echo " hwq8 copy_stream=1"
