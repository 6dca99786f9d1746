#!/bin/bash
# config 3 under rocprofv3: per-launch durations of the colouring kernels of the last solve -> gpurun_out/<tag>/mis_trace_<name>.txt
TAG=$1; NAME=$2
mkdir -p gpurun_out/$TAG
bash scripts/gpu.sh $TAG c3prof > /dev/null 2>&1
f=$(find gpurun_out/$TAG/c3_prof -name "*kernel_trace.csv" | head -1)
python scripts/trace_kernels.py $f mis_,colour_,root_prune,exact_count 40 | grep -v "exact_count" | tee gpurun_out/$TAG/mis_trace_$NAME.txt | awk '{print $1, $5, $6}' | paste -sd' ' | fold -w 220
rm -rf gpurun_out/$TAG/c3_prof
