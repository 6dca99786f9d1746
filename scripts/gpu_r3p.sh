#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3p
export OUT=$GRAFT_REPO_ROOT/gpurun_out/r3p
timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "estimate_scaling" > $OUT/tests_scale.log 2>&1; echo "scale rc=$?"; tail -5 $OUT/tests_scale.log
for n in 800 2000; do
  TEASER_SCALE_MID_BATCH=0 timeout 300 python scripts/scale_batch_probe.py $n 64 2>&1 | tail -1 | tee -a $OUT/scale_batch.jsonl
  timeout 300 python scripts/scale_batch_probe.py $n 64 2>&1 | tail -1 | tee -a $OUT/scale_batch.jsonl
done
