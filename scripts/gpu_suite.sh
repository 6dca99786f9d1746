#!/bin/bash
# the whole GPU suite, as the driver runs it at round end (+ durations)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/suite
export OUT=$GRAFT_REPO_ROOT/gpurun_out/suite
SECONDS=0
TEASER_CERT_DEBUG=$OUT/cert_warmup.txt timeout 1200 python -m pytest tests/ -x -q -m gpu --durations=12 > $OUT/gpu_tests.txt 2>&1; echo "suite rc=$? in ${SECONDS}s"
tail -25 $OUT/gpu_tests.txt; cat $OUT/cert_warmup.txt
cp gpurun_out/scale_bitmap_diff.json $OUT/ 2>/dev/null
