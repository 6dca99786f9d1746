#!/bin/bash
# round 3, session A: the never-executed 2-rank path (--share-gpu), H2D probes, a baseline bench line
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3a
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3a
env | grep -E "HSA|HIP|ROC|GPU" > $OUT/env.txt
timeout 60 scripts/probe/hostread_probe 31 > $OUT/hostread.json 2> $OUT/hostread.err; echo "hostread rc=$?"; cat $OUT/hostread.json
HSA_ENABLE_SDMA=0 timeout 60 scripts/probe/hostread_probe 31 > $OUT/hostread_nosdma.json 2>&1; echo "nosdma rc=$?"; cut -c1-120 $OUT/hostread_nosdma.json
timeout 200 python bench.py --gpus 2 --share-gpu --steps 10 --no-cpu-baseline > $OUT/bench_2rank.log 2>&1; echo "2rank rc=$?"; tail -2 $OUT/bench_2rank.log | cut -c1-600
timeout 200 python bench.py --no-cpu-baseline > $OUT/bench.log 2>&1; echo "bench rc=$?"; tail -1 $OUT/bench.log | cut -c1-400
