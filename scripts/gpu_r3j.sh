#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3j
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3j
timeout 120 scripts/probe/k1_probe 64 10000 10 k1 > $OUT/k1.jsonl 2> $OUT/k1.err; echo "probe rc=$?"; cat $OUT/k1.jsonl; tail -3 $OUT/k1.err
TEASER_K1_DEBUG=1 TEASER_K1_VARIANT=8 timeout 60 scripts/probe/k1_probe 64 10000 2 one 2>&1 | tail -3
TEASER_K1_DEBUG=1 TEASER_K1_VARIANT=1 timeout 60 scripts/probe/k1_probe 64 10000 2 one 2>&1 | tail -3
