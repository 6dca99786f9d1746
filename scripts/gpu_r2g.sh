#!/bin/bash
# session G: K1 after the store/atomic re-ordering, buffer-descriptor operand loads, v_bfi transposes;
# variants timed one per process, then the whole GPU suite and the bench per variant
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2g
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2g
P=$GRAFT_REPO_ROOT/scripts/probe/k1_probe
for v in -1 0 1 2 3; do
  TEASER_K1_VARIANT=$v timeout 60 $P 64 10000 8 one > $OUT/one_$v.log 2>&1; echo "v=$v rc=$? $(grep -o '"k1_ms":[0-9.]*,\|"aux_ms":[0-9.]*\|"bitmap_hash":"[0-9a-f]*"' $OUT/one_$v.log | tr '\n' ' ')"
done
cat $OUT/one_-1.log $OUT/one_0.log $OUT/one_1.log $OUT/one_2.log $OUT/one_3.log | grep '^{' > $OUT/probe_k1.jsonl
timeout 900 python -m pytest tests -m gpu -q --timeout=300 > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -4 $OUT/tests.log
for v in 1 2 3; do
TEASER_K1_VARIANT=$v timeout 300 python bench.py --no-cpu-baseline > $OUT/bench_v$v.log 2>&1; tail -1 $OUT/bench_v$v.log | cut -c1-260
done
