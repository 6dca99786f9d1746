#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2i
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2i
P=$GRAFT_REPO_ROOT/scripts/probe/k1_probe
for lib in default MFMA6; do for v in 1 2 3; do
  if [ $lib = default ]; then L=""; else L=$GRAFT_REPO_ROOT/scripts/probe/variants/$lib; fi
  LD_LIBRARY_PATH=$L TEASER_K1_VARIANT=$v timeout 60 $P 64 10000 8 one > $OUT/m6_${lib}_$v.log 2>&1; echo "lib=$lib v=$v rc=$? $(grep -o '"k1_ms":[0-9.]*,\|"bitmap_hash":"[0-9a-f]*"' $OUT/m6_${lib}_$v.log | tr '\n' ' ')"
done; done
