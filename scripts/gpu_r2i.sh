#!/bin/bash
# session I: K1 ablations -- the same kernel with its stores / atomics issued out of range (dropped by the
# hardware; arithmetic unchanged): which memory operation, if any, bounds the kernel
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2i
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2i
P=$GRAFT_REPO_ROOT/scripts/probe/k1_probe
for lib in default NO_EPILOGUE default; do
  for v in 1 2; do
    if [ $lib = default ]; then L=""; else L=$GRAFT_REPO_ROOT/scripts/probe/variants/$lib; fi
    LD_LIBRARY_PATH=$L TEASER_K1_VARIANT=$v timeout 20 $P 64 10000 8 one > $OUT/abl_${lib}_$v.log 2>&1; echo "lib=$lib v=$v rc=$? $(grep -o '"k1_ms":[0-9.]*,\|"bitmap_hash":"[0-9a-f]*"' $OUT/abl_${lib}_$v.log | tr '\n' ' ')"
  done
done
