#!/bin/bash
# energy.sh <tag> <iters> <dir> [<dir> ...]: board power (hwmon, ~20 ms samples) x step time = energy per step of each library
# PB / PN / PRHO = batch, points, outlier ratio of the probe (default 64 x 10 000, 0.95);
# build (a name "<dir>:<mask>" runs <dir> with K1_PROBE_TAIL_SKIP=<mask>), synchronous steps ("one": K1, then its tail, nothing beside it) and the two-lane pipeline ("pipe").
cd $GRAFT_REPO_ROOT; TAG=$1; IT=$2; shift; shift; mkdir -p gpurun_out/$TAG; OUT=gpurun_out/$TAG
P=scripts/probe/k1_probe
setlib() { if [ $1 = new ]; then unset LD_LIBRARY_PATH; else export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/gpurun_ab/$1; fi; }
for w in "$@"; do
  lib=${w%%:*}; setlib $lib; if [ "$lib" != "$w" ]; then export K1_PROBE_TAIL_SKIP=${w#*:}; else unset K1_PROBE_TAIL_SKIP; fi
  for m in one pipe; do
    bash scripts/probe/clock_watch.sh $OUT/power_${w}_$m.txt -- timeout 120 $P ${PB:-64} ${PN:-10000} $IT $m ${PRHO:-0.95} default > $OUT/run_${w}_$m.jsonl 2>>$OUT/err.txt
  done
done
python scripts/probe/k1_ab/energy_summary.py $OUT "$@" | tee $OUT/energy.txt
