#!/bin/bash
# kernel_table.sh <tag> "<batch> <n> <rho>" ...: rocprofv3 --kernel-trace --stats of 30 synchronous steps of k1_probe per shape
cd $GRAFT_REPO_ROOT; TAG=$1; shift; mkdir -p gpurun_out/$TAG; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
export TMPDIR=/tmp
for cfg in "$@"; do
  set -- $cfg; name=${1}x${2}
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$name -o t -- $GRAFT_REPO_ROOT/scripts/probe/k1_probe $1 $2 30 one $3 > $OUT/kt_$name.log 2>&1)
  echo "== $cfg"
  python3 - "$OUT/kt_$name" <<'PY' | tee $OUT/kernels_$name.txt
import csv,glob,sys
f=glob.glob(sys.argv[1]+"/**/t_kernel_stats.csv",recursive=True)
if not f: print("no stats"); sys.exit()
for r in csv.DictReader(open(f[0])):
    print("%-64s calls %5s avg %9.1f us  %6s %%"%(r["Name"][:64],r["Calls"],float(r["AverageNs"])/1e3,r["Percentage"]))
PY
  rm -rf $OUT/kt_$name
done
