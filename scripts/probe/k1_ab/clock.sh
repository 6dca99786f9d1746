#!/bin/bash
# pmc2.sh <tag> <dir> ... : GRBM_GUI_ACTIVE (clock cycles) and duration of K1 per library build -> effective clock
cd $GRAFT_REPO_ROOT; TAG=$1; shift; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
K="$GRAFT_REPO_ROOT/scripts/probe/k1_probe 64 10000 6 one"
for w in "$@"; do
  if [ $w = new ]; then unset LD_LIBRARY_PATH; else export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/gpurun_ab/$w; fi
  (cd /tmp && timeout 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $OUT/c_$w -o t -- $K) > $OUT/c_$w.log 2>&1
  (cd /tmp && timeout 120 rocprofv3 --kernel-trace --output-format csv -d $OUT/k_$w -o t -- $K) > $OUT/k_$w.log 2>&1
  python - <<PY
import csv,glob
c=[float(r["Counter_Value"]) for f in glob.glob("$OUT/c_$w/**/*counter_collection.csv", recursive=True) for r in csv.DictReader(open(f)) if "tim_graph_mfma3" in r["Kernel_Name"] and r["Counter_Name"]=="GRBM_GUI_ACTIVE"]
d=[(float(r["End_Timestamp"])-float(r["Start_Timestamp"])) for f in glob.glob("$OUT/k_$w/**/*kernel_trace.csv", recursive=True) for r in csv.DictReader(open(f)) if "tim_graph_mfma3" in r["Kernel_Name"]]
c=sorted(c)[len(c)//2]; d=sorted(d)[len(d)//2]
print("$w: GRBM_GUI_ACTIVE %.0f (per XCD %.0f)  duration %.1f us  -> %.3f GHz" % (c, c/8, d/1e3, c/8/d))
PY
done | tee $OUT/clock.txt
