#!/bin/bash
# estimate_scaling = true: kernel table of the large path (n = 2000 and 10 000 single solves), mid-size batch probe
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3r
export OUT=$GRAFT_REPO_ROOT/gpurun_out/r3r
export TMPDIR=/tmp
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/k7 -o t -- python $GRAFT_REPO_ROOT/scripts/profile_scale.py large > $OUT/k7.log 2>&1; echo "k7 rc=$?"
grep '^{' $OUT/k7.log
python - <<'PY'
import csv,glob,os
f=glob.glob(os.environ['OUT']+'/k7/*kernel_stats.csv')
if f:
    rows=list(csv.DictReader(open(f[0])))
    for r in rows[:16]: print(r['Name'][:100].ljust(100), r['Calls'], r['TotalDurationNs'], r['AverageNs'])
PY
cd $GRAFT_REPO_ROOT
for n in 2000; do
  TEASER_SCALE_MID_BATCH=0 timeout 100 python scripts/scale_batch_probe.py $n 64 2>&1 | tail -1 | tee -a $OUT/scale_batch.jsonl
  timeout 100 python scripts/scale_batch_probe.py $n 64 2>&1 | tail -1 | tee -a $OUT/scale_batch.jsonl
done
