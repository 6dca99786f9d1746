#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3n
export OUT=$GRAFT_REPO_ROOT/gpurun_out/r3n
export TMPDIR=/tmp
timeout 200 python -m pytest tests/test_gpu_features.py -m gpu -q -x > $OUT/tests_feat.log 2>&1; echo "feat rc=$?"; tail -3 $OUT/tests_feat.log
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c5 -o t -- python $GRAFT_REPO_ROOT/scripts/profile_config5.py > $OUT/c5.log 2>&1; tail -1 $OUT/c5.log | cut -c1-300
python - <<'PY'
import csv,glob,os
f=glob.glob(os.environ['OUT']+'/c5/*kernel_stats.csv')
if f:
    rows=list(csv.DictReader(open(f[0])))
    for r in rows[:22]: print(r['Name'][:60].ljust(60), r['Calls'], r['TotalDurationNs'], r['AverageNs'])
PY
