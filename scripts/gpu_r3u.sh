#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3u
export OUT=$GRAFT_REPO_ROOT/gpurun_out/r3u
timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "scal or tim_product" > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -5 $OUT/tests.log
timeout 100 python scripts/profile_scale.py large 2>&1 | grep '^{' | tee $OUT/scale_large.jsonl
TEASER_SCALE_SORT64=1 timeout 100 python scripts/profile_scale.py large 2>&1 | grep '^{' | tee $OUT/scale_large_sort64.jsonl
timeout 100 python scripts/scale_batch_probe.py 2000 64 2>&1 | tail -1 | tee -a $OUT/scale_batch.jsonl
timeout 100 python scripts/scale_batch_probe.py 800 64 2>&1 | tail -1 | tee -a $OUT/scale_batch.jsonl
cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/k7 -o t -- python $GRAFT_REPO_ROOT/scripts/profile_scale.py large > $OUT/k7.log 2>&1; echo "k7 rc=$?"
python - <<'PY'
import csv,glob,os
f=glob.glob(os.environ['OUT']+'/k7/*kernel_trace.csv')
rows=list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'trim_endpoints_kernel' in r['Kernel_Name']]
i0=idx[-1]; t0=int(rows[i0]['Start_Timestamp'])
for r in rows[i0:i0+22]:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    if (e-s)>20000: print("%9.3f dur %8.3f  %s"%((s-t0)/1e6,(e-s)/1e6,r['Kernel_Name'][:80]))
PY
