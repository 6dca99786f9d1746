#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3ak
export OUT=$GRAFT_REPO_ROOT/gpurun_out/r3ak
export TMPDIR=/tmp
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c5 -o t -- python $GRAFT_REPO_ROOT/scripts/profile_config5.py > $OUT/c5.log 2>&1; echo "rc=$?"
grep '^{' $OUT/c5.log | tail -1 > $OUT/config5.jsonl; cut -c1-400 $OUT/config5.jsonl
head -8 $OUT/c5/t_kernel_stats.csv | cut -c1-150
cd $GRAFT_REPO_ROOT
timeout 100 python scripts/frontend_probe.py 2>&1 | tail -1 | tee $OUT/frontend.json
