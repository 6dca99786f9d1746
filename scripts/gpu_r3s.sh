#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3s
export OUT=$GRAFT_REPO_ROOT/gpurun_out/r3s
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "estimate_scaling or scalar_tls or k1_ or solve_for_scale" > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/tests.log
P="$GRAFT_REPO_ROOT/scripts/probe/k1_probe 64 10000 10 one"
for x in 0 1; do
  TEASER_K1_XCD=$x $P | tee -a $OUT/probe_xcd.jsonl
done
cd /tmp
for x in 0 1; do
  TEASER_K1_XCD=$x timeout 100 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write$x -o t -- $P > $OUT/write$x.log 2>&1
  TEASER_K1_XCD=$x timeout 100 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch$x -o t -- $P > $OUT/fetch$x.log 2>&1
  python $GRAFT_REPO_ROOT/scripts/summarize_pmc.py $(ls $OUT/fetch$x/*counter_collection.csv) $(ls $OUT/write$x/*counter_collection.csv) $OUT/pmc_traffic_xcd$x.json 64 10000 | grep -i "tim_graph"
done
cd $GRAFT_REPO_ROOT
timeout 100 python scripts/profile_scale.py large 2>&1 | grep '^{' | tee $OUT/scale_large.jsonl
timeout 100 python scripts/scale_batch_probe.py 2000 64 2>&1 | tail -1 | tee -a $OUT/scale_batch.jsonl
timeout 100 python scripts/scale_batch_probe.py 800 64 2>&1 | tail -1 | tee -a $OUT/scale_batch.jsonl
