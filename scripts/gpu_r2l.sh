#!/bin/bash
# session L (final): the whole GPU suite incl. the certifier, then the bench line
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2l
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2l
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
timeout 400 python -m pytest tests -m gpu -q --timeout=200 > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/tests.log
timeout 120 python bench.py --no-cpu-baseline --no-host-resident > $OUT/bench.log 2>&1; tail -1 $OUT/bench.log | cut -c1-330
