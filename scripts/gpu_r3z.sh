#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3z
export OUT=$GRAFT_REPO_ROOT/gpurun_out/r3z
TEASER_K4_DEBUG=1 timeout 100 python scripts/profile_config5.py 2>&1 | grep "exact search" | tail -4
TEASER_K4_DEBUG=1 timeout 300 python bench.py --configs 5 --no-cpu-baseline --no-latency --no-host-resident --steps 4 --warmup 1 --repeats 1 > $OUT/b5.log 2>&1
grep "exact search: [0-9]* problems" $OUT/b5.log | tail -2
grep "exact search: problem" $OUT/b5.log | tail -64 | awk '{for(i=1;i<=NF;i++) if($i=="nodes") print $(i+1)}' | sort -n | awk '{a[NR]=$1; s+=$1} END {print "nodes per problem: min",a[1],"med",a[int(NR/2)],"max",a[NR],"sum",s,"count",NR}'
grep '^{' $OUT/b5.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(json.dumps(d['config'].get('configs', d['config']))[:900])"
