#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3h
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3h
export TMPDIR=/tmp
cd /tmp
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/c3 -o t -- python $GRAFT_REPO_ROOT/scripts/profile_stages.py big > $OUT/c3.log 2>&1; tail -1 $OUT/c3.log | cut -c1-400
python - <<'PY'
import csv,glob,os
f=glob.glob(os.environ.get('OUT','/root/repo/gpurun_out/r3h')+'/c3/*kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
ev=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"].split("(")[0].replace("void ","").replace("thip::","")) for r in rows]
ev.sort()
# last solve: from the last tim_graph_mfma start
k1=[i for i,e in enumerate(ev) if e[2].startswith("tim_graph_mfma")]
i0=k1[-1]
t0=ev[i0][0]
for e in ev[i0-3:]:
    print("%9.1f %8.1f %s"%((e[0]-t0)/1e3,(e[1]-e[0])/1e3,e[2][:40]))
PY
