#!/bin/bash
# SQ instruction counters of K1 variants (one rocprofv3 --pmc run per variant; no trace flags beside)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3k
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3k
export TMPDIR=/tmp
cd /tmp
for v in 1 8 13; do
  TEASER_K1_VARIANT=$v timeout 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_v$v -o t -- $GRAFT_REPO_ROOT/scripts/probe/k1_probe 64 10000 3 one > $OUT/pmc_v$v.log 2>&1
  echo "v=$v rc=$?"
done
python - <<'PY'
import csv,glob,os,collections
OUT=os.environ.get('OUT') or '/root/repo/gpurun_out/r3k'
for v in (1,8,13):
    fs=glob.glob('%s/pmc_v%d/*counter_collection.csv'%(OUT,v))
    if not fs: print(v,'no csv'); continue
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if 'tim_graph_mfma' in r['Kernel_Name'] and int(r['Grid_Size'])>1000000:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    print(v,{k:round(sum(x)/len(x)/1e6,2) for k,x in agg.items()}, 'launches',len(next(iter(agg.values()),[])))
PY
