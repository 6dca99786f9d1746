#!/usr/bin/env python3
"""Instruction histogram / register footprint of one kernel from a hipcc -S dump.
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form \
        -Iinclude -Iteaser-plusplus_amd/csrc -S --cuda-device-only -o /tmp/kg.s teaser-plusplus_amd/csrc/kernels_graph.hip
  python scripts/isa_stats.py /tmp/kg.s 21tim_graph_mfma_kernel      # mangled-name fragment after _ZN4thip
Used while tuning K1 (VALU instructions per 1024 pairs, VGPR tier, spills)."""
import re,sys,collections
s=open(sys.argv[1]).read()
name=sys.argv[2]
m=re.search(r'^(_ZN4thip%s[^\n:]*):[^\n]*\n(.*?)\.end_amdhsa_kernel'%name, s, re.S|re.M)
body=m.group(2)
cnt=collections.Counter()
for line in body.split('\n'):
    line=line.strip()
    if not line or line.startswith(';') or line.startswith('.'): continue
    cnt[line.split()[0]]+=1
print(' '.join('%s:%d'%(k,v) for k,v in cnt.most_common(40)))
for key in ['.amdhsa_next_free_vgpr','.amdhsa_accum_offset','.amdhsa_private_segment_fixed_size','.amdhsa_next_free_sgpr']:
    for l in re.findall(r'.*%s.*'%re.escape(key), body)[:1]: print(l.strip())
