#!/bin/bash
# build_full.sh <name> <extra hipcc flags...>: the WHOLE product library with extra flags -> gpurun_ab/<name>/libteaser_hip.so
# (for A/B of compile-time constants that live in internal.h, e.g. -DTEASER_TAIL_PRIO=1)
set -e
NAME=$1; shift
C=/root/repo/teaser-plusplus_amd/csrc; O=/root/repo/gpurun_ab/$NAME; mkdir -p $O
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -I/root/repo/include -I$C -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form $*"
for f in kernels_graph kernels_heuristic kernels_clique kernels_estimate kernels_scale kernels_features kernels_certify comm solver; do
  extra=""; [ $f = kernels_graph ] && extra="-fno-slp-vectorize"
  /opt/rocm/bin/hipcc $FLAGS $extra -c $C/$f.hip -o $O/$f.o > $O/$f.log 2>&1 &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/libteaser_hip.so $O/*.o $C/synth.o -ldl -Wl,-rpath,/opt/rocm/lib
rm -f $O/*.o; ls -la $O/libteaser_hip.so
