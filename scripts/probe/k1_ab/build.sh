#!/bin/bash
# build.sh <variant> ... : gpurun_ab/kernels_graph_<v>.hip -> gpurun_ab/<v>/libteaser_hip.so
C=/root/repo/teaser-plusplus_amd/csrc
mkdir -p /root/repo/gpurun_ab; cd /root/repo/gpurun_ab
for v in "$@"; do
  mkdir -p $v
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -I/root/repo/include -I$C -Wno-unused-function -Wno-unused-variable -Wno-unused-but-set-variable -mllvm -amdgpu-mfma-vgpr-form -fno-slp-vectorize -Rpass-analysis=kernel-resource-usage -c kernels_graph_$v.hip -o $v/kg.o > $v/build.log 2>&1 || { echo "$v: COMPILE FAILED"; grep -m5 error $v/build.log; rm -f $v/libteaser_hip.so; exit 1; }
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $v/libteaser_hip.so $v/kg.o $C/kernels_heuristic.o $C/kernels_clique.o $C/kernels_estimate.o $C/kernels_scale.o $C/kernels_features.o $C/kernels_certify.o $C/comm.o $C/solver.o $C/synth.o -ldl -Wl,-rpath,/opt/rocm/lib
    echo "$v: $(grep -A12 'tim_graph_mfma3_kernel' $v/build.log | grep -E 'VGPRs:|ScratchSize|Occupancy' | tr -s ' ' | tr '\n' ' ')" ) &
done
wait
