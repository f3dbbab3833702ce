#!/bin/bash
# Variants of the register-chained first-level tile for in-situ A/Bs: r3d_k_fwd_chain.hip compiled with
# -DR3D_CHAIN_V=<v> -DR3D_CHAIN_ABL=<bits> and linked with the hooks build's other objects into tools/libray3d_hip_chain_<name>.so
# (load with R3D_LIB_OVERRIDE=...; R3D_CHAIN=1 selects the kernel).  usage: bash tools/chain_variants.sh "v0:0:0 v1:1:0 v2:2:0 v2l:2:1 ..."
set -e
R=$(cd $(dirname $0)/.. && pwd)
C=$R/ray3d_amd/csrc
B=$C/build
(cd $C && make -j8 >/dev/null)
OTHERS="$B/r3d_kernels.o $B/r3d_metrics.o $B/r3d_k_gemm.o $B/r3d_k_gemm_enc.o $B/r3d_k_gemm_b3.o $B/r3d_k_fwd_f32.o $B/r3d_k_fwd_b3.o $B/r3d_k_fwd_lat.o $B/r3d_k_fwd_clip.o $B/r3d_model.hooks.o $B/r3d_plan.hooks.o $B/r3d_schedule.hooks.o $B/r3d_api.hooks.o"
for spec in ${1:-v0:0:0 v1:1:0 v2:2:0}; do
  IFS=: read name v abl <<< "$spec"
  (cd $C && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$R/include -I. -Wall -Wno-unused-result -DR3D_CHAIN_V=$v -DR3D_CHAIN_ABL=$abl \
     -Rpass-analysis=kernel-resource-usage -x hip -c -o $B/r3d_k_fwd_chain.$name.o r3d_k_fwd_chain.hip 2>&1 | grep -i "VGPRs Spill\|ScratchSize\|error" | sed "s/^/$name: /" ; \
   /opt/rocm/bin/hipcc -fPIC --offload-arch=gfx950 -shared -o $R/tools/libray3d_hip_chain_$name.so $OTHERS $B/r3d_k_fwd_chain.$name.o) &
done
wait
ls -la $R/tools/libray3d_hip_chain_*.so
