#!/bin/bash
# A variant of the hooks library with ONE kernel family recompiled under extra flags (same-box A/Bs of tile experiments):
#   bash tools/variant_lib.sh <name> <family> "<flags>"      e.g.  bash tools/variant_lib.sh pin r3d_k_fwd_f32 "-DR3D_PIN_PREFETCH"
# -> tools/libray3d_hip_<name>.so; run with R3D_USE_HOOKS_LIB=1 R3D_HOOKS_LIB=$PWD/tools/libray3d_hip_<name>.so (tools/variant_ab.sh)
set -e
R=$(cd $(dirname $0)/.. && pwd); C=$R/ray3d_amd/csrc; B=$C/build
name=$1; fam=$2; flags=$3
(cd $C && make -j8 >/dev/null)
objs=""
for f in r3d_kernels r3d_metrics r3d_k_gemm r3d_k_gemm_enc r3d_k_gemm_b3 r3d_k_fwd_f32 r3d_k_fwd_b3 r3d_k_fwd_lat r3d_k_fwd_clip r3d_k_fwd_chain; do
  if [[ " $fam " == *" $f "* ]]; then
    (cd $C && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$R/include -I. -Wall -Wno-unused-result $flags -Rpass-analysis=kernel-resource-usage \
       -x hip -c -o $B/$f.$name.o $f.hip 2>&1 | grep -i "Function Name\|VGPRs Spill\|ScratchSize" | sed "s/^.*remark: [^ ]* *//" | paste - - - | sed "s/^/$name: /") &
    objs="$objs $B/$f.$name.o"
  else
    objs="$objs $B/$f.o"
  fi
done
wait
/opt/rocm/bin/hipcc -fPIC --offload-arch=gfx950 -shared -o $R/tools/libray3d_hip_$name.so $objs $B/r3d_model.hooks.o $B/r3d_plan.hooks.o $B/r3d_schedule.hooks.o $B/r3d_api.hooks.o
ls -la $R/tools/libray3d_hip_$name.so
