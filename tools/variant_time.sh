#!/bin/bash
# forward times (no parity gate: for ablation variants with wrong results) of variant libraries: bash tools/variant_time.sh "<name> ..."
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for rep in 1 2; do
  for v in base $1; do
    for B in 256 1024; do
      if [ $v = base ]; then R3D_USE_HOOKS_LIB=1 python tools/chain_pos_only.py $B 200 2>&1 | grep "^CHAIN" | sed "s/^CHAIN=[^ ]*/$v/"
      else R3D_USE_HOOKS_LIB=1 R3D_HOOKS_LIB=$PWD/tools/libray3d_hip_$v.so python tools/chain_pos_only.py $B 200 2>&1 | grep "^CHAIN" | sed "s/^CHAIN=[^ ]*/$v/"; fi
    done
  done
done | cut -c1-100 | tee gpurun_out/variant_time_$(echo $1 | tr ' ' '_').txt
