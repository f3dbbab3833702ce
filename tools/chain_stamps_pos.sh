#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for nc in 0 1; do
  echo "== pos only, R3D_NO_CHAIN=$nc"
  R3D_STAGED=1 R3D_NO_CHAIN=$nc R3D_LIB_OVERRIDE=tools/libray3d_hip_timing.so R3D_TIMING_STAGE=0 python tools/stage_times_pos.py 256 2>&1 | tail -40
done > gpurun_out/chain_stamps_pos.txt 2>&1
cat gpurun_out/chain_stamps_pos.txt
