#!/bin/bash
# phase stamps of the chained first-level tiles in the staged first-level launch (timing build), B = 256
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for nc in 0 1; do
  echo "== R3D_CHAIN=$((1-nc))"
  R3D_STAGED=1 R3D_CHAIN=$((1-nc)) R3D_LIB_OVERRIDE=tools/libray3d_hip_timing.so R3D_TIMING_STAGE=0 python tools/stage_times.py ${B:-256} 1 2>&1 | grep "timing\|wg " | head -${LINES_:-70}
done > gpurun_out/chain_stamps.txt 2>&1
cat gpurun_out/chain_stamps.txt
