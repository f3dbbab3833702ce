#!/bin/bash
# per-tile timelines of the 256- and 1024-window forward with and without the register-chained first-level tile (timing build)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for B in ${BATCHES:-256}; do
  for nc in 0 1; do
    R3D_CHAIN=$((1-nc)) R3D_LIB_OVERRIDE=tools/libray3d_hip_timing.so R3D_TIMING_STAGE=all R3D_TIMING_DUMP=gpurun_out/gantt_${B}_nc$nc.txt python tools/stage_times.py $B 1 > gpurun_out/gantt_${B}_nc$nc.log 2>&1
    echo "== B=$B R3D_CHAIN=$((1-nc))"; python tools/fwd_gantt.py gpurun_out/gantt_${B}_nc$nc.txt 2>&1 | tee gpurun_out/fwd_gantt_${B}_nc$nc.txt | grep -v "^FuseBlocks\|^GlobalInfo\|^Integration\|^embedder\|^trj.Integration"
  done
done
python tools/chain_runs.py gpurun_out/gantt_*_nc0.txt gpurun_out/gantt_*_nc1.txt
