#!/bin/bash
# Slope (time per K tile) and intercept (fixed cost) of an M = B launch: 7 problems of 256 x 1024 x K, K = 512 ... 4096,
# for the eight-wave tiles, the four-wave tiles (two workgroups per CU) and the four-wave tiles one per CU (64 x 128).
cd ${GRAFT_REPO_ROOT:-/root/repo}
for mode in ""; do   # (round 4 ran it for the four-wave tiles too: profiles/r04_w4_ab/kslope.txt, commit a436808)
  for K in 512 1024 2048 4096; do
    echo -n "[$mode] K=$K: "; env $mode ./tools/gemm_probe.bin 7 256 1024 $K 30 | grep "best"
  done
done
