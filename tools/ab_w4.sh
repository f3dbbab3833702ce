#!/bin/bash
# Same-box A/B of the four-wave kernels (R3D_W4=1) against the eight-wave product path, with the parity subset first.
# Through gpurun from the repo root: bash tools/ab_w4.sh > gpurun_out/ab_w4.txt
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
R3D_W4=1 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "reference_fixture or window_counts or every_window" 2>&1 | tail -15
for B in 256 1024; do
  bash tools/ab_env.sh $B "R3D_W4=0" "R3D_W4=1" "R3D_W4=0 R3D_STAGED=1" "R3D_W4=1 R3D_STAGED=1"
done
