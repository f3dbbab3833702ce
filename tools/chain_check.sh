#!/bin/bash
# quick check of the chained tile: a few parity tests, phase stamps, A/B at 256 / 1024 windows
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "lifter_pair_matches or full_size_batch or ragged or (reference_fixture_in_every_mode and f32)" 2>&1 | grep -v "^parity" | tail -8
LINES_=14 bash tools/chain_stamps.sh | grep -v "^\[timing\] launch 0: \(first\|shader\|wg tile\)"
for B in 256 1024; do
  bash tools/ab_env.sh $B "R3D_CHAIN=1" "R3D_CHAIN=0" "R3D_CHAIN=1" "R3D_CHAIN=0"
done 2>&1 | tee gpurun_out/chain_ab.txt
