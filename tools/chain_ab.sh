#!/bin/bash
# GPU check of the register-chained first-level tile: parity subset, then the same-box A/B (hooks build: R3D_NO_CHAIN=1 = first_level_taps)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
R3D_F32_BUDGET=${BUDGET:-1.5} timeout 1800 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "${1:-lifter_pair_matches or reference_fixture_in_every_mode or full_size_batch or f32_error_budget or ragged or large_batch_1024}" > gpurun_out/chain_parity.log 2>&1
grep -v "^parity test_\|^parity RELAXED" gpurun_out/chain_parity.log | tail -70
for B in 256 1024; do
  bash tools/ab_env.sh $B "R3D_CHAIN=1" "R3D_CHAIN=0" "R3D_CHAIN=1" "R3D_CHAIN=0"
done 2>&1 | tee gpurun_out/chain_ab.txt
