#!/bin/bash
# parity subset of the GPU suite on a variant library (tools/variant_lib.sh), then the A/B: bash tools/variant_check.sh "<name> ..." [eval]
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for v in $1; do
  R3D_USE_HOOKS_LIB=1 R3D_HOOKS_LIB=$PWD/tools/libray3d_hip_$v.so timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -x \
    -k "lifter_pair_matches or reference_fixture_in_every_mode or full_size_batch or ragged or large_batch_1024 or cfg4 or abort or two_processes or schedule_invariance or graph" > gpurun_out/variant_check_$v.log 2>&1
  echo "== $v: $(grep -v '^parity' gpurun_out/variant_check_$v.log | tail -3 | tr '\n' ' ')"
done
bash tools/variant_ab.sh "$1" $2
