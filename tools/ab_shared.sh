#!/bin/bash
# clip-mode parity tests, then the eval line (BASELINE configs[2] shape) with and without the per-frame first layers
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "clip or eval or h36m or flip" 2>&1 | tail -6
for cfg in "R3D_NO_SHARED_L0=0" "R3D_NO_SHARED_L0=1" "R3D_NO_SHARED_L0=0" "R3D_NO_SHARED_L0=1"; do
  env R3D_USE_HOOKS_LIB=1 $cfg python bench.py --mode eval 2>/dev/null | python -c "
import sys,json
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', l['value'], l['ms_per_step'], l['mpjpe_mm']['checksum'])"
done
bash tools/ab_env.sh 256 "R3D_X=0"
bash tools/ab_env.sh 1024 "R3D_X=0"
