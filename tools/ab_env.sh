#!/bin/bash
# same-box A/B of environment switches through bench.py: bash tools/ab_env.sh <batch> "<VAR=val ...>" "<VAR=val ...>" ...
B=$1; shift
for cfg in "$@"; do
  for i in 1 2; do
    env R3D_USE_HOOKS_LIB=1 $cfg python bench.py --batch $B --no-cpu-baseline --no-bf16x3 --no-shipped-cfgs --no-b1024 --no-c1024 --steps 200 --warmup 10 2>/dev/null | python -c "
import sys,json
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', $B, l['ms_per_step'], l['parity_max_abs_err'])"
  done
done
