#!/bin/bash
# same-box A/B of variant libraries (tools/variant_lib.sh) against the hooks build: bench.py (parity gate first) at 256 / 1024 windows, twice,
# and the eval pass once per library.   usage: bash tools/variant_ab.sh "<name> ..." [eval]
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
one() {
  local name=$1; shift
  for B in 256 1024; do
    env R3D_USE_HOOKS_LIB=1 "$@" python bench.py --batch $B --no-cpu-baseline --no-bf16x3 --no-shipped-cfgs --no-b1024 --no-c1024 --steps 200 --warmup 10 2>/dev/null | python -c "
import sys,json
t=sys.stdin.read().strip().splitlines()
try:
    l=json.loads(t[-1]); print('$name', $B, 'ms', l['ms_per_step'], 'parity', l['parity_max_abs_err'], 'frac', l['roofline']['frac'])
except Exception as e: print('$name', $B, 'FAILED', e, t[-1:])"
  done
  if [ "$EVAL" = "eval" ]; then
    env R3D_USE_HOOKS_LIB=1 "$@" python bench.py --mode eval 2>/dev/null | python -c "
import sys,json
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', 'eval', l['value'], l['ms_per_step'], l['roofline']['frac'])"
  fi
}
EVAL=$2
for rep in 1 2; do
  one base
  for v in $1; do one $v R3D_HOOKS_LIB=$PWD/tools/libray3d_hip_$v.so; done
done 2>&1 | tee gpurun_out/variant_ab_$(echo $1 | tr ' ' '_').txt
