#!/bin/bash
# same-box A/B of the chained tile's variants (tools/chain_variants.sh) inside the forward: bench.py (parity gate first) at 256 / 1024
# windows and the pos model alone; the product's first_level_taps beside them.   usage: bash tools/chain_variants_ab.sh "v0 v1 v2 ..."
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
run() {   # name, env...
  local name=$1; shift
  for B in 256 1024; do
    env R3D_USE_HOOKS_LIB=1 "$@" python bench.py --batch $B --no-cpu-baseline --no-bf16x3 --no-shipped-cfgs --no-b1024 --no-c1024 --steps 200 --warmup 10 2>/dev/null | python -c "
import sys,json
t=sys.stdin.read().strip().splitlines()
try:
    l=json.loads(t[-1]); print('$name', $B, 'pair ms', l['ms_per_step'], 'parity', l['parity_max_abs_err'], 'kernel', l['roofline']['kernel'])
except Exception as e: print('$name', $B, 'FAILED', e, t[-1:] )"
  done
  env R3D_USE_HOOKS_LIB=1 "$@" python tools/chain_pos_only.py 1024 200 2>&1 | grep "^CHAIN" | sed "s/^/$name /"
}
for rep in 1 2; do
  run taps R3D_CHAIN=0
  for v in ${1:-v0 v1 v2}; do run $v R3D_CHAIN=1 R3D_HOOKS_LIB=$PWD/tools/libray3d_hip_chain_$v.so; done
done 2>&1 | tee gpurun_out/chain_variants_ab.txt
