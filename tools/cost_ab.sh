#!/bin/bash
# same-box A/B of a set of packer cost constants against the defaults, through bench.py on the hooks build:
#   bash tools/cost_ab.sh "<R3D_COST string>" [rounds]     (256 and 1024 windows every round, the eval pass in the first)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
C="$1"; N=${2:-4}
line() { python -c "
import sys,json
l=json.loads([x for x in sys.stdin.read().strip().splitlines() if x.startswith('{')][-1]); print('$1', '$2', l['ms_per_step'], l.get('parity_max_abs_err'), l['roofline']['frac'])"; }
for i in $(seq 1 $N); do
  for B in 256 1024; do
    env R3D_USE_HOOKS_LIB=1 python bench.py --batch $B --no-cpu-baseline --no-bf16x3 --no-shipped-cfgs --no-b1024 --no-c1024 --steps 200 --warmup 10 2>/dev/null | line base $B
    env R3D_USE_HOOKS_LIB=1 R3D_COST="$C" python bench.py --batch $B --no-cpu-baseline --no-bf16x3 --no-shipped-cfgs --no-b1024 --no-c1024 --steps 200 --warmup 10 2>/dev/null | line cost $B
  done
  if [ $i = 1 ]; then
    env R3D_USE_HOOKS_LIB=1 python bench.py --mode eval 2>/dev/null | line base eval
    env R3D_USE_HOOKS_LIB=1 R3D_COST="$C" python bench.py --mode eval 2>/dev/null | line cost eval
  fi
done | tee gpurun_out/cost_ab.txt
