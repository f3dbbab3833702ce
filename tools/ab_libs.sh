#!/bin/bash
# same-box A/B of two builds of the library through bench.py (3 alternating runs of 300 steps each):
#   cp ray3d_amd/libray3d_hip.so tools/libray3d_hip_prev.so   (before the change), rebuild, then through gpurun:
#   [R3D_BF16X3=1] bash tools/ab_libs.sh
for i in 1 2 3; do for lib in tools/libray3d_hip_prev.so ray3d_amd/libray3d_hip.so; do echo "$lib $(python - <<PY
import os,sys
sys.path.insert(0,".")
from ray3d_amd import _capi
_capi.LIB_PATH=os.path.abspath("$lib")
import bench
sys.argv=["bench.py","--no-cpu-baseline","--no-b1024","--no-bf16x3","--steps","300","--warmup","30"]
bench.main()
PY
)" | python -c "
import sys,json
l=sys.stdin.read(); i=l.index('{'); d=json.loads(l[i:].strip().splitlines()[-1]); print(l[:i].strip(), d['ms_per_step'])"; done; done
