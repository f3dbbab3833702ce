#!/bin/bash
# same-box A/B of tools/libray3d_hip_prev.so against ray3d_amd/libray3d_hip.so: bash tools/ab_libs2.sh <batch> [env assignments]
B=$1; shift
for i in 1 2 3; do for lib in tools/libray3d_hip_prev.so ray3d_amd/libray3d_hip.so; do
env "$@" R3D_LIB=$lib python - <<PY
import os,sys,json,io,contextlib
sys.path.insert(0,".")
from ray3d_amd import _capi
_capi.LIB_PATH=os.path.abspath(os.environ["R3D_LIB"])
import bench
sys.argv=["bench.py","--batch","$B","--no-cpu-baseline","--no-b1024","--no-bf16x3","--no-shipped-cfgs","--steps","200","--warmup","20"]
buf=io.StringIO()
with contextlib.redirect_stdout(buf): bench.main()
l=json.loads(buf.getvalue().strip().splitlines()[-1]); print(os.environ["R3D_LIB"], $B, l["ms_per_step"], l["dtype"])
PY
done; done
