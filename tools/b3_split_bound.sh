#!/bin/bash
# What would pre-split operands buy the bf16x3 mode?  Upper bounds, measured: the library built with the split of the WEIGHTS
# (R3D_EXP_NOSPLIT_W), of the ACTIVATIONS (R3D_EXP_NOSPLIT_A) or of both compiled out (all three planes = the leading bf16 term:
# same loads, same LDS traffic, same MFMAs, no split arithmetic - the results are wrong, only the time means anything).
#   build (here, no GPU needed):   bash tools/b3_split_bound.sh build
#   measure (through gpurun):      bash tools/b3_split_bound.sh run  > gpurun_out/b3_split_bound.txt
cd "$(dirname "$0")/.."
SRC="ray3d_amd/csrc/r3d_kernels.hip ray3d_amd/csrc/r3d_k_gemm.hip ray3d_amd/csrc/r3d_k_gemm_enc.hip ray3d_amd/csrc/r3d_k_gemm_b3.hip ray3d_amd/csrc/r3d_k_fwd_f32.hip ray3d_amd/csrc/r3d_k_fwd_b3.hip ray3d_amd/csrc/r3d_k_fwd_lat.hip ray3d_amd/csrc/r3d_k_fwd_clip.hip ray3d_amd/csrc/r3d_metrics.hip ray3d_amd/csrc/r3d_model.cpp ray3d_amd/csrc/r3d_plan.cpp ray3d_amd/csrc/r3d_schedule.cpp ray3d_amd/csrc/r3d_api.cpp"
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -Iray3d_amd/csrc -Wno-unused-result -x hip -shared"
if [ "$1" = build ]; then
  /opt/rocm/bin/hipcc $F -DR3D_EXP_NOSPLIT_W -o tools/libray3d_hip_exp_w.so $SRC &
  /opt/rocm/bin/hipcc $F -DR3D_EXP_NOSPLIT_A -o tools/libray3d_hip_exp_a.so $SRC &
  /opt/rocm/bin/hipcc $F -DR3D_EXP_NOSPLIT_W -DR3D_EXP_NOSPLIT_A -o tools/libray3d_hip_exp_wa.so $SRC &
  wait
  exit 0
fi
for B in 256 1024; do for i in 1 2; do for lib in ray3d_amd/libray3d_hip.so tools/libray3d_hip_exp_a.so tools/libray3d_hip_exp_w.so tools/libray3d_hip_exp_wa.so; do
echo "B=$B $lib $(R3D_BF16X3=1 R3D_LIB=$lib python - <<PY
import os,sys,json,io,contextlib
sys.path.insert(0,".")
from ray3d_amd import _capi
_capi.LIB_PATH=os.path.abspath(os.environ["R3D_LIB"])
import bench
bench.parity_gate=lambda *a, **k: (0.0, 1.0)      # (the experiment builds compute wrong values on purpose)
sys.argv=["bench.py","--batch","$B","--no-cpu-baseline","--no-b1024","--no-bf16x3","--no-shipped-cfgs","--steps","200","--warmup","20"]
buf=io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
d=json.loads(buf.getvalue().strip().splitlines()[-1])
print(d["dtype"], d["ms_per_step"], (d.get("roofline") or {}).get("kernel"))
PY
)"; done; done; done
