#!/bin/bash
# builds /tmp-free in-tree probe binaries (they travel to the GPU box with the snapshot)
set -e
cd "$(dirname "$0")/.."
F="-O3 -std=c++17 --offload-arch=gfx950 -Iinclude -Iray3d_amd/csrc -Wno-unused-result"
S="tools/gemm_probe.cpp ray3d_amd/csrc/r3d_kernels.hip ray3d_amd/csrc/r3d_k_gemm.hip ray3d_amd/csrc/r3d_k_gemm_enc.hip ray3d_amd/csrc/r3d_k_gemm_b3.hip ray3d_amd/csrc/r3d_k_fwd_f32.hip ray3d_amd/csrc/r3d_k_fwd_b3.hip ray3d_amd/csrc/r3d_k_fwd_lat.hip ray3d_amd/csrc/r3d_k_fwd_clip.hip ray3d_amd/csrc/r3d_metrics.hip ray3d_amd/csrc/r3d_schedule.cpp ray3d_amd/csrc/r3d_model.cpp ray3d_amd/csrc/r3d_plan.cpp ray3d_amd/csrc/r3d_api.cpp"
/opt/rocm/bin/hipcc $F -x hip $S -o tools/gemm_probe.bin
/opt/rocm/bin/hipcc $F -DR3D_TIMING -x hip $S -o tools/gemm_probe_timing.bin
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/bf16x3_probe.cpp -o tools/bf16x3_probe.bin
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/coexec_probe.cpp -o tools/coexec_probe.bin
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/coexec_probe2.cpp -o tools/coexec_probe2.bin
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/coexec_probe3.cpp -o tools/coexec_probe3.bin
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/mfma_f32_probe.cpp -o tools/mfma_f32_probe.bin
# the library with phase stamps (R3D_TIMING_STAGE=<launch>|all R3D_LIB_OVERRIDE=tools/libray3d_hip_timing.so python tools/stage_times.py;
# -DR3D_TS=1 / 2 moves a first-level tile's fine stamps from its first tap phase to the second / third)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DR3D_TIMING -Iinclude -Iray3d_amd/csrc -Wno-unused-result -x hip -shared \
  -o tools/libray3d_hip_timing.so ray3d_amd/csrc/r3d_kernels.hip ray3d_amd/csrc/r3d_k_gemm.hip ray3d_amd/csrc/r3d_k_gemm_enc.hip ray3d_amd/csrc/r3d_k_gemm_b3.hip ray3d_amd/csrc/r3d_k_fwd_f32.hip ray3d_amd/csrc/r3d_k_fwd_b3.hip ray3d_amd/csrc/r3d_k_fwd_lat.hip ray3d_amd/csrc/r3d_k_fwd_clip.hip ray3d_amd/csrc/r3d_metrics.hip ray3d_amd/csrc/r3d_model.cpp \
  ray3d_amd/csrc/r3d_plan.cpp ray3d_amd/csrc/r3d_schedule.cpp ray3d_amd/csrc/r3d_api.cpp
