// r3d_gemm_f32 / r3d_gemm_uv_f32: one launch per level of the plan (the level-by-level form; per-frame first layers of clip calls).
// One of the kernel translation units (r3d_tiles.hpp holds the tile code; r3d_kernels.hip the launchers that pick a kernel).
#include "r3d_tiles.hpp"

namespace r3d {

// every layer whose input is an activation matrix in HBM, plus the gathered first layers: one workgroup per CU
// (the fp32 tiles only: the bf16x3 tile kinds live in r3d_gemm_b3)
extern "C" __global__ __launch_bounds__(GEMM_THREADS) void r3d_gemm_f32(const LaunchArgs args_) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    (void)args_;
    gemm_persistent<false, false, false, false>(smem);
}
// the same for launches whose gathered operands are pixel keypoints (UV input mode: rays encoded while staging)
extern "C" __global__ __launch_bounds__(GEMM_THREADS) void r3d_gemm_uv_f32(const LaunchArgs args_) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    (void)args_;
    gemm_persistent<false, true, false, false>(smem);
}
GemmKernel gemm_kernel_f32(bool uv) { return uv ? r3d_gemm_uv_f32 : r3d_gemm_f32; }

}  // namespace r3d
