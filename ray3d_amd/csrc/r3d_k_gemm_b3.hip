// r3d_gemm_b3 / r3d_gemm_uv_b3: the level launches of handles in bf16x3 mode (some problem carries GemmProb::wb3).
// One of the kernel translation units (r3d_tiles.hpp holds the tile code; r3d_kernels.hip the launchers that pick a kernel).
#include "r3d_tiles.hpp"

namespace r3d {

extern "C" __global__ __launch_bounds__(GEMM_THREADS) void r3d_gemm_b3(const LaunchArgs args_) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    (void)args_;
    gemm_persistent<false, false, false, true>(smem);
}
extern "C" __global__ __launch_bounds__(GEMM_THREADS) void r3d_gemm_uv_b3(const LaunchArgs args_) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    (void)args_;
    gemm_persistent<false, true, false, true>(smem);
}
GemmKernel gemm_kernel_b3(bool uv) { return uv ? r3d_gemm_uv_b3 : r3d_gemm_b3; }

}  // namespace r3d
