// r3d_gemm_enc_f32 / r3d_gemm_enc_uv_f32: first layers with the input encoding fused in, where first_level_taps is not used.
// One of the kernel translation units (r3d_tiles.hpp holds the tile code; r3d_kernels.hip the launchers that pick a kernel).
#include "r3d_tiles.hpp"

namespace r3d {

// two workgroups per CU (4 wavefronts per SIMD)
extern "C" __global__ __launch_bounds__(GEMM_THREADS, 4) void r3d_gemm_enc_f32(const LaunchArgs args_) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    (void)args_;
    gemm_persistent<true, false, false, false, false>(smem);
}
extern "C" __global__ __launch_bounds__(GEMM_THREADS, 4) void r3d_gemm_enc_uv_f32(const LaunchArgs args_) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    (void)args_;
    gemm_persistent<true, true, false, false, false>(smem);
}
GemmKernel gemm_kernel_enc(bool uv) { return uv ? r3d_gemm_enc_uv_f32 : r3d_gemm_enc_f32; }

}  // namespace r3d
