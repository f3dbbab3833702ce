// r3d_forward_f32 / r3d_forward_uv_f32: the single-launch forward with the fp32 throughput tiles only (the headline kernel).
// One of the kernel translation units (r3d_tiles.hpp holds the tile code; r3d_kernels.hip the launchers that pick a kernel).
#include "r3d_tiles.hpp"

namespace r3d {

// The whole forward in one launch: every level's tiles, ordered by ready counters (wait_deps).  One workgroup per CU, all of
// them resident (grid <= CU count: a waiting workgroup can only wait for tiles of resident workgroups or of its own past).
#define R3D_FORWARD_KERNEL(name, UV_, B3_, NARROW_, CLIP_, CHAIN_)                                            \
    extern "C" __global__ __launch_bounds__(GEMM_THREADS) void name(const FwdArgs args_) {             \
        extern __shared__ __attribute__((aligned(16))) float smem[];                                    \
        (void)args_;                                                                                    \
        gemm_persistent<false, UV_, true, B3_, NARROW_, CLIP_, CHAIN_>(smem);                       \
    }
R3D_FORWARD_KERNEL(r3d_forward_f32, false, false, false, false, false)
R3D_FORWARD_KERNEL(r3d_forward_uv_f32, true, false, false, false, false)
FwdKernel fwd_kernel_f32(bool uv) { return uv ? r3d_forward_uv_f32 : r3d_forward_f32; }

}  // namespace r3d
