// r3d_forward_b3 / r3d_forward_uv_b3: the single-launch forward + the bf16x3 tiles (r3d_config.bf16x3, calls of >= 96 windows).
// One of the kernel translation units (r3d_tiles.hpp holds the tile code; r3d_kernels.hip the launchers that pick a kernel).
#include "r3d_tiles.hpp"

namespace r3d {

// The whole forward in one launch: every level's tiles, ordered by ready counters (wait_deps).  One workgroup per CU, all of
// them resident (grid <= CU count: a waiting workgroup can only wait for tiles of resident workgroups or of its own past).
#define R3D_FORWARD_KERNEL(name, UV_, B3_, NARROW_, CLIP_)                                              \
    extern "C" __global__ __launch_bounds__(GEMM_THREADS) void name(const FwdArgs args_) {             \
        extern __shared__ __attribute__((aligned(16))) float smem[];                                    \
        (void)args_;                                                                                    \
        gemm_persistent<false, UV_, true, B3_, NARROW_, CLIP_>(smem);                                   \
    }
R3D_FORWARD_KERNEL(r3d_forward_b3, false, true, false, false)
R3D_FORWARD_KERNEL(r3d_forward_uv_b3, true, true, false, false)
FwdKernel fwd_kernel_b3(bool uv) { return uv ? r3d_forward_uv_b3 : r3d_forward_b3; }

}  // namespace r3d
