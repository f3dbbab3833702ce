// r3d_forward_chain_f32: r3d_forward_f32 + the register-chained first-level tile (r3d_chain.hpp) - an EXPERIMENT kept for its A/B:
// the tile is 10 % faster than first_level_taps stand-alone (tools/chain_probe4.cpp: 71 - 73 us per 64 rows against ~79) and 5 % slower
// inside the persistent forward.  Only handles of the hooks build with R3D_CHAIN=1 select it (fill_prob sets
// GemmProb::wchain); the product's kernels do not carry the tile.
#include "r3d_tiles.hpp"

namespace r3d {

extern "C" __global__ __launch_bounds__(GEMM_THREADS) void r3d_forward_chain_f32(const FwdArgs args_) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    (void)args_;
    gemm_persistent<false, false, true, false, false, false, true>(smem);
}
FwdKernel fwd_kernel_chain(bool) { return r3d_forward_chain_f32; }

}  // namespace r3d
