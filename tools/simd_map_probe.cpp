// Which SIMD does wavefront w of a 512-thread workgroup run on?  (gemm_tile_nb deals whole column blocks to wavefronts 0-3
// and quarter blocks to 4-7, assuming w and w + 4 share a SIMD.)  Prints HW_ID's SIMD / CU fields per wavefront for a few
// workgroups, and the time of an MFMA loop in which wavefronts 0-3 issue 16 and wavefronts 4-7 issue `x` MFMAs per round.
// build: hipcc -O3 --offload-arch=gfx950 tools/simd_map_probe.cpp -o tools/simd_map_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(512) void who(unsigned *out) {
    extern __shared__ float lds[];
    const unsigned id = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));     // HW_REG_HW_ID
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = id;
    lds[threadIdx.x] = 0;
}
template <int EXTRA>
__global__ __launch_bounds__(512) void mix(float *out, int rounds, int swap) {
    extern __shared__ float lds[];
    const int wave = threadIdx.x >> 6;
    const bool main_w = swap ? (wave & 1) == 0 : wave < 4;
    f32x16 acc = {};
    float a = threadIdx.x * 1e-3f, b = 1.0f;
    const int n4 = __builtin_amdgcn_readfirstlane(main_w ? 4 : EXTRA / 4);     // (scalar: groups of four MFMAs, as the tile's slots)
    for (int r = 0; r < rounds; ++r) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (g >= n4) break;
#pragma unroll
            for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        __syncthreads();
    }
    out[blockIdx.x * 512 + threadIdx.x] = acc[0] + lds[0];
}
int main() {
    unsigned *d; hipMalloc(&d, 256 * 8 * 4);
    who<<<256, 512, 150000>>>(d);
    std::vector<unsigned> h(256 * 8);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    for (int b = 0; b < 4; ++b) {
        printf("workgroup %d:", b);
        for (int w = 0; w < 8; ++w) printf("  w%d simd %u cu %u", w, (h[b * 8 + w] >> 4) & 3, (h[b * 8 + w] >> 8) & 15);
        printf("\n");
    }
    int same = 0;
    for (int b = 0; b < 256; ++b) { bool ok = true; for (int w = 0; w < 4; ++w) ok = ok && (((h[b*8+w] >> 4) & 3) == ((h[b*8+w+4] >> 4) & 3)); same += ok; }
    printf("workgroups in which wavefronts w and w+4 share a SIMD for all w: %d of 256\n", same);
    float *o; hipMalloc(&o, 256 * 512 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](auto kern, const char *name, int swap) {
        kern<<<256, 512, 150000>>>(o, 2000, swap); hipDeviceSynchronize();
        hipEventRecord(e0); kern<<<256, 512, 150000>>>(o, 20000, swap); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-40s swap %d: %.3f ms for 20000 rounds = %.1f ns per round\n", name, swap, ms, ms * 1e6 / 20000);
    };
    for (int swap = 0; swap < 2; ++swap) {
        run(mix<16>, "16 + 16 MFMAs (whole tile)", swap);
        run(mix<12>, "16 + 12", swap);
        run(mix<8>, "16 + 8", swap);
        run(mix<4>, "16 + 4", swap);
        run(mix<0>, "16 + 0", swap);
    }
    return 0;
}
