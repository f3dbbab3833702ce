#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int MODE> __global__ __launch_bounds__(512) void k(float *out, int rounds) {
    float a = threadIdx.x * 1e-3f, b = 1.0f;
    if (MODE == 0) {
        f32x4 acc[16] = {};
        for (int r = 0; r < rounds; ++r) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        }
        float s = 0; for (int i = 0; i < 16; ++i) s += acc[i][0];
        out[blockIdx.x * 512 + threadIdx.x] = s;
    } else {
        f32x16 acc[4] = {};
        for (int r = 0; r < rounds; ++r) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        }
        float s = 0; for (int i = 0; i < 4; ++i) s += acc[i][0];
        out[blockIdx.x * 512 + threadIdx.x] = s;
    }
}
int main() {
    float *o; hipMalloc(&o, 256 * 512 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode) {
        const int rounds = 200000;
        auto run = [&](int r) { if (mode == 0) k<0><<<256, 512>>>(o, r); else k<1><<<256, 512>>>(o, r); };
        run(20000); hipDeviceSynchronize();
        hipEventRecord(e0); run(rounds); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        // flops: mode0: 16 MFMAs x 2048 flop per round per wave; mode1: 4 x 4096
        const double flop = (double)rounds * 256 * 8 * (mode == 0 ? 16 * 2048.0 : 4 * 4096.0);
        printf("%s: %.3f ms, %.1f TFLOP/s\n", mode == 0 ? "16x16x4 (16 independent accumulators)" : "32x32x2 (4 independent accumulators)", ms, flop / ms / 1e9);
    }
}
