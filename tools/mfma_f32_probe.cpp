// Development probe: issue interval of v_mfma_f32_32x32x2_f32 (clock64 ticks and ns), whole-chip rate and the clock it
// implies, with one and two wavefronts per SIMD, one or two accumulator chains, on all 256 CUs or a subset.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <bool TWO>
__global__ void __launch_bounds__(512) k(float *out, long long *cyc, int iters) {
    float a = threadIdx.x * 0.001f, b = 1.5f;
    f32x16 c0 = {0}, c1 = {0};
    long long w0 = wall_clock64(), t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0\n" : "+v"(c0) : "v"(a), "v"(b));
            if (TWO) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0\n" : "+v"(c1) : "v"(a), "v"(b));
            else asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0\n" : "+v"(c0) : "v"(a), "v"(b));
        }
    }
    long long t1 = clock64(), w1 = wall_clock64();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; cyc[1024 + blockIdx.x] = w1 - w0; }
}

template <bool TWO>
void run(int grid, int threads, float *out, long long *cyc, const char *tag) {
    const int iters = 4000;
    hipLaunchKernelGGL((k<TWO>), dim3(grid), dim3(threads), 0, 0, out, cyc, iters);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<TWO>), dim3(grid), dim3(threads), 0, 0, out, cyc, iters);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    long long h[2048];
    hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    const double n = iters * 16.0;
    const double tf = (double)grid * (threads / 64) * n * 4096.0 / (ms * 1e-3) / 1e12;
    printf("%-12s grid %3d threads %3d: wave 0 of WG 0: %6.1f ticks, %6.2f ns per MFMA (clock %.2f GHz); kernel %.0f us = %.1f TFLOP/s\n", tag, grid, threads,
           h[0] / n, h[1024] * 10.0 / n, (double)h[0] / (h[1024] * 10.0), ms * 1e3, tf);
}

int main() {
    float *out; long long *cyc;
    hipMalloc((void **)&out, 256 * 512 * 4);
    hipMalloc((void **)&cyc, 2048 * 8);
    run<false>(256, 256, out, cyc, "one chain");
    run<false>(256, 512, out, cyc, "one chain");
    run<true>(256, 256, out, cyc, "two chains");
    run<true>(256, 512, out, cyc, "two chains");
    run<true>(192, 512, out, cyc, "two chains");
    run<true>(128, 512, out, cyc, "two chains");
    run<true>(32, 512, out, cyc, "two chains");
    return 0;
}
