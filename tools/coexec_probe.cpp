// Development probe: how many independent VALU instructions fit between two v_mfma_f32_32x32x16_bf16 of one wavefront
// before the loop slows down, with the accumulator in VGPRs or in AccVGPRs, at one and two wavefronts per SIMD.
//   hipcc -O3 --offload-arch=gfx950 tools/coexec_probe.cpp -o tools/coexec_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define VALU1 "v_add_f32 %[x0], %[x0], %[x0]\n"
#define VALU2 VALU1 "v_add_f32 %[x1], %[x1], %[x1]\n"
#define VALU4 VALU2 "v_add_f32 %[x2], %[x2], %[x2]\n" "v_add_f32 %[x3], %[x3], %[x3]\n"
#define MFMA_V(acc) "v_mfma_f32_32x32x16_bf16 %[" #acc "], %[a], %[b], %[" #acc "]\n"

template <int NV, bool AGPR, bool TWO>
__global__ void __launch_bounds__(512) k(float *out, long long *cyc, int iters) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(1.0f + i * 0.5f); }
    f32x16 c0 = {0}, c1 = {0};
    float x0 = threadIdx.x, x1 = 1.f, x2 = 2.f, x3 = 3.f;
    long long w0 = wall_clock64();
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#define BODY(CONSTR)                                                                                        \
        asm volatile(                                                                                       \
            MFMA_V(c0)                                                                                      \
            "%=:\n"                                                                                         \
            : [c0] CONSTR(c0), [c1] CONSTR(c1), [x0] "+v"(x0), [x1] "+v"(x1), [x2] "+v"(x2), [x3] "+v"(x3)  \
            : [a] "v"(a), [b] "v"(b));
        // one MFMA (dependent chain on c0, or alternating c0/c1 when TWO) followed by NV VALU instructions
#define STEP(ACC, CONSTR)                                                                                          \
        asm volatile(MFMA_V(ACC) : [c0] CONSTR(c0), [c1] CONSTR(c1) : [a] "v"(a), [b] "v"(b));                      \
        if (NV >= 1) asm volatile(VALU1 : [x0] "+v"(x0));                                                           \
        if (NV >= 2) asm volatile("v_add_f32 %[x1], %[x1], %[x1]\n" : [x1] "+v"(x1));                               \
        if (NV >= 4) asm volatile("v_add_f32 %[x2], %[x2], %[x2]\nv_add_f32 %[x3], %[x3], %[x3]\n" : [x2] "+v"(x2), [x3] "+v"(x3)); \
        if (NV >= 6) asm volatile("v_add_f32 %[x0], %[x0], %[x0]\nv_add_f32 %[x1], %[x1], %[x1]\n" : [x0] "+v"(x0), [x1] "+v"(x1)); \
        if (NV >= 8) asm volatile("v_add_f32 %[x2], %[x2], %[x2]\nv_add_f32 %[x3], %[x3], %[x3]\n" : [x2] "+v"(x2), [x3] "+v"(x3)); \
        if (NV >= 12) asm volatile("v_add_f32 %[x0], %[x0], %[x0]\nv_add_f32 %[x1], %[x1], %[x1]\nv_add_f32 %[x2], %[x2], %[x2]\nv_add_f32 %[x3], %[x3], %[x3]\n" : [x0] "+v"(x0), [x1] "+v"(x1), [x2] "+v"(x2), [x3] "+v"(x3));
        if constexpr (AGPR) {
            STEP(c0, "+a") if constexpr (TWO) { STEP(c1, "+a") } else { STEP(c0, "+a") }
            STEP(c0, "+a") if constexpr (TWO) { STEP(c1, "+a") } else { STEP(c0, "+a") }
        } else {
            STEP(c0, "+v") if constexpr (TWO) { STEP(c1, "+v") } else { STEP(c0, "+v") }
            STEP(c0, "+v") if constexpr (TWO) { STEP(c1, "+v") } else { STEP(c0, "+v") }
        }
    }
    long long t1 = clock64();
    long long w1 = wall_clock64();
    float s = x0 + x1 + x2 + x3;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; cyc[256 + blockIdx.x] = w1 - w0; }
}

template <int NV, bool AGPR, bool TWO>
void run(int threads, float *out, long long *cyc, const char *tag) {
    const int iters = 20000;
    hipLaunchKernelGGL((k<NV, AGPR, TWO>), dim3(256), dim3(threads), 0, 0, out, cyc, iters);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<NV, AGPR, TWO>), dim3(256), dim3(threads), 0, 0, out, cyc, iters);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double tf = 256.0 * (threads / 64) * iters * 4.0 * 32768.0 / (ms * 1e-3) / 1e12;
    long long h[512];
    hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    long long mn = h[0], wn = h[256];
    for (int i = 1; i < 256; ++i) { mn = h[i] < mn ? h[i] : mn; wn = h[256 + i] < wn ? h[256 + i] : wn; }
    printf("%-28s threads %4d: %7.1f clock64 ticks, %6.2f ns per MFMA (+%d VALU); kernel %.1f us = %.0f TFLOP/s\n", tag, threads,
           (double)mn / (iters * 4.0), (double)wn * 10.0 / (iters * 4.0), NV, ms * 1e3, tf);
}

int main() {
    float *out; long long *cyc;
    hipMalloc((void **)&out, 256 * 512 * 4);
    hipMalloc((void **)&cyc, 512 * 8);
#define ALL(NV)                                                        \
    run<NV, false, false>(256, out, cyc, "vgpr acc, one chain");       \
    run<NV, false, false>(512, out, cyc, "vgpr acc, one chain");       \
    run<NV, true, false>(256, out, cyc, "agpr acc, one chain");        \
    run<NV, true, false>(512, out, cyc, "agpr acc, one chain");        \
    run<NV, false, true>(256, out, cyc, "vgpr acc, two chains");       \
    run<NV, false, true>(512, out, cyc, "vgpr acc, two chains");       \
    run<NV, true, true>(512, out, cyc, "agpr acc, two chains");
    ALL(0) ALL(2) ALL(4) ALL(8) ALL(12)
    return 0;
}
