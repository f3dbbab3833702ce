// Development probe 2: a K-tile-like loop body (12 dependent bf16 MFMAs + 96 VALU) with optional pieces added one at a
// time: streaming weight loads (4 x b128 per wavefront and iteration, two iterations ahead), LDS operand reads
// (6 x ds_read_b128), a workgroup barrier.  512 threads per workgroup, one workgroup per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <bool LOADS, bool LDS, bool BAR, int NV>
__global__ void __launch_bounds__(512) k(const float *w, float *out, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[8192];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 8192; i += 512) lds[i] = i * 0.001f;
    __syncthreads();
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(tid * 0.001f + i); b[i] = (__bf16)(1.0f + i * 0.5f); }
    f32x16 c = {0};
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = tid + i;
    // every workgroup streams the same 1 MB (L2 hits after the first), each wavefront its own 128 KB
    const f32x4 *wp = reinterpret_cast<const f32x4 *>(w) + (size_t)wave * 8192 + lane;
    f32x4 r0[4], r1[4];
    if (LOADS) for (int j = 0; j < 4; ++j) { r0[j] = wp[j * 64]; r1[j] = wp[256 + j * 64]; }
    float sink = 0;
    for (int it = 0; it < iters; ++it) {
        f32x4 *cur = (it & 1) ? r1 : r0;
        if (LOADS) {
            for (int j = 0; j < 4; ++j) sink += cur[j][0];
            const int kt = (it + 2) & 31;
            for (int j = 0; j < 4; ++j) cur[j] = wp[kt * 256 + j * 64];
        }
        f32x4 l[6];
        if (LDS) for (int j = 0; j < 6; ++j) l[j] = *reinterpret_cast<const f32x4 *>(&lds[((lane * 20 + j * 1280 + (it & 3) * 8) & 8188)]);
#pragma unroll
        for (int m = 0; m < 12; ++m) {
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n" : "+v"(c) : "v"(a), "v"(b));
#pragma unroll
            for (int v = 0; v < NV; ++v) asm volatile("v_add_f32 %0, %0, %0\n" : "+v"(x[v & 7]));
        }
        if (LDS) for (int j = 0; j < 6; ++j) sink += l[j][0];
        if (BAR) __syncthreads();
    }
    float s = sink;
    for (int i = 0; i < 8; ++i) s += x[i];
    for (int i = 0; i < 16; ++i) s += c[i];
    out[blockIdx.x * 512 + tid] = s;
}

template <bool LOADS, bool LDS, bool BAR, int NV>
void run(const float *w, float *out, const char *tag) {
    const int iters = 4000;
    hipLaunchKernelGGL((k<LOADS, LDS, BAR, NV>), dim3(256), dim3(512), 0, 0, w, out, iters);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<LOADS, LDS, BAR, NV>), dim3(256), dim3(512), 0, 0, w, out, iters);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s NV %2d: %7.1f ns per iteration (2 wavefronts per SIMD x 12 MFMA = 768 cycles of matrix work)\n", tag, NV, ms * 1e6 / iters);
}

int main() {
    float *w, *out;
    hipMalloc((void **)&w, 1 << 20);
    hipMemset(w, 0, 1 << 20);
    hipMalloc((void **)&out, 256 * 512 * 4);
    run<false, false, false, 0>(w, out, "mfma only");
    run<false, false, false, 4>(w, out, "mfma + valu");
    run<false, false, false, 8>(w, out, "mfma + valu");
    run<false, false, true, 8>(w, out, "mfma + valu + barrier");
    run<false, true, false, 8>(w, out, "mfma + valu + lds reads");
    run<true, false, false, 8>(w, out, "mfma + valu + weight loads");
    run<true, true, false, 8>(w, out, "mfma + valu + weight loads + lds");
    run<true, true, true, 8>(w, out, "mfma + valu + weight loads + lds + barrier");
    run<true, true, true, 4>(w, out, "mfma + valu + weight loads + lds + barrier");
    run<true, true, true, 0>(w, out, "mfma + weight loads + lds + barrier");
    run<true, false, false, 0>(w, out, "mfma + weight loads");
    return 0;
}
