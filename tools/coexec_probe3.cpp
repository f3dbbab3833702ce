// Development probe 3: weights into the MFMA's registers through global->VGPR loads against global->LDS loads
// (global_load_lds, no VGPR write) followed by ds_read_b128, in a K-tile-shaped loop, for the fp32 MFMA (16 per
// wavefront and K tile) and the bf16 one (12 per wavefront and K tile).  512 threads, one workgroup per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define AS1 __attribute__((address_space(1)))
#define AS3 __attribute__((address_space(3)))

// MODE 0: no weight traffic; 1: 4 x global_load_dwordx4 per wavefront and iteration into VGPRs (two iterations ahead);
// 2: 4 x global_load_lds_dwordx4 into a per-wavefront LDS ring of three slots, read back with 4 x ds_read_b128
template <int MODE, bool FP32, bool AREAD>
__global__ void __launch_bounds__(512) k(const float *w, float *out, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float *ring = lds + wave * 3 * 1024;                  // 3 slots x 4 KB per wavefront
    float *abuf = lds + 8 * 3 * 1024;                     // 8 K floats of "A tile"
    for (int i = tid; i < 8192; i += 512) abuf[i] = i * 0.001f;
    __syncthreads();
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(tid * 0.001f + i); b[i] = (__bf16)(1.0f + i * 0.5f); }
    float fa = tid * 0.01f, fb = 1.25f;
    f32x16 c = {0};
    const float *wp = w + (size_t)wave * 32768 + lane * 4;    // each wavefront streams its own 128 KB (L2 hits)
    f32x4 r0[4], r1[4];
    float sink = 0;
    auto issue = [&](int kt, int slot, f32x4 *r) {
        const float *src = wp + (size_t)(kt & 31) * 1024;
        if (MODE == 1) for (int j = 0; j < 4; ++j) r[j] = *(const AS1 f32x4 *)(src + j * 256);
        if (MODE == 2)
            for (int j = 0; j < 4; ++j)
                __builtin_amdgcn_global_load_lds((const AS1 void *)(src + j * 256), (AS3 void *)(ring + slot * 1024 + j * 256), 16, 0, 0);
    };
    issue(0, 0, r0);
    issue(1, 1, r1);
    int slot = 0;
    // one K tile: consume the weights of iteration `it` (registers `cur` or ring slot `slot`), request iteration it + 2
    auto body = [&](int it, f32x4 *cur) {
        f32x4 wv[4];
        if (MODE == 1) for (int j = 0; j < 4; ++j) wv[j] = cur[j];
        if (MODE == 2) {
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");           // the loads of iteration it (issued two iterations ago) have landed
            for (int j = 0; j < 4; ++j) wv[j] = *reinterpret_cast<const f32x4 *>(ring + slot * 1024 + j * 256 + lane * 4);
        }
        const int nslot = slot == 2 ? 0 : slot + 1, n2 = nslot == 2 ? 0 : nslot + 1;
        if (MODE != 0) for (int j = 0; j < 4; ++j) { fb += wv[j][0]; }
        issue(it + 2, n2, cur);
        f32x4 l[6];
        if (AREAD) for (int j = 0; j < (FP32 ? 4 : 6); ++j) l[j] = *reinterpret_cast<const f32x4 *>(&abuf[((lane * 36 + j * 1280 + (it & 3) * 8) & 8188)]);
        if (FP32) {
#pragma unroll
            for (int m = 0; m < 16; ++m) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0\n" : "+v"(c) : "v"(fa), "v"(fb));
        } else {
#pragma unroll
            for (int m = 0; m < 12; ++m) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n" : "+v"(c) : "v"(a), "v"(b));
        }
        if (AREAD) for (int j = 0; j < (FP32 ? 4 : 6); ++j) sink += l[j][0];
        slot = nslot;
    };
    for (int it = 0; it < iters; it += 2) {
        body(it, r0);
        body(it + 1, r1);
    }
    float s = sink + fb;
    for (int i = 0; i < 16; ++i) s += c[i];
    out[blockIdx.x * 512 + tid] = s;
}

template <int MODE, bool FP32, bool AREAD>
void run(const float *w, float *out, const char *tag) {
    const int iters = 4000;
    const size_t ldsb = (8 * 3 * 1024 + 8192) * 4;
    hipFuncSetAttribute(reinterpret_cast<const void *>(k<MODE, FP32, AREAD>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    hipLaunchKernelGGL((k<MODE, FP32, AREAD>), dim3(256), dim3(512), ldsb, 0, w, out, iters);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<MODE, FP32, AREAD>), dim3(256), dim3(512), ldsb, 0, w, out, iters);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    printf("%-5s %-34s A reads %d: %7.1f ns per iteration\n", FP32 ? "fp32" : "bf16", tag, (int)AREAD, ms * 1e6 / iters);
}

int main() {
    float *w, *out;
    hipMalloc((void **)&w, 1 << 20);
    hipMemset(w, 0, 1 << 20);
    hipMalloc((void **)&out, 256 * 512 * 4);
    run<0, true, false>(w, out, "mfma only");
    run<0, true, true>(w, out, "mfma");
    run<1, true, true>(w, out, "weights global -> VGPR");
    run<2, true, true>(w, out, "weights global -> LDS -> VGPR");
    run<1, true, false>(w, out, "weights global -> VGPR");
    run<2, true, false>(w, out, "weights global -> LDS -> VGPR");
    run<0, false, false>(w, out, "mfma only");
    run<0, false, true>(w, out, "mfma");
    run<1, false, true>(w, out, "weights global -> VGPR");
    run<2, false, true>(w, out, "weights global -> LDS -> VGPR");
    run<1, false, false>(w, out, "weights global -> VGPR");
    run<2, false, false>(w, out, "weights global -> LDS -> VGPR");
    return 0;
}
