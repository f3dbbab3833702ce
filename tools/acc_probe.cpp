// Development probe: does the fp32 MFMA's accumulator traffic share a register-file port with operand data arriving
// from LDS / memory?  v_mfma_f32_32x32x2_f32 reads and writes 16 accumulator registers per 64 cycles - 64 B per clock and
// direction per SIMD - for 4096 FLOP.  A K-tile-shaped loop (16 MI MFMAs per wavefront and iteration, 4 MI ds_read_b128 of
// "A fragments", 4 global b128 loads of "weights" two iterations ahead) is timed with the accumulators in ArchVGPRs ("+v")
// and in AccVGPRs ("+a"), with eight wavefronts per CU (two per SIMD) and with four (one per SIMD).
//   hipcc -O3 --offload-arch=gfx950 tools/acc_probe.cpp -o tools/acc_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define AS1 __attribute__((address_space(1)))

// LOADS: 0 none, 1 ds reads only, 2 global loads only, 3 both
template <int THREADS, int MI, bool AGPR, int LOADS>
__global__ void __launch_bounds__(THREADS) k(const float *w, float *out, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 16384; i += THREADS) lds[i] = w[(i * 37 + blockIdx.x) & 262143];      // (zero or random: main)
    __syncthreads();
    f32x16 c[MI];
    for (int m = 0; m < MI; ++m)
        for (int i = 0; i < 16; ++i) c[m][i] = 0.f;
    const float *wp = w + (size_t)wave * 32768 + lane * 4;    // each wavefront streams its own 128 KB (L2 hits)
    f32x4 r0[4], r1[4], r2[4];
    auto issue = [&](int kt, f32x4 *r) {
        const float *src = wp + (size_t)(kt & 31) * 1024;
        if (LOADS & 2)
            for (int j = 0; j < 4; ++j) r[j] = *(const AS1 f32x4 *)(src + j * 256);
    };
    for (int j = 0; j < 4; ++j) r0[j] = r1[j] = r2[j] = *(const AS1 f32x4 *)(wp + j * 256);
    issue(0, r0);
    issue(1, r1);
    f32x4 an[MI];
    for (int m = 0; m < MI; ++m) an[m] = *reinterpret_cast<const f32x4 *>(&lds[lane * 36 + m * 1152]);
    auto body = [&](int it, f32x4 *cur, f32x4 *nxt2) {
        issue(it + 2, nxt2);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 av[MI];
#pragma unroll
            for (int m = 0; m < MI; ++m) av[m] = an[m];
            if (LOADS & 1) {                            // the next quad's A fragments, one quad ahead
#pragma unroll
                for (int m = 0; m < MI; ++m)
                    an[m] = *reinterpret_cast<const f32x4 *>(&lds[(lane * 36 + m * 1152 + q * 4 + (it & 3) * 2304) & 16380]);
            }
            const f32x4 wv = cur[q];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int m = 0; m < MI; ++m) {
                    if (AGPR) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0\n" : "+a"(c[m]) : "v"(av[m][kk]), "v"(wv[kk]));
                    else asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0\n" : "+v"(c[m]) : "v"(av[m][kk]), "v"(wv[kk]));
                }
        }
    };
    for (int it = 0; it < iters; it += 3) {
        body(it, r0, r2);
        body(it + 1, r1, r0);
        body(it + 2, r2, r1);
    }
    float s = 0;
    for (int m = 0; m < MI; ++m)
        for (int i = 0; i < 16; ++i) s += c[m][i];
    out[blockIdx.x * THREADS + tid] = s;
}

template <int THREADS, int MI, bool AGPR, int LOADS>
void run(const float *w, float *out) {
    const int iters = 3000;
    const size_t ldsb = 16384 * 4;
    auto fn = k<THREADS, MI, AGPR, LOADS>;
    hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    hipLaunchKernelGGL(fn, dim3(256), dim3(THREADS), ldsb, 0, w, out, iters);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(fn, dim3(256), dim3(THREADS), ldsb, 0, w, out, iters);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double ns = ms * 1e6 / iters;
    const double mfma_per_simd = 16.0 * MI * (THREADS / 256);          // MFMAs per SIMD and iteration
    const double flops = 256.0 * (THREADS / 64) * 16.0 * MI * 4096.0;  // per iteration, whole chip
    printf("%d waves/CU  MI %d  acc in %s  loads %s: %7.1f ns per iteration = %5.1f ns per 16 MFMAs of a SIMD, %6.1f TFLOP/s\n", THREADS / 64, MI,
           AGPR ? "AccVGPR" : "ArchVGPR", LOADS == 0 ? "none      " : LOADS == 1 ? "ds        " : LOADS == 2 ? "global    " : "ds+global ", ns,
           ns * 16.0 / mfma_per_simd, flops / ns * 1e-3);
}

template <int THREADS, int MI>
void sweep(const float *w, float *out) {
    run<THREADS, MI, false, 0>(w, out);
    run<THREADS, MI, true, 0>(w, out);
    run<THREADS, MI, false, 1>(w, out);
    run<THREADS, MI, true, 1>(w, out);
    run<THREADS, MI, false, 2>(w, out);
    run<THREADS, MI, true, 2>(w, out);
    run<THREADS, MI, false, 3>(w, out);
    run<THREADS, MI, true, 3>(w, out);
}

int main(int argc, char **argv) {
    // operands: zeros (a quiet datapath: the clock DVFS allows an idle-looking chip) or, with "rand", values in [-1, 1)
    float *w, *out;
    hipMalloc((void **)&w, 1 << 20);
    hipMemset(w, 0, 1 << 20);
    if (argc > 1 && argv[1][0] == 'r') {
        static float h[1 << 18];
        unsigned s = 12345u;
        for (int i = 0; i < (1 << 18); ++i) { s = s * 1664525u + 1013904223u; h[i] = (float)(s >> 8) / 8388608.0f - 1.0f; }
        hipMemcpy(w, h, 1 << 20, hipMemcpyHostToDevice);
        printf("random operands\n");
    } else printf("zero operands\n");
    hipMalloc((void **)&out, 256 * 512 * 4);
    for (int rep = 0; rep < 1; ++rep) {
        sweep<512, 1>(w, out);
        sweep<512, 2>(w, out);
        sweep<256, 1>(w, out);
        sweep<256, 2>(w, out);
        sweep<256, 4>(w, out);
    }
    return 0;
}
