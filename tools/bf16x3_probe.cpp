// Feasibility probe for round 2 (not part of the library): an fp32 GEMM  C = A W^T  evaluated on the bf16 matrix
// cores.  Every fp32 operand is split exactly into three bf16 terms (a = a0 + a1 + a2, 8 mantissa bits each) and
// the six products a0b0, a0b1, a1b0, a0b2, a1b1, a2b0 are accumulated in fp32 by v_mfma_f32_32x32x16_bf16
// (16x the FLOP rate of v_mfma_f32_32x32x2_f32 on gfx950).  Measures speed (operands pre-split and pre-packed in
// MFMA fragment order on the host, streamed straight from HBM/L2, no LDS) and error against a float64 reference,
// next to the error of a plain fp32 evaluation.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/bf16x3_probe.cpp -o tools/bf16x3_probe.bin
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int MI = 4, NI = 2;              // 32x32 blocks per wavefront: 128 rows x 64 columns
constexpr int WM = 2, WN = 2;              // wavefronts per workgroup: 256 x 128 tile, one workgroup per CU
constexpr int KSTEP = 16;

// A_pack[plane][row block][k step][lane][8], W_pack likewise: one 16-byte load per lane and fragment
extern "C" __global__ __launch_bounds__(256) void gemm_bf16x3(const uint16_t *__restrict__ ap, const uint16_t *__restrict__ wp,
                                                              float *__restrict__ c, int M, int N, int K, int terms) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int rb0 = (blockIdx.x * WM + wm) * MI, cb0 = (blockIdx.y * WN + wn) * NI;     // first row / column block
    const int nks = K / KSTEP, nrb = M / 32, ncb = N / 32;
    const size_t a_plane = (size_t)nrb * nks * 64 * 8, w_plane = (size_t)ncb * nks * 64 * 8;
    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    struct Frags { bf16x8 a[MI][3], w[NI][3]; };
    auto load = [&](int ks, Frags &f) {
#pragma unroll
        for (int p = 0; p < 3; ++p) {
#pragma unroll
            for (int i = 0; i < MI; ++i)
                f.a[i][p] = *reinterpret_cast<const bf16x8 *>(ap + p * a_plane + (((size_t)(rb0 + i) * nks + ks) * 64 + lane) * 8);
#pragma unroll
            for (int j = 0; j < NI; ++j)
                f.w[j][p] = *reinterpret_cast<const bf16x8 *>(wp + p * w_plane + (((size_t)(cb0 + j) * nks + ks) * 64 + lane) * 8);
        }
    };
    auto mma = [&](const Frags &f) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                // smallest terms first
                if (terms >= 6) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i][2], f.w[j][0], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i][1], f.w[j][1], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i][0], f.w[j][2], acc[i][j], 0, 0, 0);
                }
                if (terms >= 3) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i][1], f.w[j][0], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i][0], f.w[j][1], acc[i][j], 0, 0, 0);
                }
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i][0], f.w[j][0], acc[i][j], 0, 0, 0);
            }
    };
    Frags f0, f1;
    load(0, f0);
    int ks = 0;
    for (; ks + 1 < nks; ks += 2) {
        load(ks + 1, f1);
        mma(f0);
        load(ks + 2 < nks ? ks + 2 : nks - 1, f0);
        mma(f1);
    }
    if (ks < nks) mma(f0);
    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (rb0 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = (cb0 + j) * 32 + (lane & 31);
                c[(size_t)row * N + col] = acc[i][j][r];
            }
}

static uint16_t bf16_rne(float x) {
    uint32_t u; memcpy(&u, &x, 4);
    const uint32_t r = u + 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(r >> 16);
}
static float bf16_f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
static void split3(float x, uint16_t out[3]) {
    out[0] = bf16_rne(x);
    const float r1 = x - bf16_f(out[0]);
    out[1] = bf16_rne(r1);
    const float r2 = r1 - bf16_f(out[1]);
    out[2] = bf16_rne(r2);
}

int main(int argc, char **argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 65536, N = argc > 2 ? atoi(argv[2]) : 256, K = argc > 3 ? atoi(argv[3]) : 768;
    const int reps = argc > 4 ? atoi(argv[4]) : 20;
    if (M % (32 * MI * WM) || N % (32 * NI * WN) || K % KSTEP) { printf("M %% %d, N %% %d, K %% 16 must be 0\n", 32 * MI * WM, 32 * NI * WN); return 1; }
    std::vector<float> A((size_t)M * K), W((size_t)N * K);
    uint64_t s = 0x9e3779b97f4a7c15ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (float)((s >> 11) * (1.0 / 9007199254740992.0)) * 2.0f - 1.0f; };
    for (auto &v : A) v = rnd();
    for (auto &v : W) v = rnd() * 0.1f;
    const int nks = K / KSTEP, nrb = M / 32, ncb = N / 32;
    std::vector<uint16_t> ap((size_t)3 * nrb * nks * 64 * 8), wp((size_t)3 * ncb * nks * 64 * 8);
    auto pack = [&](const std::vector<float> &X, int nb, std::vector<uint16_t> &out) {
        const size_t plane = (size_t)nb * nks * 64 * 8;
        for (int b = 0; b < nb; ++b)
            for (int ks = 0; ks < nks; ++ks)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        uint16_t t[3];
                        split3(X[(size_t)(b * 32 + lane % 32) * K + ks * 16 + 8 * (lane / 32) + j], t);
                        for (int p = 0; p < 3; ++p) out[p * plane + (((size_t)b * nks + ks) * 64 + lane) * 8 + j] = t[p];
                    }
    };
    pack(A, nrb, ap);
    pack(W, ncb, wp);
    uint16_t *dap, *dwp; float *dc;
    CK(hipMalloc((void **)&dap, ap.size() * 2)); CK(hipMalloc((void **)&dwp, wp.size() * 2)); CK(hipMalloc((void **)&dc, (size_t)M * N * 4));
    CK(hipMemcpy(dap, ap.data(), ap.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dwp, wp.data(), wp.size() * 2, hipMemcpyHostToDevice));
    const dim3 grid(M / (32 * MI * WM), N / (32 * NI * WN));
    std::vector<float> C((size_t)M * N);
    for (int terms : {6, 3, 1}) {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int i = 0; i < 3; ++i) gemm_bf16x3<<<grid, 256>>>(dap, dwp, dc, M, N, K, terms);
        CK(hipDeviceSynchronize());
        float best = 1e9;
        for (int i = 0; i < reps; ++i) {
            CK(hipEventRecord(e0, 0));
            gemm_bf16x3<<<grid, 256>>>(dap, dwp, dc, M, N, K, terms);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best;
        }
        CK(hipMemcpy(C.data(), dc, C.size() * 4, hipMemcpyDeviceToHost));
        // error against float64 on sampled entries; and the error of an fp32 dot product in natural order for scale
        double emax = 0, e32max = 0, vmax = 0;
        for (int t = 0; t < 4096; ++t) {
            const int r = (int)(((uint64_t)t * 2654435761u) % M), cc = (int)(((uint64_t)t * 40503u) % N);
            double ref = 0; float f32 = 0;
            for (int k = 0; k < K; ++k) { ref += (double)A[(size_t)r * K + k] * W[(size_t)cc * K + k]; f32 += A[(size_t)r * K + k] * W[(size_t)cc * K + k]; }
            emax = fmax(emax, fabs(C[(size_t)r * N + cc] - ref)); e32max = fmax(e32max, fabs((double)f32 - ref)); vmax = fmax(vmax, fabs(ref));
        }
        printf("%d-term bf16: M %d N %d K %d  %.1f us  -> %.1f fp32-equivalent TFLOP/s | max abs err %.3e (plain fp32 dot: %.3e; |C| max %.2f)\n",
               terms, M, N, K, best * 1e3, 2.0 * M * N * K / (best * 1e-3) / 1e12, emax, e32max, vmax);
    }
    return 0;
}
