// Register-chained, transposed first-level tile - a PROBE of the design DESIGN.md section 8 names as the next step, not product code.
//
// Today's first_level_taps hands every layer's activations over through LDS (C/D layout -> A-operand layout: a transposition)
// with two barriers per hand-over; the tile is 0.82 matrix-busy.  Here the product is computed TRANSPOSED - weights as the MFMA's
// A operand, activations as its B operand - so that a layer's accumulators ARE the next layer's B fragments:
//   v_mfma_f32_16x16x4_f32   D[i][j] += sum_k A[i][k] B[k][j];  lane l: A[i = l & 15][k = l >> 4], B[k = l >> 4][j = l & 15],
//                            D[i = 4 (l >> 4) + r][j = l & 15], r = 0..3        with i = channel, j = row, k = input feature.
// Lane (j, g) of a 16-channel block's accumulator holds channels 4 g + r: MFMA step (block cb, register r) of the NEXT layer
// takes K-quad {16 cb + r, + 4, + 8, + 12} from exactly those registers, and the next layer's weights are packed in that K order.
// A wavefront owns 16 rows and ALL 256 channels (16 accumulator blocks of 4 registers); the eight wavefronts of a workgroup share
// every weight, which streams through an LDS ring of 32 KB slabs (8 MFMA steps x 16 channel blocks), one barrier per slab.
// One tile = 128 output rows of a body-part branch: expand_conv (K0 = 64) on rows 3 r + t, the tap's third of the 3-tap
// convolution (K = 256) chained on its activations, for t = 0, 1, 2; then the 1 x 1 convolution, + the centre tap's activations
// (recomputed: they do not fit the 256-register budget of two wavefronts per SIMD next to two accumulator sets).
// Prints: max error against a CPU evaluation of one tile, and the time per tile with every CU busy, next to the MFMA-issue bound.
// build: hipcc -O3 --offload-arch=gfx950 tools/chain_probe.cpp -o tools/chain_probe.bin
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int C = 256, K0 = 64, NCB = C / 16, TR = 128, WROWS = 16;
#ifndef ABL
#define ABL 0      // ablations (timing only): 1 no slab staging / barrier, 2 no LDS reads in the K loop, 3 both, 4 weights per wavefront from L2 (exact), 5 barrier only, 6 staging without the barrier
#endif
#ifndef SYNC
#define SYNC 1     // 1: slabs handed over by LDS counters (wavefronts free-run, no s_barrier in the K loop); 0: one s_barrier per slab;
                   // 2: counters + direct-to-LDS loads (global_load_lds_dwordx4: no VGPR staging, no ds_write), four stages
#endif
#ifndef SLAB_STEPS_
#define SLAB_STEPS_ 8
#endif
constexpr int SLAB_STEPS = SLAB_STEPS_, SLAB_FLOATS = SLAB_STEPS * NCB * 64;    // 8 steps: 8192 floats = 32 KB
constexpr int NSTAGE = SYNC == 2 ? 4 : SLAB_STEPS == 8 ? 3 : 2, AHEAD = SYNC == 2 ? 2 : NSTAGE - 1;   // slabs requested AHEAD slabs before their use
// slabs per layer: a layer with Kc input features has Kc / 4 steps
constexpr int SL_EXP = K0 / 4 / SLAB_STEPS, SL_C = C / 4 / SLAB_STEPS;       // 2, 8
constexpr int SLABS_PER_TILE = 3 * (SL_EXP + SL_C) + SL_C + SL_EXP;         // + the residual tap's expand_conv again
constexpr float SLOPE = 0.2f;

__device__ __forceinline__ float lrelu(float v) { return fmaxf(v, v * SLOPE); }

// weights in slab order: wsl[slab][step][cbq][lane][4] (cbq = channel-block quad); the sequence of slabs of one tile is
// [W0 (2) | W1 tap (8)] x 3 taps, W2 (8), W0 (2)
struct Args {
    const float *x;        // [tiles][3 * TR expand rows][K0]
    const float *wsl;      // SLABS_PER_TILE slabs
    const float *b0, *b1, *b2;
    float *out;            // [tiles][TR][C]
    int tiles_per_wg;
};

__global__ __launch_bounds__(512) void chain_tile(const Args a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, g = lane >> 4;
    constexpr int SV = SLAB_FLOATS / 2048;                     // 16-byte pieces of a slab per thread
    f32x4 stg[SV];                                             // a slab on its way global -> LDS
    auto slab_issue = [&](int s) {
        const float *src = a.wsl + (size_t)(s % SLABS_PER_TILE) * SLAB_FLOATS + tid * 4;
#pragma unroll
        for (int i = 0; i < SV; ++i) stg[i] = *reinterpret_cast<const f32x4 *>(src + i * 2048);
    };
    auto slab_commit = [&](int s) {
        float *dst = lds + (s % NSTAGE) * SLAB_FLOATS + tid * 4;
#pragma unroll
        for (int i = 0; i < SV; ++i) *reinterpret_cast<f32x4 *>(dst + i * 2048) = stg[i];
    };
    // SYNC == 1: ready[stage] counts the wavefronts that have written their share of the slab now in the stage (8 per generation),
    // done[stage] those that have finished reading it; a wavefront polls only when it is more than a slab ahead of the slowest one
    unsigned *ctr = reinterpret_cast<unsigned *>(lds + NSTAGE * SLAB_FLOATS);      // ready[0..NSTAGE), done[NSTAGE..2 NSTAGE)
    auto ctr_add = [&](int idx) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) __hip_atomic_fetch_add(ctr + idx, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    auto ctr_wait = [&](int idx, unsigned need) {
        while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(ctr + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < need) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };
#ifdef STAGGER
    for (int i = 0; i < (int)(blockIdx.x % 32) * STAGGER; ++i) __builtin_amdgcn_s_sleep(16);   // (workgroups out of phase: do they fight for the same L2 lines?)
#endif
    if (SYNC && tid < 2 * NSTAGE) ctr[tid] = 0u;
    if (SYNC) __syncthreads();
    // SYNC == 2: this wavefront's eighth of slab s straight into its ring stage: 4 x 1 KiB per wavefront (LDS address = uniform base + 16 lane)
    auto slab_dma = [&](int s) {
        const float *src = a.wsl + (size_t)(s % SLABS_PER_TILE) * SLAB_FLOATS + wave * 1024 + lane * 4;
        float *dst = lds + (s % NSTAGE) * SLAB_FLOATS + wave * 1024;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + i * 256),
                                             (__attribute__((address_space(3))) void *)(dst + i * 256), 16, 0, 0);
    };
    int slab = 0;                                              // running slab index (the ring position)
    for (int p = 0; p < AHEAD; ++p) {
        if (SYNC == 2) { slab_dma(p); __builtin_amdgcn_s_waitcnt(0x0F70); }
        else { slab_issue(p); slab_commit(p); }
        if (SYNC) ctr_add(p % NSTAGE);
    }
    __syncthreads();
    f32x4 D1[NCB], D2[NCB];
    for (int tile = 0; tile < a.tiles_per_wg; ++tile) {
        const size_t t_idx = (size_t)blockIdx.x * a.tiles_per_wg + tile;
        const float *xt = a.x + t_idx * (3 * TR) * K0;
        const int row = wave * WROWS + j;                      // this lane's output row of the tile
        // one slab's worth of MFMAs: steps [0, 8) of the slab, B operand by step from `bsrc`
#if ABL == 4
        // variant: no LDS, no barrier - every wavefront streams the fragments straight from L2 / L1 into registers, two steps ahead
        // (all eight wavefronts read the same addresses: 32 B/clk of the CU's 64 B/clk vector-memory path)
        auto wload = [&](int gstep, f32x4 (&dst)[4]) {
            const float *src = a.wsl + ((size_t)(gstep % (SLABS_PER_TILE * SLAB_STEPS)) * 4) * 256 + lane * 4;
#pragma unroll
            for (int q = 0; q < 4; ++q) dst[q] = *reinterpret_cast<const f32x4 *>(src + q * 256);
        };
        auto run_slab = [&](auto bsrc, f32x4 (&acc)[NCB]) {
            f32x4 w3[3][4];
            wload(slab * SLAB_STEPS, w3[0]);
            wload(slab * SLAB_STEPS + 1, w3[1]);
#pragma unroll
            for (int st = 0; st < SLAB_STEPS; ++st) {
                wload(slab * SLAB_STEPS + st + 2, w3[(st + 2) % 3]);
                __builtin_amdgcn_sched_barrier(0);
                const float b = bsrc(st);
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        acc[q * 4 + e] = __builtin_amdgcn_mfma_f32_16x16x4f32(w3[st % 3][q][e], b, acc[q * 4 + e], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            ++slab;
        };
#else
        auto run_slab = [&](auto bsrc, f32x4 (&acc)[NCB]) {
#if SYNC == 2 && ABL == 0
            {   // slab + AHEAD goes into the stage that held slab + AHEAD - NSTAGE (two slabs back): every wavefront done with that one?
                const int m = slab + AHEAD;
                if (m >= NSTAGE) ctr_wait(NSTAGE + m % NSTAGE, 8u * (unsigned)(m / NSTAGE));
                slab_dma(m);
            }
#elif ABL != 1 && ABL != 3 && ABL != 5
            slab_issue(slab + AHEAD);
#endif
            __builtin_amdgcn_sched_barrier(0);                 // (the requests stay HERE: the compiler would sink them to their LDS writes)
            const float *sp = lds + (slab % NSTAGE) * SLAB_FLOATS + lane * 4;
            f32x4 wq[2][4];
#pragma unroll
            for (int q = 0; q < 4; ++q) wq[0][q] = *reinterpret_cast<const f32x4 *>(sp + q * 256);
#pragma unroll
            for (int st = 0; st < SLAB_STEPS; ++st) {
                if (ABL < 2 && st + 1 < SLAB_STEPS) {          // the next step's A fragments, a step ahead of their MFMAs
#pragma unroll
                    for (int q = 0; q < 4; ++q) wq[(st + 1) & 1][q] = *reinterpret_cast<const f32x4 *>(sp + ((st + 1) * 4 + q) * 256);
                }
                __builtin_amdgcn_sched_barrier(0);             // (the reads stay at the top of the step, a whole step ahead of their use)
                const float b = bsrc(st);
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#if ABL >= 2
                        acc[q * 4 + e] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[0][q][e], b, acc[q * 4 + e], 0, 0, 0);   // (one step's fragments for all)
#else
                        acc[q * 4 + e] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[st & 1][q][e], b, acc[q * 4 + e], 0, 0, 0);
#endif
                __builtin_amdgcn_sched_barrier(0);
            }
#if SYNC == 2 && ABL == 0
            ctr_add(NSTAGE + slab % NSTAGE);                                       // done reading slab `slab`
            __builtin_amdgcn_s_waitcnt(0x0F70);                                    // vmcnt(0): this wavefront's share of slab + AHEAD has landed
            ctr_add((slab + AHEAD) % NSTAGE);
            ++slab;
            ctr_wait(slab % NSTAGE, 8u * (unsigned)(slab / NSTAGE + 1));           // the next slab is complete
#elif SYNC && ABL == 0
            ctr_add(NSTAGE + slab % NSTAGE);                                       // this wavefront is done reading slab `slab`
            {   // slab + AHEAD goes into the stage that held slab + AHEAD - NSTAGE: every wavefront must be done with that one
                const int m = slab + AHEAD;
                if (m >= NSTAGE) ctr_wait(NSTAGE + m % NSTAGE, 8u * (unsigned)(m / NSTAGE));
                slab_commit(m);
                ctr_add(m % NSTAGE);
            }
            ++slab;
            ctr_wait(slab % NSTAGE, 8u * (unsigned)(slab / NSTAGE + 1));           // the next slab is complete
#else
#if ABL != 1 && ABL != 3 && ABL != 5
            slab_commit(slab + AHEAD);
#endif
#if ABL != 1 && ABL != 3 && ABL != 6
            __syncthreads();
#endif
            ++slab;
#endif
        };
#endif
        float xv[K0 / 4];                                      // this lane's operand values of the NEXT expand_conv, requested a phase ahead
        auto gather = [&](int tap) {
            const float *xr = xt + (size_t)(3 * row + tap) * K0 + g;
#pragma unroll
            for (int s = 0; s < K0 / 4; ++s) xv[s] = xr[4 * s];
            __builtin_amdgcn_sched_barrier(0);
        };
        auto expand = [&](int next_tap) {                      // D1 <- lrelu(W0 x + b0) on the gathered values; then request the next ones
            float xc[K0 / 4];
#pragma unroll
            for (int s = 0; s < K0 / 4; ++s) xc[s] = xv[s];
            if (next_tap >= 0) gather(next_tap);
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) D1[cb] = *reinterpret_cast<const f32x4 *>(a.b0 + cb * 16 + 4 * g);
#pragma unroll
            for (int m = 0; m < SL_EXP; ++m) run_slab([&](int st) { return xc[m * SLAB_STEPS + st]; }, D1);
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                for (int r = 0; r < 4; ++r) D1[cb][r] = lrelu(D1[cb][r]);
        };
        gather(0);
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) D2[cb] = *reinterpret_cast<const f32x4 *>(a.b1 + cb * 16 + 4 * g);
#pragma unroll 1
        for (int tap = 0; tap < 3; ++tap) {
            expand(tap < 2 ? tap + 1 : 1);                     // (after the last tap: the residual tap's values again)
            // the tap's third of the 3-tap convolution: step (cbp, r) takes D1[cbp][r]
#pragma unroll
            for (int m = 0; m < SL_C; ++m) run_slab([&](int st) { return D1[(m * SLAB_STEPS + st) >> 2][st & 3]; }, D2);
        }
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int r = 0; r < 4; ++r) D2[cb][r] = lrelu(D2[cb][r]);
        // the 1 x 1 convolution on D2 -> D1 (reused as the output accumulators)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) D1[cb] = *reinterpret_cast<const f32x4 *>(a.b2 + cb * 16 + 4 * g);
#pragma unroll
        for (int m = 0; m < SL_C; ++m) run_slab([&](int st) { return D2[(m * SLAB_STEPS + st) >> 2][st & 3]; }, D1);
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int r = 0; r < 4; ++r) D2[cb][r] = lrelu(D1[cb][r]);            // level output before the residual, parked in D2
        expand(-1);                                            // the residual tap's activations again (5 % of the tile's MFMAs)
        float *orow = a.out + (t_idx * TR + row) * C + 4 * g;
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            const f32x4 v = D2[cb] + D1[cb];
            *reinterpret_cast<f32x4 *>(orow + cb * 16) = v;
        }
    }
}

static float frand(unsigned &s) { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; }

int main(int argc, char **argv) {
    const int tiles_per_wg = argc > 1 ? atoi(argv[1]) : 8, nwg = 256;
    const size_t tiles = (size_t)nwg * tiles_per_wg;
    unsigned seed = 12345;
    std::vector<float> W0(C * K0), W1((size_t)C * 3 * C), W2((size_t)C * C), b0(C), b1(C), b2(C), x(tiles * 3 * TR * K0);
    for (auto &v : W0) v = frand(seed) * 0.25f;
    for (auto &v : W1) v = frand(seed) * 0.07f;
    for (auto &v : W2) v = frand(seed) * 0.12f;
    for (auto &v : b0) v = frand(seed) * 0.1f;
    for (auto &v : b1) v = frand(seed) * 0.1f;
    for (auto &v : b2) v = frand(seed) * 0.1f;
    for (auto &v : x) v = frand(seed);
    // ---- slabs.  element [slab][step][cbq][lane][e]: channel 16 (4 cbq + e) + (lane & 15), input feature kq(step, lane >> 4)
    std::vector<float> wsl((size_t)SLABS_PER_TILE * SLAB_FLOATS);
    auto fill = [&](int slab0, int nslab, auto weight, bool chained) {
        for (int m = 0; m < nslab; ++m)
            for (int st = 0; st < SLAB_STEPS; ++st)
                for (int cbq = 0; cbq < 4; ++cbq)
                    for (int l = 0; l < 64; ++l)
                        for (int e = 0; e < 4; ++e) {
                            const int step = m * SLAB_STEPS + st, g = l >> 4, ch = 16 * (4 * cbq + e) + (l & 15);
                            const int k = chained ? 16 * (step >> 2) + 4 * g + (step & 3) : 4 * step + g;
                            wsl[(size_t)(slab0 + m) * SLAB_FLOATS + ((st * 4 + cbq) * 64 + l) * 4 + e] = weight(ch, k);
                        }
    };
    int s0 = 0;
    for (int tap = 0; tap < 3; ++tap) {
        fill(s0, SL_EXP, [&](int ch, int k) { return W0[ch * K0 + k]; }, false); s0 += SL_EXP;
        fill(s0, SL_C, [&](int ch, int k) { return W1[(size_t)ch * 3 * C + tap * C + k]; }, true); s0 += SL_C;
    }
    fill(s0, SL_C, [&](int ch, int k) { return W2[(size_t)ch * C + k]; }, true); s0 += SL_C;
    fill(s0, SL_EXP, [&](int ch, int k) { return W0[ch * K0 + k]; }, false); s0 += SL_EXP;
    if (s0 != SLABS_PER_TILE) { printf("slab count\n"); return 1; }
    Args a;
    float *dx, *dw, *db0, *db1, *db2, *dout;
    hipMalloc(&dx, x.size() * 4); hipMalloc(&dw, wsl.size() * 4); hipMalloc(&db0, C * 4); hipMalloc(&db1, C * 4); hipMalloc(&db2, C * 4);
    hipMalloc(&dout, tiles * TR * C * 4);
    hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dw, wsl.data(), wsl.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(db0, b0.data(), C * 4, hipMemcpyHostToDevice); hipMemcpy(db1, b1.data(), C * 4, hipMemcpyHostToDevice); hipMemcpy(db2, b2.data(), C * 4, hipMemcpyHostToDevice);
    a.x = dx; a.wsl = dw; a.b0 = db0; a.b1 = db1; a.b2 = db2; a.out = dout; a.tiles_per_wg = tiles_per_wg;
    const int lds_bytes = NSTAGE * SLAB_FLOATS * 4 + 64;
    hipFuncSetAttribute(reinterpret_cast<const void *>(chain_tile), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    chain_tile<<<nwg, 512, lds_bytes>>>(a);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
    // ---- check tile 0 of workgroup 0 and the last tile against a double-precision CPU evaluation
    std::vector<float> out(tiles * TR * C);
    hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost);
    double worst = 0, ref_max = 0;
    for (size_t t : {(size_t)0, tiles - 1}) {
        for (int row = 0; row < TR; row += 7) {
            std::vector<double> h1(3 * C), h2(C);
            for (int tap = 0; tap < 3; ++tap)
                for (int c = 0; c < C; ++c) {
                    double s = b0[c];
                    for (int k = 0; k < K0; ++k) s += (double)W0[c * K0 + k] * x[(t * 3 * TR + 3 * row + tap) * K0 + k];
                    h1[tap * C + c] = s > 0 ? s : s * SLOPE;
                }
            for (int c = 0; c < C; ++c) {
                double s = b1[c];
                for (int k = 0; k < 3 * C; ++k) s += (double)W1[(size_t)c * 3 * C + k] * h1[k];
                h2[c] = s > 0 ? s : s * SLOPE;
            }
            for (int c = 0; c < C; ++c) {
                double s = b2[c];
                for (int k = 0; k < C; ++k) s += (double)W2[(size_t)c * C + k] * h2[k];
                const double ref = (s > 0 ? s : s * SLOPE) + h1[C + c];
                worst = std::max(worst, std::fabs(ref - out[(t * TR + row) * C + c]));
                ref_max = std::max(ref_max, std::fabs(ref));
            }
        }
    }
    printf("max abs error against the CPU evaluation: %.3e (|ref| max %.2f)\n", worst, ref_max);
    // ---- time
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 20; ++i) chain_tile<<<nwg, 512, lds_bytes>>>(a);
    hipEventRecord(e0);
    const int reps = 50;
    for (int i = 0; i < reps; ++i) chain_tile<<<nwg, 512, lds_bytes>>>(a);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us_tile = ms * 1e3 / reps / tiles_per_wg;
    const double mfma = 8.0 * (3 * (SL_EXP + SL_C) + SL_C + SL_EXP) * SLAB_STEPS * NCB;     // per workgroup and tile
    const double flop_alg = 2.0 * TR * (3.0 * K0 * C + 3.0 * C * C + (double)C * C);
    printf("%d tiles of %d rows per workgroup, 256 workgroups: %.1f us per tile (%.1f us per 64 rows; the product's 64-row tile: ~79 us, ~88 in its timing build)\n",
           tiles_per_wg, TR, us_tile, us_tile / 2);
    printf("MFMA issue bound (%.0f x 16x16x4 per tile, 32 cycles each, 4 SIMDs at 2.4 GHz): %.1f us per tile -> %.2f of it; algorithmic %.1f TFLOP/s of 157.3 chip-wide\n",
           mfma, mfma * 32 / 4 / 2.4e3, mfma * 32 / 4 / 2.4e3 / us_tile, flop_alg * 256 / us_tile / 1e6);
    return 0;
}
