// Register-chained first-level tile, fourth form (round 6) - a PROBE, not product code.
//
// chain_probe3.cpp reaches 74.4 us per 64 rows (0.87 of the MFMA-issue bound) with four wavefronts of 512 registers - a budget
// the product's persistent kernel (512-thread workgroups: two wavefronts per SIMD, 256 registers each) cannot give a tile.
// This form fits that kernel: a 512-thread workgroup whose wavefronts 0-3 (one per SIMD) are MFMA wavefronts of 16 rows x 256
// channels each on v_mfma_f32_16x16x4_f32 (accumulator sets of 64 registers) and whose wavefronts 4-7 - their SIMD partners -
// are LOADERS: they stream the weight slabs global -> VGPR -> LDS ring, three slabs in flight, and do nothing else.
//   v_mfma_f32_16x16x4_f32   D[i][j] += sum_k A[i][k] B[k][j];  lane l: A[i = l & 15][k = l >> 4], B[k = l >> 4][j = l & 15],
//                            D[i = 4 (l >> 4) + r][j = l & 15], r = 0..3        i = channel, j = row, k = input feature
// Register r of channel block cb (16 channels) of a layer's accumulators is the B operand of the next layer's K step (cb, r) =
// the features {16 cb + r, + 4, + 8, + 12}; the next layer's weights are packed in that order.
// Hand-over: ONE s_barrier per 16 KiB slab (64 MFMAs per MFMA wavefront) that all eight wavefronts pass: the loaders have
// written slab t + 2 before barrier t, the MFMA wavefronts read slab t + 1 after it (ring of four stages).
// One tile = 64 output rows of a body-part branch; layers, order of the taps, lazy activations, output-stationary 1 x 1
// convolution (four channel blocks of the output at a time): chain_probe3.cpp.
// build: hipcc -O3 --offload-arch=gfx950 tools/chain_probe4.cpp -o tools/chain_probe4.bin   (-DABL=1: no loaders, no barriers)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int C = 256, K0 = 64, NCB = C / 16, TR = 64;
#ifndef ABL
#define ABL 0
#endif
#ifndef LOAD_AHEAD
#define LOAD_AHEAD 3      // slabs a loader has in flight (16 registers each)
#endif
#ifndef HANDOVER
#define HANDOVER 0        // 0: one s_barrier per slab for all eight wavefronts; 1: progress words in LDS (plain ds_write / ds_read, no barrier:
                          // a loader publishes the number of slabs it has written, an MFMA wavefront the number it has finished reading)
#endif
#ifndef NSTAGE_
#define NSTAGE_ 4
#endif
#ifndef LOADER_ABL
#define LOADER_ABL 0      // timing ablations of the loaders: bit 0 no global loads, bit 1 no LDS writes (results are wrong)
#endif
#ifndef LOADER_DMA
#define LOADER_DMA 0      // 1: the loaders use buffer_load ... lds (LDS-DMA) instead of global_load + ds_write_b128 (with HANDOVER 0)
#endif
#ifndef GATHER_SCALAR
#define GATHER_SCALAR 0   // 1: the operand gather as the product does it (4-byte loads)
#endif
#ifndef STORE_SC1
#define STORE_SC1 0       // 1: write-through output stores, as the product's hand-off needs them
#endif
#ifndef STAGGER
#define STAGGER 0         // n: workgroup b starts (37 b mod 64) * n * 64 * 64 cycles late
#endif
#ifndef NIMG
#define NIMG 1
#endif
#ifndef BARRIER_AT
#define BARRIER_AT 10     // position of the slab's barrier in its last half group (0..15)
#endif
constexpr int SLAB_FLOATS = 4096;                                            // 16 KiB: 16 fragments = 64 MFMAs per MFMA wavefront
constexpr int NSTAGE = NSTAGE_;
constexpr int SL_EXP = K0 / 16, SL_C = C / 16;                               // 4, 16 slabs per layer
constexpr int SL_TAP = SL_EXP + SL_C, SLABS_PER_TILE = 3 * SL_TAP + SL_C;    // 20, 76
static_assert(SL_TAP % NSTAGE == 0 && SLABS_PER_TILE % NSTAGE == 0, "a slab's ring stage is a compile-time constant");
constexpr float SLOPE = 0.2f;

__device__ __forceinline__ float lrelu(float v) { return __builtin_fmaxf(v, v * SLOPE); }
#define CFENCE() __atomic_signal_fence(__ATOMIC_SEQ_CST)

struct Args {
    long long *clk;        // per workgroup: shader cycles and 100 MHz wall-clock ticks of its run (the clock the tile code sustains)
    const float *x;        // [tiles][3 * TR expand rows][K0]
    const float *wsl;      // SLABS_PER_TILE slabs
    const float *b0, *b1, *b2;
    float *out;            // [tiles][TR][C]
    int tiles_per_wg;
};

template <int V> using IC = std::integral_constant<int, V>;

__global__ __launch_bounds__(512) void chain_tile(const Args a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    auto rsrc_of = [](const float *p) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p), 0, 0x7fffffff, 0x00020000); };
    auto wg_barrier = [&]() {                                   // this wavefront's LDS traffic done; everyone here (no vmcnt wait)
        CFENCE();
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        CFENCE();
    };
#if STAGGER
    for (int i = 0; i < (int)((blockIdx.x * 37) % 64) * STAGGER; ++i) __builtin_amdgcn_s_sleep(64);   // (workgroups out of phase, as in the persistent kernel)
#endif
    const __amdgpu_buffer_rsrc_t rw = rsrc_of(a.wsl + (size_t)(blockIdx.x % NIMG) * SLABS_PER_TILE * SLAB_FLOATS);   // (NIMG copies of the stream: the product has five branches)
    const int total = a.tiles_per_wg * SLABS_PER_TILE;          // slabs this workgroup consumes; barrier t ends slab t
    // HANDOVER 1: prog[0..3] = slabs written by loader 0..3, prog[4..7] = slabs finished by MFMA wavefront 0..3
    // (plain LDS accesses between compiler fences, NOT volatile: a volatile access is followed by s_waitcnt lgkmcnt(0) - a stall of
    //  an LDS round trip in the MFMA wavefront, twice per slab: 80.8 us per tile against 73.8 with the barrier)
    int *prog = reinterpret_cast<int *>(lds + NSTAGE * SLAB_FLOATS);
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    auto prog_peek = [&](int first) {                           // four progress words: one 16-byte read (its use may come much later)
        CFENCE();
        const i32x4 v = *reinterpret_cast<const i32x4 *>(prog + first);
        CFENCE();
        return v;
    };
    auto min4 = [](const i32x4 &v) { return min(min(v[0], v[1]), min(v[2], v[3])); };
    auto prog_min = [&](int first) { return min4(prog_peek(first)); };
    if (HANDOVER == 1) {
        if (tid < 8) prog[tid] = 0;
        __syncthreads();
    }
    if (wave >= 4) {
        // ---------------------------------------------------------------- loader: a quarter of every slab
        if (ABL == 1) return;
        const int lw = wave - 4;
        const int voff = (lw * 1024 + lane * 4) * 4;
        float *dst = lds + lw * 1024 + lane * 4;
        f32x4 stg[LOAD_AHEAD][4];
        auto issue = [&](int m, f32x4 (&r)[4]) {                // slab m of the workgroup's stream (the tile sequence repeats)
            const int off = __builtin_amdgcn_readfirstlane((m % SLABS_PER_TILE) * SLAB_FLOATS * 4);
            if (LOADER_ABL & 1) {                               // (timing ablation: no global loads after the first round)
                if (m >= LOAD_AHEAD) return;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) r[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, voff, off + i * 1024, 0));
        };
        auto commit = [&](int m, const f32x4 (&r)[4]) {
            float *d = dst + (m % NSTAGE) * SLAB_FLOATS;
            if ((LOADER_ABL & 2) && m >= NSTAGE) return;        // (timing ablation: no LDS writes after the ring's first fill)
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4 *>(d + i * 256) = r[i];
        };
#if LOADER_DMA
        // global -> LDS directly (buffer_load ... lds: no VGPR, no ds_write_b128 - the ablations put 2.6 of the loaders' 4.3 us per tile
        // on the LDS stores): a quarter slab = 4 pieces of 1 KiB (LDS address = uniform base + 16 lane).  Slab i + 3 is requested between
        // barriers i - 1 and i, into the stage slab i - 1 has just left; vmcnt(4) then says slab i + 2 has landed.
        auto dma = [&](int m) {
            const int off = __builtin_amdgcn_readfirstlane((m % SLABS_PER_TILE) * SLAB_FLOATS * 4);
            float *d = lds + (m % NSTAGE) * SLAB_FLOATS + lw * 1024;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void *)(d + i * 256), 16, lane * 16, off + (lw * 1024 + i * 256) * 4, 0, 0);
        };
        dma(0); dma(1); dma(2);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");        // slabs 0, 1 landed
        wg_barrier();                                           // opening barrier
        for (int t = 0; t < total; ++t) {
            dma(t + 3);
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");    // slab t + 2 landed
            wg_barrier();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#elif HANDOVER == 0
        // slabs 0 and 1 before the opening barrier; then slab t + 2 before barrier t
#pragma unroll
        for (int i = 0; i < LOAD_AHEAD; ++i) issue(i, stg[i]);
        int m = 0;                                              // next slab to commit; slab m + LOAD_AHEAD is the next to request
        auto step = [&](auto i_tag) {
            constexpr int I = decltype(i_tag)::value;
            commit(m, stg[I]);
            issue(m + LOAD_AHEAD, stg[I]);
            ++m;
        };
        step(IC<0>{});
        step(IC<1 % LOAD_AHEAD>{});
        wg_barrier();                                           // opening barrier: slabs 0, 1 are in the ring
        // (unrolled by LOAD_AHEAD so that the register sets are addressed statically)
        for (int t = 0; t < total; t += LOAD_AHEAD) {
#pragma unroll
            for (int u = 0; u < LOAD_AHEAD; ++u) {
                if (t + u >= total) break;
                if (u == 0) step(IC<2 % LOAD_AHEAD>{});
                if (u == 1) step(IC<(2 + 1) % LOAD_AHEAD>{});
                if (u == 2) step(IC<(2 + 2) % LOAD_AHEAD>{});
                if (u == 3) step(IC<(2 + 3) % LOAD_AHEAD>{});
                wg_barrier();
            }
        }
#else
        // slab m may be written once every MFMA wavefront has finished slab m - NSTAGE; its quarter written, the loader publishes m + 1
#pragma unroll
        for (int i = 0; i < LOAD_AHEAD; ++i) issue(i, stg[i]);
        int m = 0;
        auto step = [&](auto i_tag) {
            constexpr int I = decltype(i_tag)::value;
            while (__builtin_amdgcn_readfirstlane(prog_min(4)) < m - NSTAGE + 1) __builtin_amdgcn_s_sleep(8);
            CFENCE();
            commit(m, stg[I]);
            CFENCE();
            prog[lw] = m + 1;
            issue(m + LOAD_AHEAD, stg[I]);
            ++m;
        };
        for (int t = 0; t < total + 1; t += LOAD_AHEAD) {
#pragma unroll
            for (int u = 0; u < LOAD_AHEAD; ++u) {
                if (t + u >= total + 1) break;
                if (u == 0) step(IC<0>{});
                if (u == 1) step(IC<1 % LOAD_AHEAD>{});
                if (u == 2) step(IC<2 % LOAD_AHEAD>{});
                if (u == 3) step(IC<3 % LOAD_AHEAD>{});
            }
        }
#endif
        return;
    }
    // -------------------------------------------------------------------- MFMA wavefront: 16 rows x 256 channels
    const int j = lane & 15, g = lane >> 4;
    const long long c_start = __builtin_readcyclecounter(), w_start = wall_clock64();
    f32x4 wq[2][4];                                             // weight fragments of the current / the next half group
    int slab_t = 0;                                             // the slab this wavefront is multiplying (HANDOVER 1)
    if (ABL != 1) {
        if (HANDOVER == 0) wg_barrier();
        else { while (__builtin_amdgcn_readfirstlane(prog_min(0)) < 1) __builtin_amdgcn_s_sleep(1); CFENCE(); }
    }
#pragma unroll
    for (int f = 0; f < 4; ++f) wq[0][f] = *reinterpret_cast<const f32x4 *>(lds + lane * 4 + f * 256);

    f32x4 DA[NCB], DB[NCB], O[2][4];     // expand_conv activations (the residual after the last tap) | 3-tap sums | 1 x 1 outputs of two block groups
    float xv[K0 / 4];                    // this lane's operand values of the next expand_conv
    // One slab of 64 MFMAs in 4 half groups of 16 (a 16-byte fragment read per lane = the A operands of 4 MFMAs): MFMA n of half
    // group hg takes element e = n >> 2 of fragment f = n & 3.
    //   WIDE   (NARROW = 0): 4 K steps x 16 channel blocks: accumulator 4 hg + f, K step e of the slab's 4;
    //   NARROW (NARROW = 1): 16 K steps x 4 channel blocks: accumulator f, K step 4 hg + e of the slab's 16.
    // ZERO: the slab's first K step starts its accumulators (C = 0).  SI = ring stage (compile-time).
    auto run_slab = [&](auto si_tag, auto narrow_tag, auto zero_tag, auto bsrc, auto &acc, auto side) {
        constexpr int SI = decltype(si_tag)::value;
        constexpr bool NARROW = decltype(narrow_tag)::value != 0, ZERO = decltype(zero_tag)::value != 0;
        constexpr int ST_NEXT = (SI + 1) % NSTAGE;
        i32x4 seen = {0, 0, 0, 0};
#pragma unroll
        for (int hg = 0; hg < 4; ++hg) {
#pragma unroll
            for (int n = 0; n < 16; ++n) {
                const int e = n >> 2, f = n & 3;
                const int ai = NARROW ? f : 4 * hg + f, step = NARROW ? 4 * hg + e : e;
                const bool first = ZERO && step == 0;
                const f32x4 zero = {0, 0, 0, 0};
                acc[ai] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[hg & 1][f][e], bsrc(step), first ? zero : acc[ai], 0, 0, 0);
                // ---- the fillers behind this MFMA
                if (hg < 3 && n < 4) wq[(hg + 1) & 1][n] = *reinterpret_cast<const f32x4 *>(lds + SI * SLAB_FLOATS + (hg + 1) * 1024 + lane * 4 + n * 256);
                // The next slab was written a whole slab ago (before the previous barrier): its first fragments are read EARLY in the last
                // half group, away from the barrier - right behind it the four loaders write 16 KiB, and a read queued behind those stores
                // (13+ cycles each) would hold the next slab's first MFMA.
                if (HANDOVER == 1 && ABL == 0 && hg == 0 && n == 8) seen = prog_peek(0);
                if (HANDOVER == 1 && ABL == 0 && hg == 2 && n == 8) {                         // slab t + 1 written by all four loaders?
                    if (__builtin_amdgcn_readfirstlane(min4(seen)) < slab_t + 2) {
                        while (__builtin_amdgcn_readfirstlane(prog_min(0)) < slab_t + 2) __builtin_amdgcn_s_sleep(1);
                    }
                    CFENCE();
                }
                if (hg == 3) {
                    if (n < 4) wq[0][n] = *reinterpret_cast<const f32x4 *>(lds + ST_NEXT * SLAB_FLOATS + lane * 4 + n * 256);
                    if (HANDOVER == 0 && ABL == 0 && n == BARRIER_AT) wg_barrier();
                    if (HANDOVER == 1 && ABL == 0 && n == 4) { CFENCE(); prog[4 + wave] = slab_t + 1; CFENCE(); }   // every fragment read of this slab has been issued
                }
                side(hg * 16 + n);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        ++slab_t;
    };
    // a wide layer of NS slabs from ring position SI0: K step 4 M + st of slab M
    auto run_layer = [&](auto si0_tag, auto ns_tag, auto zero_tag, auto bsrc, f32x4 (&acc)[NCB], auto side) {
        constexpr int SI0 = decltype(si0_tag)::value, NS = decltype(ns_tag)::value;
        auto rec = [&](auto self, auto m_tag) {
            constexpr int M = decltype(m_tag)::value;
            if constexpr (M < NS) {
                run_slab(IC<(SI0 + M) % NSTAGE>{}, IC<0>{}, IC<(decltype(zero_tag)::value && M == 0) ? 1 : 0>{},
                         [&](int st) { return bsrc(M * 4 + st); }, acc, [&](int p) { side(M, p); });
                self(self, IC<M + 1>{});
            }
        };
        rec(rec, IC<0>{});
    };
    auto d_elem = [&](const f32x4 (&D)[NCB], int step) { return D[step >> 2][step & 3]; };
    const __amdgpu_buffer_rsrc_t rb0 = rsrc_of(a.b0), rb1 = rsrc_of(a.b1), rb2 = rsrc_of(a.b2), rx = rsrc_of(a.x);
    auto bias_quad = [&](__amdgpu_buffer_rsrc_t rs, int cb) {      // channels 16 cb + 4 g .. + 3: the lane's registers of block cb
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, 16 * g, cb * 64, 0));
    };
    f32x4 bq, bq2;                                             // bias quads in flight
    const int row = wave * 16 + j;                             // this lane's output row of the tile
    for (int tile = 0; tile < a.tiles_per_wg; ++tile) {
        const size_t t_idx = (size_t)blockIdx.x * a.tiles_per_wg + tile;
        const int x_tile = (int)(t_idx * (3 * TR) * K0 * 4);   // byte offset of the tile's operand rows
        // lane (j, g): features 16 g .. 16 g + 15 of its row (K step s = the features {s, s + 16, s + 32, s + 48}); piece i = 4 of them
        auto gather_piece = [&](int tile_off, int tap, int i) {
#if GATHER_SCALAR
            // as the product gathers: sixteen 4-byte loads per lane, the four lanes of a row on four consecutive elements
            // (the values are not the ones the weights expect: timing only)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int voff = ((3 * row + tap) * K0 + 4 * (4 * i + e) + g) * 4;
                xv[4 * i + e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, voff, tile_off, 0));
            }
#else
            const int voff = ((3 * row + tap) * K0 + 16 * g) * 4;
            const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, voff, tile_off + i * 16, 0));
            xv[4 * i] = v[0]; xv[4 * i + 1] = v[1]; xv[4 * i + 2] = v[2]; xv[4 * i + 3] = v[3];
#endif
        };
        if (tile == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) gather_piece(x_tile, 0, i);
        }
        const __amdgpu_buffer_rsrc_t ro = rsrc_of(a.out + t_idx * TR * C);
        const int o_voff = (row * C + 4 * g) * 4;              // channels 16 cb + 4 g .. + 3 of the lane's row at byte 64 cb behind it
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) DB[cb] = f32x4{0, 0, 0, 0};
        const bool more = tile + 1 < a.tiles_per_wg;
#pragma unroll 1
        for (int ts = 0; ts < 3; ++ts) {
            // taps in the order 0, 2, 1: the residual (centre) tap last - its activations stay in DA
            run_layer(IC<0>{}, IC<SL_EXP>{}, IC<1>{}, [&](int st) { return xv[st]; }, DA, [&](int M, int p) {
                if (M == SL_EXP - 1 && p == 40) bq = bias_quad(rb0, 0);
                if (M == SL_EXP - 1 && p == 41) bq2 = bias_quad(rb0, 1);
            });
            // activations (+ bias): block 0 now, block cb + 1 behind the MFMAs of the slab that reads block cb
#pragma unroll
            for (int r = 0; r < 4; ++r) DA[0][r] = lrelu(DA[0][r] + bq[r]);
            const int next_tap = ts == 0 ? 2 : ts == 1 ? 1 : 0;
            const int next_off = __builtin_amdgcn_readfirstlane(ts == 2 && more ? x_tile + 3 * TR * K0 * 4 : x_tile);
            run_layer(IC<SL_EXP % NSTAGE>{}, IC<SL_C>{}, IC<0>{}, [&](int st) { return d_elem(DA, st); }, DB, [&](int M, int p) {
                if (M + 1 < NCB) {
                    if (p == 4) { bq = bq2; if (M + 2 < NCB) bq2 = bias_quad(rb0, M + 2); }
                    if (p == 8) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) DA[M + 1][r] = lrelu(DA[M + 1][r] + bq[r]);
                    }
                }
                // the next expand_conv's operand values (unconditional: the last tap requests the next tile's first tap, the last tile its own again)
                if (M == 8 && p >= 32 && p < 36) gather_piece(next_off, next_tap, p - 32);
                if (M == SL_C - 1 && p == 40) bq = bias_quad(rb1, 0);
                if (M == SL_C - 1 && p == 41) bq2 = bias_quad(rb1, 1);
            });
        }
        // ---- the 1 x 1 convolution, four output blocks at a time over all K steps; B operand = lrelu(DB + b1), applied block by block
#pragma unroll
        for (int r = 0; r < 4; ++r) DB[0][r] = lrelu(DB[0][r] + bq[r]);
        // epilogue of block group G (from O[G & 1]): out = lrelu(O + b2) + residual, one block per filler slot
        f32x4 ob[4];                                            // the bias quads of the group whose epilogue comes next
        auto out_block = [&](int G, int f) {
            const int cb = 4 * G + f;
            const f32x4 b = ob[f];
            f32x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = lrelu(O[G & 1][f][r] + b[r]) + DA[cb][r];
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ro, o_voff, cb * 64, STORE_SC1 ? 16 : 0);
        };
        auto group_rec = [&](auto self, auto g_tag) {
            constexpr int G = decltype(g_tag)::value;
            if constexpr (G < NCB / 4) {
                auto slab_rec = [&](auto self2, auto m_tag) {
                    constexpr int M = decltype(m_tag)::value;
                    if constexpr (M < 4) {
                        constexpr int SQ = 3 * SL_TAP + G * 4 + M;
                        run_slab(IC<SQ % NSTAGE>{}, IC<1>{}, IC<M == 0 ? 1 : 0>{}, [&](int st) { return d_elem(DB, M * 16 + st); }, O[G & 1], [&](int p) {
                            // the first group activates the B operand as it goes: half group hg of slab M reads block 4 M + hg; the next block behind it
                            if (G == 0) {
                                const int hg = p >> 4, nb = 4 * M + hg + 1;
                                if (nb < NCB) {
                                    if ((p & 15) == 2) { bq = bq2; if (nb + 1 < NCB) bq2 = bias_quad(rb1, nb + 1); }
                                    if ((p & 15) == 6) {
#pragma unroll
                                        for (int r = 0; r < 4; ++r) DB[nb][r] = lrelu(DB[nb][r] + bq[r]);
                                    }
                                }
                            }
                            // the previous group's epilogue (its bias quads were requested in that group's last slab)
                            if (G > 0 && M == 0 && p >= 8 && p < 40 && (p & 7) == 2) out_block(G - 1, (p - 8) >> 3);
                            if (M == 3 && p >= 48 && p < 52) ob[p - 48] = bias_quad(rb2, 4 * G + (p - 48));
                        });
                        self2(self2, IC<M + 1>{});
                    }
                };
                slab_rec(slab_rec, IC<0>{});
                self(self, IC<G + 1>{});
            }
        };
        group_rec(group_rec, IC<0>{});
#pragma unroll
        for (int f = 0; f < 4; ++f) out_block(NCB / 4 - 1, f);
    }
    if (tid == 0 && a.clk) { a.clk[2 * blockIdx.x] = __builtin_readcyclecounter() - c_start; a.clk[2 * blockIdx.x + 1] = wall_clock64() - w_start; }
}

static float frand(unsigned &s) { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; }

int main(int argc, char **argv) {
    const int tiles_per_wg = argc > 1 ? atoi(argv[1]) : 16, nwg = 256;
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
    // ---- slabs: 16 fragments [hg][f][lane][e].  WIDE slab m of a layer: out channel 16 (4 hg + f) + (lane & 15), K step 4 m + e;
    // NARROW slab m of group G: out channel 16 (4 G + f) + (lane & 15), K step 16 m + 4 hg + e.
    // Feature of lane quarter gg = lane >> 4: gathered layer: step + 16 gg; chained layer: 16 (step >> 2) + 4 gg + (step & 3)
    std::vector<float> wsl((size_t)SLABS_PER_TILE * SLAB_FLOATS);
    auto kfeat = [](int step, int gg, bool chained) { return chained ? 16 * (step >> 2) + 4 * gg + (step & 3) : step + 16 * gg; };
    auto fill_wide = [&](int slab0, int nslab, auto weight, bool chained) {
        for (int m = 0; m < nslab; ++m)
            for (int hg = 0; hg < 4; ++hg)
                for (int f = 0; f < 4; ++f)
                    for (int l = 0; l < 64; ++l)
                        for (int e = 0; e < 4; ++e)
                            wsl[(size_t)(slab0 + m) * SLAB_FLOATS + ((hg * 4 + f) * 64 + l) * 4 + e] = weight(16 * (4 * hg + f) + (l & 15), kfeat(4 * m + e, l >> 4, chained));
    };
    auto fill_narrow = [&](int slab0, auto weight) {       // NCB / 4 block groups x 4 slabs
        for (int G = 0; G < NCB / 4; ++G)
            for (int m = 0; m < 4; ++m)
                for (int hg = 0; hg < 4; ++hg)
                    for (int f = 0; f < 4; ++f)
                        for (int l = 0; l < 64; ++l)
                            for (int e = 0; e < 4; ++e)
                                wsl[(size_t)(slab0 + G * 4 + m) * SLAB_FLOATS + ((hg * 4 + f) * 64 + l) * 4 + e] = weight(16 * (4 * G + f) + (l & 15), kfeat(16 * m + 4 * hg + e, l >> 4, true));
    };
    int s0 = 0;
    for (int tap : {0, 2, 1}) {                               // (in the order of use: the residual tap last)
        fill_wide(s0, SL_EXP, [&](int ch, int k) { return W0[ch * K0 + k]; }, false); s0 += SL_EXP;
        fill_wide(s0, SL_C, [&](int ch, int k) { return W1[(size_t)ch * 3 * C + tap * C + k]; }, true); s0 += SL_C;
    }
    fill_narrow(s0, [&](int ch, int k) { return W2[(size_t)ch * C + k]; }); s0 += SL_C;
    if (s0 != SLABS_PER_TILE) { printf("slab count\n"); return 1; }
    Args a;
    float *dx, *dw, *db0, *db1, *db2, *dout;
    (void)hipMalloc(&dx, x.size() * 4); (void)hipMalloc(&dw, NIMG * wsl.size() * 4 + 16 * SLAB_FLOATS * 4); (void)hipMalloc(&db0, C * 4); (void)hipMalloc(&db1, C * 4); (void)hipMalloc(&db2, C * 4);
    (void)hipMalloc(&dout, tiles * TR * C * 4);
    (void)hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice); for (int im = 0; im < NIMG; ++im) (void)hipMemcpy(dw + (size_t)im * wsl.size(), wsl.data(), wsl.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(db0, b0.data(), C * 4, hipMemcpyHostToDevice); (void)hipMemcpy(db1, b1.data(), C * 4, hipMemcpyHostToDevice); (void)hipMemcpy(db2, b2.data(), C * 4, hipMemcpyHostToDevice);
    long long *dclk; (void)hipMalloc(&dclk, 2 * nwg * 8); (void)hipMemset(dclk, 0, 2 * nwg * 8);
    a.clk = dclk; a.x = dx; a.wsl = dw; a.b0 = db0; a.b1 = db1; a.b2 = db2; a.out = dout; a.tiles_per_wg = tiles_per_wg;
    const int lds_bytes = NSTAGE * SLAB_FLOATS * 4 + 64;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(chain_tile), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    chain_tile<<<nwg, 512, lds_bytes>>>(a);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
    // ---- check tiles against a double-precision CPU evaluation
    std::vector<float> out(tiles * TR * C);
    (void)hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost);
    double worst = 0, ref_max = 0;
    for (size_t t : {(size_t)0, (size_t)1, tiles - 1}) {
        for (int row = 0; row < TR; row += 3) {
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
    printf("max abs error against the CPU evaluation: %.3e (|ref| max %.2f)%s\n", worst, ref_max, ABL ? "  [ablation: results are not expected to match]" : "");
    // ---- time
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 20; ++i) chain_tile<<<nwg, 512, lds_bytes>>>(a);
    (void)hipEventRecord(e0);
    const int reps = 50;
    for (int i = 0; i < reps; ++i) chain_tile<<<nwg, 512, lds_bytes>>>(a);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double us_tile = ms * 1e3 / reps / tiles_per_wg;
    {
        std::vector<long long> hc(2 * nwg);
        (void)hipMemcpy(hc.data(), dclk, hc.size() * 8, hipMemcpyDeviceToHost);
        std::vector<double> ghz;
        for (int w = 0; w < nwg; ++w) if (hc[2 * w + 1] > 0) ghz.push_back((double)hc[2 * w] / (hc[2 * w + 1] * 10.0));
        std::sort(ghz.begin(), ghz.end());
        if (!ghz.empty()) printf("shader clock during the run (cycle counter / 100 MHz wall clock, per workgroup): median %.2f GHz, %.2f .. %.2f; cycles per tile %.0f\n",
                                 ghz[ghz.size() / 2], ghz.front(), ghz.back(), us_tile * 1e3 * ghz[ghz.size() / 2]);
    }
    const double mfma = 4.0 * SLABS_PER_TILE * 64;     // per workgroup and tile
    const double flop_alg = 2.0 * TR * (3.0 * K0 * C + 3.0 * C * C + (double)C * C);
    printf("STAGGER=%d GATHER_SCALAR=%d STORE_SC1=%d NIMG=%d HANDOVER=%d LOADER_DMA=%d LOADER_ABL=%d ABL=%d LOAD_AHEAD=%d: %d tiles of %d rows per workgroup, 256 workgroups: %.1f us per tile of 64 rows (the product's 64-row tile: ~79 us, ~88 in its timing build)\n",
           STAGGER, GATHER_SCALAR, STORE_SC1, NIMG, HANDOVER, LOADER_DMA, LOADER_ABL, ABL, LOAD_AHEAD, tiles_per_wg, TR, us_tile);
    printf("MFMA issue bound (%.0f x 16x16x4 per tile, 32 cycles each, 4 SIMDs at 2.4 GHz): %.1f us per tile -> %.2f of it; algorithmic %.1f TFLOP/s of 157.3 chip-wide\n",
           mfma, mfma * 32 / 4 / 2.4e3, mfma * 32 / 4 / 2.4e3 / us_tile, flop_alg * 256 / us_tile / 1e6);
    return 0;
}
