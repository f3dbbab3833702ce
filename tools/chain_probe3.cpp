// Register-chained first-level tile, third form (round 6) - a PROBE, not product code.
//
// chain_probe2.cpp with what its first measurement (79.4 us per 64 rows; 73.7 without staging and counters) pointed at:
//  * hand-over of a weight slab: SYNC=0 the two LDS counters per ring stage of chain_probe2, SYNC=1 ONE s_barrier per slab (four
//    wavefronts, one per SIMD: no SIMD partner is re-aligned by it; a slab is written two slabs before it is read, so a
//    barrier per slab orders both the refill of a stage and the read of the next one);
//  * no bias loads in front of a layer: a layer's first K step takes C = 0 (an inline constant of the MFMA), the bias is added
//    where the activation is applied, behind MFMAs;
//  * the 1 x 1 convolution is OUTPUT-stationary: two channel blocks of the output at a time over all 128 K steps (two
//    accumulators, 32 registers), so that the epilogue of a block pair (+ bias, activation, + residual, store) rides behind the
//    next pair's MFMAs and the level needs 64 instead of 128 output registers: the centre tap's activations (the residual) stay
//    in their registers (taps in the order 0, 2, 1), nothing is parked in memory;
//  * the operand values of the next expand_conv are requested sixteen slabs ahead instead of two.
// Tile, layouts and the MFMA operand mapping: chain_probe2.cpp.
// build: hipcc -O3 --offload-arch=gfx950 tools/chain_probe3.cpp -o tools/chain_probe3.bin   (-DSYNC=0|1, -DABL=1|2)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int C = 256, K0 = 64, NCB = C / 32, TR = 128, NW = 4;
#ifndef ABL
#define ABL 0      // ablations (timing only): 1 no staging and no hand-over (the ring is never refilled), 2 staging without hand-over
#endif
#ifndef SYNC
#define SYNC 1
#endif
#ifndef XV_AGPR
#define XV_AGPR 0
#endif
constexpr int SLAB_FLOATS = 4096;                                            // 16 KiB: 64 MFMAs per wavefront
constexpr int NSTAGE = 4, AHEAD = 2;
constexpr int SL_EXP = K0 / 2 / 8, SL_C = C / 2 / 8;                         // 4, 16 slabs (a "wide" slab: 8 K steps x 8 channel blocks)
constexpr int SL_TAP = SL_EXP + SL_C, SLABS_PER_TILE = 3 * SL_TAP + SL_C;    // 20, 76 (a "narrow" slab: 32 K steps x 2 channel blocks)
static_assert(SL_TAP % NSTAGE == 0 && SLABS_PER_TILE % NSTAGE == 0, "a slab's ring stage is a compile-time constant");
constexpr float SLOPE = 0.2f;

__device__ __forceinline__ float lrelu(float v) { return __builtin_fmaxf(v, v * SLOPE); }
#define CFENCE() __atomic_signal_fence(__ATOMIC_SEQ_CST)     // compiler-only ordering of LDS data accesses against the hand-over

struct Args {
    const float *x;        // [tiles][3 * TR expand rows][K0]
    const float *wsl;      // SLABS_PER_TILE slabs
    const float *b0, *b1, *b2;
    float *out;            // [tiles][TR][C]
    int tiles_per_wg;
};

template <int V> using IC = std::integral_constant<int, V>;

__global__ __launch_bounds__(256) void chain_tile(const Args a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    unsigned *ctr = reinterpret_cast<unsigned *>(lds + NSTAGE * SLAB_FLOATS);      // ready[0..NSTAGE), done[NSTAGE..2 NSTAGE)
    if (tid < 2 * NSTAGE) ctr[tid] = 0u;
    __syncthreads();
    auto rsrc_of = [](const float *p) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p), 0, 0x7fffffff, 0x00020000); };
    // ---- this wavefront's quarter of a slab: global -> registers -> ring stage
    f32x4 stg[4];
    const __amdgpu_buffer_rsrc_t rw = rsrc_of(a.wsl);
    const int w_voff = (wave * 1024 + lane * 4) * 4;
    float *wdst = lds + wave * 1024 + lane * 4;
    auto ctr_add = [&](int idx) {
        CFENCE();
        if (lane == 0) __hip_atomic_fetch_add(ctr + idx, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        CFENCE();
    };
    auto ctr_peek = [&](int idx) { return __hip_atomic_load(ctr + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    auto ctr_wait = [&](unsigned seen, int idx, unsigned need) {    // `seen`: a value of the counter read a while ago
        if (__builtin_amdgcn_readfirstlane(seen) < need) {
            while (__builtin_amdgcn_readfirstlane(ctr_peek(idx)) < need) __builtin_amdgcn_s_sleep(1);
        }
        CFENCE();
    };
    auto wg_barrier = [&]() {                                   // LDS traffic of this wavefront done; everyone here (no vmcnt wait)
        CFENCE();
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        CFENCE();
    };
    unsigned gen = 0;                                           // times every ring stage has been filled before the current group of NSTAGE slabs
    for (int p = 0; p < AHEAD; ++p) {                           // prologue: slabs 0 .. AHEAD-1 of the sequence
#pragma unroll
        for (int i = 0; i < 4; ++i) stg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, w_voff, p * SLAB_FLOATS * 4 + i * 1024, 0));
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4 *>(wdst + p * SLAB_FLOATS + i * 256) = stg[i];
        if (SYNC == 0) ctr_add(p);
    }
    f32x4 wq[2][4];                                             // weight fragments of the current / the next half group
    if (SYNC == 0 && ABL != 1) ctr_wait(0u, 0, NW);
    else __syncthreads();
#pragma unroll
    for (int f = 0; f < 4; ++f) wq[0][f] = *reinterpret_cast<const f32x4 *>(lds + lane * 4 + f * 256);

    f32x16 DA[NCB], DB[NCB], O[2][2];    // expand_conv activations (the residual after the last tap) | 3-tap sums | 1 x 1 outputs of two block pairs
    float xv[K0 / 2];                    // this lane's operand values of the next expand_conv
    // One slab of 64 MFMAs in 4 half groups of 16 (one 16-byte fragment read per lane = the A operands of 4 MFMAs).
    //   WIDE   (NARROW = 0): 8 K steps x 8 channel blocks; half group hg = (step quad hg >> 1, block half hg & 1), MFMA n of it:
    //          element e = n >> 2 of fragment f = n & 3 -> accumulator 4 (hg & 1) + f, K step 4 (hg >> 1) + e of the slab's 8;
    //   NARROW (NARROW = 1): 32 K steps x 2 channel blocks; MFMA n of half group hg: block n & 1, element e = (n >> 1) & 3 of fragment
    //          (n & 1) + 2 (n >> 3) -> accumulator n & 1, K step 8 hg + 4 (n >> 3) + e of the slab's 32.
    // ZERO: the slab's first K step starts its accumulators (C = 0).  SI = ring stage (compile-time), seq = slab of the sequence
    // (run-time: where slab seq + AHEAD lies).  Everything that is not an MFMA is pinned behind ONE MFMA (position p = 16 hg + n):
    // with one wavefront per SIMD a run of non-matrix instructions longer than an MFMA (64 cycles) is a bubble in the matrix pipe.
    auto run_slab = [&](auto si_tag, auto narrow_tag, auto zero_tag, const int seq, auto bsrc, auto &acc, auto side) {
        constexpr int SI = decltype(si_tag)::value;
        constexpr bool NARROW = decltype(narrow_tag)::value != 0, ZERO = decltype(zero_tag)::value != 0;
        constexpr int ST_NEXT = (SI + 1) % NSTAGE, ST_FILL = (SI + AHEAD) % NSTAGE;
        const unsigned gen_next = gen + (SI + 1 >= NSTAGE ? 1u : 0u), gen_fill = gen + (SI + AHEAD >= NSTAGE ? 1u : 0u);
        unsigned seen_ready = 0, seen_done = 0;
        int src_off;                                                // byte offset of slab seq + AHEAD of the sequence (wraps)
        {
            int m = seq + AHEAD;
            if (m >= SLABS_PER_TILE) m -= SLABS_PER_TILE;
            src_off = __builtin_amdgcn_readfirstlane(m * SLAB_FLOATS * 4);
        }
        float *fill = wdst + ST_FILL * SLAB_FLOATS;
#pragma unroll
        for (int hg = 0; hg < 4; ++hg) {
#pragma unroll
            for (int n = 0; n < 16; ++n) {
                const int e = NARROW ? (n >> 1) & 3 : n >> 2, f = NARROW ? (n & 1) + 2 * (n >> 3) : n & 3;
                const int ai = NARROW ? n & 1 : 4 * (hg & 1) + f, step = NARROW ? 8 * hg + 4 * (n >> 3) + e : 4 * (hg >> 1) + e;
                const bool first = ZERO && step == 0;
                const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                acc[ai] = __builtin_amdgcn_mfma_f32_32x32x2f32(wq[hg & 1][f][e], bsrc(step), first ? zero : acc[ai], 0, 0, 0);
                // ---- the fillers behind this MFMA
                if (hg == 0 && n >= 4 && n < 8 && ABL != 1)
                    stg[n - 4] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, w_voff, src_off + (n - 4) * 1024, 0));
                if (hg < 3 && n < 4) wq[(hg + 1) & 1][n] = *reinterpret_cast<const f32x4 *>(lds + SI * SLAB_FLOATS + (hg + 1) * 1024 + lane * 4 + n * 256);
                if (SYNC == 0 && ABL == 0) {
                    if (hg == 1 && n == 8) seen_ready = ctr_peek(ST_NEXT);
                    if (hg == 1 && n == 9) seen_done = ctr_peek(NSTAGE + ST_FILL);
                }
                if (hg == 3) {
                    if (SYNC == 0 && ABL == 0) {
                        if (n == 0) ctr_add(NSTAGE + SI);                // every fragment read of this slab has been issued
                        if (n == 1) ctr_wait(seen_done, NSTAGE + ST_FILL, NW * gen_fill);
                    }
                    if (n >= 2 && n < 6 && ABL != 1) *reinterpret_cast<f32x4 *>(fill + (n - 2) * 256) = stg[n - 2];
                    if (SYNC == 0 && ABL == 0) {
                        if (n == 6) ctr_add(ST_FILL);
                        if (n == 7) ctr_wait(seen_ready, ST_NEXT, NW * (gen_next + 1));
                    }
                    if (SYNC == 1 && ABL == 0 && n == 7) wg_barrier();
                    if (n >= 8 && n < 12) wq[0][n - 8] = *reinterpret_cast<const f32x4 *>(lds + ST_NEXT * SLAB_FLOATS + lane * 4 + (n - 8) * 256);
                }
                side(hg * 16 + n);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (SI == NSTAGE - 1) ++gen;
    };
    // a wide layer of NS slabs from ring position SI0 (compile-time), sequence index seq0 (run-time): K step 8 M + st of slab M
    auto run_layer = [&](auto si0_tag, auto ns_tag, auto zero_tag, const int seq0, auto bsrc, f32x16 (&acc)[NCB], auto side) {
        constexpr int SI0 = decltype(si0_tag)::value, NS = decltype(ns_tag)::value;
        auto rec = [&](auto self, auto m_tag) {
            constexpr int M = decltype(m_tag)::value;
            if constexpr (M < NS) {
                run_slab(IC<(SI0 + M) % NSTAGE>{}, IC<0>{}, IC<(decltype(zero_tag)::value && M == 0) ? 1 : 0>{}, seq0 + M,
                         [&](int st) { return bsrc(M * 8 + st); }, acc, [&](int p) { side(M, p); });
                self(self, IC<M + 1>{});
            }
        };
        rec(rec, IC<0>{});
    };
    auto d_elem = [&](const f32x16 (&D)[NCB], int step) { return D[step >> 4][step & 15]; };
    const __amdgpu_buffer_rsrc_t rb0 = rsrc_of(a.b0), rb1 = rsrc_of(a.b1), rb2 = rsrc_of(a.b2), rx = rsrc_of(a.x);
    auto bias_quad = [&](__amdgpu_buffer_rsrc_t rs, int cb, int g) {      // channels 32 cb + 8 g + 4 h .. + 3: registers 4 g .. 4 g + 3 of block cb
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, 16 * h, (cb * 32 + 8 * g) * 4, 0));
    };
    // activation of one register quad of a block, its bias requested one filler slot earlier (bq: the quad in flight)
    f32x4 bq;
    auto act_quad = [&](f32x16 &D) { return 0; };
    (void)act_quad;
    const int row = wave * 32 + j;                             // this lane's output row of the tile
    for (int tile = 0; tile < a.tiles_per_wg; ++tile) {
        const size_t t_idx = (size_t)blockIdx.x * a.tiles_per_wg + tile;
        const int x_tile = (int)(t_idx * (3 * TR) * K0 * 4);   // byte offset of the tile's operand rows
        // lane (j, h): features 32 h .. 32 h + 31 of its row (K step s = the pair {s, s + 32}); piece i = features 4 i .. 4 i + 3 of those
        auto gather_piece = [&](int tile_off, int tap, int i) {
            const int voff = ((3 * row + tap) * K0 + 32 * h) * 4;
            const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, voff, tile_off + i * 16, 0));
            xv[4 * i] = v[0]; xv[4 * i + 1] = v[1]; xv[4 * i + 2] = v[2]; xv[4 * i + 3] = v[3];
#if XV_AGPR
            // parked in AccVGPRs until the expand_conv that reads them (the ArchVGPRs hold the activations / the residual)
            asm volatile("" : "+a"(xv[4 * i]), "+a"(xv[4 * i + 1]), "+a"(xv[4 * i + 2]), "+a"(xv[4 * i + 3]));
#endif
        };
        if (tile == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) gather_piece(x_tile, 0, i);
        }
        const __amdgpu_buffer_rsrc_t ro = rsrc_of(a.out + t_idx * TR * C);
        const int o_voff = (row * C + 4 * h) * 4;              // channels 32 cb + 8 g + 4 h .. + 3 of the lane's row at byte (32 cb + 8 g) * 4 behind it
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) DB[cb][r] = 0.0f;
        const bool more = tile + 1 < a.tiles_per_wg;
#pragma unroll 1
        for (int ts = 0; ts < 3; ++ts) {
            // taps in the order 0, 2, 1: the residual (centre) tap last - its activations stay in DA
            const int seq0 = ts * SL_TAP;                      // (the slab sequence is stored in the order of use)
            run_layer(IC<0>{}, IC<SL_EXP>{}, IC<1>{}, seq0, [&](int st) { return xv[st]; }, DA, [&](int M, int p) {
                if (M == SL_EXP - 1 && p == 60) bq = bias_quad(rb0, 0, 0);
            });
            // activations (+ bias): block 0 now, block cb + 1 behind the MFMAs of the two slabs that read block cb
            auto act_da = [&](int cb, int g, int ncb, int ng) {     // quad (cb, g) with the bias in flight; then request the next quad's
                const f32x4 b = bq;
                if (ncb < NCB) bq = bias_quad(rb0, ncb, ng);
#pragma unroll
                for (int r = 0; r < 4; ++r) DA[cb][4 * g + r] = lrelu(DA[cb][4 * g + r] + b[r]);
            };
            act_da(0, 0, 0, 1); act_da(0, 1, 0, 2); act_da(0, 2, 0, 3); act_da(0, 3, 1, 0);
            const int next_tap = ts == 0 ? 2 : ts == 1 ? 1 : 0;
            const int next_off = __builtin_amdgcn_readfirstlane(ts == 2 && more ? x_tile + 3 * TR * K0 * 4 : x_tile);
            run_layer(IC<SL_EXP % NSTAGE>{}, IC<SL_C>{}, IC<0>{}, seq0 + SL_EXP, [&](int st) { return d_elem(DA, st); }, DB, [&](int M, int p) {
                // slab M reads block M / 2; block M / 2 + 1 is activated behind it: quads 2 (M & 1), 2 (M & 1) + 1
                if ((p == 16 || p == 24) && M / 2 + 1 < NCB) {
                    const int cb = M / 2 + 1, g = 2 * (M & 1) + (p == 24), nx = 4 * cb + g + 1;
                    act_da(cb, g, nx >> 2, nx & 3);
                }
                // the next tap's operand values
                // (half way through the layer: the blocks already read are dead by then - unless they are the residual - and free the registers)
                // (unconditional - a branch here would end in a wait for the load: the last tap requests the next tile's first tap, the last tile its own again)
                if ((M == 8 || M == 9) && p >= 32 && p < 36) gather_piece(next_off, next_tap, (M - 8) * 4 + (p - 32));
            });
        }
        // ---- the 1 x 1 convolution, two output blocks at a time over all K steps; B operand = lrelu(DB + b1), applied block by block
        auto act_db = [&](int cb, int g, int ncb, int ng) {
            const f32x4 b = bq;
            if (ncb < NCB) bq = bias_quad(rb1, ncb, ng);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float t = lrelu(DB[cb][4 * g + r] + b[r]);
                asm volatile("" : "+a"(t));            // stays an AccVGPR (the MFMA reads its B operand from there): the ArchVGPRs hold the residual
                DB[cb][4 * g + r] = t;
            }
        };
        bq = bias_quad(rb1, 0, 0);
        act_db(0, 0, 0, 1); act_db(0, 1, 0, 2); act_db(0, 2, 0, 3); act_db(0, 3, 1, 0);
        act_db(1, 0, 1, 1); act_db(1, 1, 1, 2); act_db(1, 2, 1, 3); act_db(1, 3, 2, 0);
        // epilogue of block pair bp (from O[bp & 1]): out = lrelu(O + b2) + residual, one register quad per filler slot
        auto out_quad = [&](int bp, int q) {                   // q = 0..7: block 2 bp + (q >> 2), quad q & 3
            const int cb = 2 * bp + (q >> 2), g = q & 3;
            const f32x4 b = bias_quad(rb2, cb, g);
            f32x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = lrelu(O[bp & 1][q >> 2][4 * g + r] + b[r]) + DA[cb][4 * g + r];
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ro, o_voff, (cb * 32 + 8 * g) * 4, 0);
        };
        auto pair_rec = [&](auto self, auto bp_tag) {
            constexpr int BP = decltype(bp_tag)::value;
            if constexpr (BP < NCB / 2) {
                auto slab_rec = [&](auto self2, auto m_tag) {
                    constexpr int M = decltype(m_tag)::value;
                    if constexpr (M < 4) {
                        constexpr int SQ = 3 * SL_TAP + BP * 4 + M;
                        run_slab(IC<SQ % NSTAGE>{}, IC<1>{}, IC<M == 0 ? 1 : 0>{}, SQ, [&](int st) { return d_elem(DB, M * 32 + st); }, O[BP & 1], [&](int p) {
                            // the first pair activates the B operand as it goes: slab M reads blocks 2 M, 2 M + 1; blocks 2 M + 2, 2 M + 3 behind it
                            if (BP == 0 && M < 3 && p >= 8 && p < 40 && (p & 3) == 0) {
                                const int q = (p - 8) >> 2, cb = 2 * M + 2 + (q >> 2), g = q & 3, nx = 4 * cb + g + 1;
                                act_db(cb, g, nx >> 2, nx & 3);
                            }
                            // the previous pair's epilogue
                            if (BP > 0 && M == 0 && p >= 8 && p < 40 && (p & 3) == 2) out_quad(BP - 1, (p - 8) >> 2);
                        });
                        self2(self2, IC<M + 1>{});
                    }
                };
                slab_rec(slab_rec, IC<0>{});
                self(self, IC<BP + 1>{});
            }
        };
        pair_rec(pair_rec, IC<0>{});
#pragma unroll
        for (int q = 0; q < 8; ++q) out_quad(NCB / 2 - 1, q);
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
    // ---- slabs.  WIDE element [slab][hg = 2 sq + bh][f][lane][e]: out channel 32 (4 bh + f) + (lane & 31), K step 8 slab + 4 sq + e;
    // NARROW element [slab][hg][f = blk + 2 sqq][lane][e]: out channel 32 (2 pair + blk) + (lane & 31), K step 32 slab + 8 hg + 4 sqq + e.
    // Feature of lane half hh = lane >> 5: gathered layer: step + 32 hh; chained layer: 32 cb + (r & 3) + 8 (r >> 2) + 4 hh, (cb, r) = (step >> 4, step & 15)
    std::vector<float> wsl((size_t)SLABS_PER_TILE * SLAB_FLOATS);
    auto kfeat = [](int step, int hh, bool chained) {
        const int r = step & 15, cb = step >> 4;
        return chained ? 32 * cb + (r & 3) + 8 * (r >> 2) + 4 * hh : step + 32 * hh;
    };
    auto fill_wide = [&](int slab0, int nslab, auto weight, bool chained) {
        for (int m = 0; m < nslab; ++m)
            for (int hg = 0; hg < 4; ++hg)
                for (int f = 0; f < 4; ++f)
                    for (int l = 0; l < 64; ++l)
                        for (int e = 0; e < 4; ++e) {
                            const int step = m * 8 + 4 * (hg >> 1) + e, ch = 32 * (4 * (hg & 1) + f) + (l & 31);
                            wsl[(size_t)(slab0 + m) * SLAB_FLOATS + ((hg * 4 + f) * 64 + l) * 4 + e] = weight(ch, kfeat(step, l >> 5, chained));
                        }
    };
    auto fill_narrow = [&](int slab0, auto weight) {       // NCB / 2 block pairs x 4 slabs
        for (int bp = 0; bp < NCB / 2; ++bp)
            for (int m = 0; m < 4; ++m)
                for (int hg = 0; hg < 4; ++hg)
                    for (int f = 0; f < 4; ++f)
                        for (int l = 0; l < 64; ++l)
                            for (int e = 0; e < 4; ++e) {
                                const int step = m * 32 + hg * 8 + 4 * (f >> 1) + e, ch = 32 * (2 * bp + (f & 1)) + (l & 31);
                                wsl[(size_t)(slab0 + bp * 4 + m) * SLAB_FLOATS + ((hg * 4 + f) * 64 + l) * 4 + e] = weight(ch, kfeat(step, l >> 5, true));
                            }
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
    (void)hipMalloc(&dx, x.size() * 4); (void)hipMalloc(&dw, wsl.size() * 4); (void)hipMalloc(&db0, C * 4); (void)hipMalloc(&db1, C * 4); (void)hipMalloc(&db2, C * 4);
    (void)hipMalloc(&dout, tiles * TR * C * 4);
    (void)hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dw, wsl.data(), wsl.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(db0, b0.data(), C * 4, hipMemcpyHostToDevice); (void)hipMemcpy(db1, b1.data(), C * 4, hipMemcpyHostToDevice); (void)hipMemcpy(db2, b2.data(), C * 4, hipMemcpyHostToDevice);
    a.x = dx; a.wsl = dw; a.b0 = db0; a.b1 = db1; a.b2 = db2; a.out = dout; a.tiles_per_wg = tiles_per_wg;
    const int lds_bytes = NSTAGE * SLAB_FLOATS * 4 + 64;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(chain_tile), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    chain_tile<<<nwg, 256, lds_bytes>>>(a);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
    // ---- check tiles against a double-precision CPU evaluation
    std::vector<float> out(tiles * TR * C);
    (void)hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost);
    double worst = 0, ref_max = 0;
    for (size_t t : {(size_t)0, (size_t)1, tiles - 1}) {
        for (int row = 0; row < TR; row += 5) {
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
    for (int i = 0; i < 20; ++i) chain_tile<<<nwg, 256, lds_bytes>>>(a);
    (void)hipEventRecord(e0);
    const int reps = 50;
    for (int i = 0; i < reps; ++i) chain_tile<<<nwg, 256, lds_bytes>>>(a);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double us_tile = ms * 1e3 / reps / tiles_per_wg;
    const double mfma = 4.0 * SLABS_PER_TILE * 64;     // per workgroup and tile
    const double flop_alg = 2.0 * TR * (3.0 * K0 * C + 3.0 * C * C + (double)C * C);
    printf("SYNC=%d ABL=%d: %d tiles of %d rows per workgroup, 256 workgroups: %.1f us per tile (%.1f us per 64 rows; the product's 64-row tile: ~79 us, ~88 in its timing build)\n",
           SYNC, ABL, tiles_per_wg, TR, us_tile, us_tile / 2);
    printf("MFMA issue bound (%.0f x 32x32x2 per tile, 64 cycles each, 4 SIMDs at 2.4 GHz): %.1f us per tile -> %.2f of it; algorithmic %.1f TFLOP/s of 157.3 chip-wide\n",
           mfma, mfma * 64 / 4 / 2.4e3, mfma * 64 / 4 / 2.4e3 / us_tile, flop_alg * 256 / us_tile / 1e6);
    return 0;
}
