// first_level_chain: the fused first level of a body-part branch (expand_conv on the gathered input, the level-1 3-tap
// convolution and its 1x1 convolution, lib/model/rie.py:85-97) as a REGISTER-CHAINED tile of 64 output rows.
// Included by r3d_tiles.hpp (inside namespace r3d); probes: tools/chain_probe{2,3,4}.cpp, profiles/r06_chain_probe/.
//
// first_level_taps hands every layer's activations over through LDS (accumulator layout -> A-operand layout, a transposition)
// with two workgroup barriers per hand-over and all eight wavefronts in every phase: its 64-row tile is 0.82 matrix-busy.
// Here the product is computed TRANSPOSED - weights as the MFMA's A operand, activations as its B operand - so that a layer's
// accumulators ARE the next layer's B operands and no activation touches LDS:
//   v_mfma_f32_16x16x4_f32   D[i][j] += sum_k A[i][k] B[k][j];  lane l: A[i = l & 15][k = l >> 4], B[k = l >> 4][j = l & 15],
//                            D[i = 4 (l >> 4) + r][j = l & 15], r = 0..3        i = channel, j = row, k = input feature
// Register r of channel block cb (16 channels) of a layer's accumulators is the B operand of the next layer's K step (cb, r) =
// the features {16 cb + r, + 4, + 8, + 12}; r3d_finalize packs the chained layers' weights in that order (r3d_model.cpp,
// pack_chain).  The workgroup's wavefronts 0-3 - one per SIMD - are the MFMA wavefronts: 16 rows x 256 channels each, accumulator
// sets of 64 registers (expand_conv activations DA, the 3-tap sums DB, 2 x 16 output registers).  Wavefronts 4-7, their SIMD
// partners, are LOADERS: the weights (1.19 MB per tile, the same for the four MFMA wavefronts) stream global -> LDS by
// buffer_load ... lds (no VGPR, no ds_write: the LDS stores of a VGPR-staged loader cost 2.6 us per tile, the direct form 0.7)
// into a ring of six 16 KiB slabs (64 MFMAs per MFMA wavefront); ONE s_barrier per slab, passed by all eight wavefronts, hands
// a slab over: slab t + 2 has landed before barrier t, slab t + 1 is read after barrier t - 1; three more slabs are in flight.
// The taps are visited in the order (0, 3 - res_tap, res_tap): the residual tap last, its activations stay in DA for the epilogue.
// The 1x1 convolution is output-stationary (four channel blocks at a time over all K steps), so that a block group's epilogue
// (+ bias, activation, + residual, 16-byte row stores) rides behind the next group's MFMAs.  Everything that is not an MFMA is
// pinned behind ONE MFMA of a slab (position p = 16 hg + n): with one MFMA wavefront per SIMD a run of other instructions longer
// than an MFMA (32 cycles) is a bubble in the matrix pipe.  Biases are added where the activation is applied (a layer's first K
// step takes C = 0), so the summation order differs from first_level_taps': same values to fp32 rounding.
// Measured stand-alone (tools/chain_probe4.cpp -DLOADER_DMA=1): 71.2 us per 64-row tile against ~79 for first_level_taps.
#pragma once

#ifndef R3D_CHAIN_V
#define R3D_CHAIN_V 2           // 0: round 6's first form; 1: + no vmcnt wait with a write-through store pending; 2: + bias quads two half groups ahead
#endif
#ifndef R3D_CHAIN_ABL
#define R3D_CHAIN_ABL 0         // ablations (wrong results, timing only): 1 loaders request nothing, 2 no output stores, 4 no gathers, 8 no bias loads
#endif
// s_waitcnt vmcnt(0) as an INSTRUCTION THE COMPILER SEES (its wait-count pass clears its scoreboard; an asm string would not):
// gfx9 encoding vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[15:14]
#define CHAIN_WAIT_VM0() __builtin_amdgcn_s_waitcnt(0x0F70)
constexpr int CHAIN_SLAB_FLOATS = 4096;                     // 16 KiB: 16 fragments of 1 KiB
constexpr int CHAIN_NSTAGE = 6;                             // ring stages: one being read, one ahead of it, one landed, three in flight (96 KiB)
constexpr int CHAIN_NCB = 16;                               // channel blocks of 16: C = 256
constexpr int CHAIN_LUT_OFF = FLT_LUT_OFF;                  // the gather tables where first_level_taps keeps them (`new_prob` then means the same for both)
static_assert(CHAIN_LUT_OFF >= CHAIN_NSTAGE * CHAIN_SLAB_FLOATS, "the tables lie behind the ring");

template <int V> using ChainIC = std::integral_constant<int, V>;

// SL_EXP: slabs of expand_conv = K0 / 16 (4: the body-part branches, K0 = 64)
template <int SL_EXP>
__device__ __forceinline__ void first_level_chain(ProbRef P, const int4 *tile_list, const int tstride, const int ntiles, const bool new_prob,
                                                  float *smem, const gu32 cnt, long long *dbg_base) {
    constexpr int NCB = CHAIN_NCB, NSTAGE = CHAIN_NSTAGE, SLAB_FLOATS = CHAIN_SLAB_FLOATS;
    constexpr int SL_C = NCB, SL_TAP = SL_EXP + SL_C, SLABS_PER_TILE = 3 * SL_TAP + SL_C, K0 = 16 * SL_EXP;
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    auto wg_barrier = [&]() {                                   // this wavefront's LDS traffic done; everyone here (no vmcnt wait)
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
    };
    int *lut_lds = reinterpret_cast<int *>(smem + CHAIN_LUT_OFF);
    if (new_prob) {                                             // (the tables where first_level_taps expects them: 32-row tiles of this problem may follow)
        for (int i = tid; i < K0 + K0 / 4; i += GEMM_THREADS) lut_lds[i] = *(const R3D_AS1 int *)(P.lut + i);
    }
    // ... and this tile's own copy of the element offsets, lane-quarter-major: perm[g][i] = offset of column 4 i + g - the column whose
    // value lane quarter g holds in operand register i.  Four consecutive columns (one chunk: same base) belong to the four lanes of
    // a row, so a gather instruction touches 16 short runs of addresses instead of 64 single ones.
    int *perm_lds = lut_lds + FL_LUT_INTS;
    if (tid < K0) perm_lds[(tid & 3) * (K0 / 4) + (tid >> 2)] = *(const R3D_AS1 int *)(P.lut + tid);
    __syncthreads();                                            // (also: the previous tile is done with LDS)
    const int total = ntiles * SLABS_PER_TILE;                  // slabs of this run; barrier t ends slab t
    if (wave >= 4) {
        // ---------------------------------------------------------------- loader: a quarter of every slab, global -> LDS
        // The stream of one tile: [expand_conv | tap (0)] [expand_conv | tap (1)] [expand_conv | tap (2)] [1x1], taps in the order
        // of use; the image holds expand_conv once: [expand_conv: SL_EXP][taps: 3 x 16][1x1: 16] slabs.
        const int lw = wave - 4;
        const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.wchain), 0, (SL_EXP + 4 * SL_C) * SLAB_FLOATS * 4, 0x00020000);
        int pos = 0, stage = 0;                                 // position of the next slab to request in its tile, its ring stage
        auto dma = [&]() {
            int s;                                              // slab of the image
            if (pos < 3 * SL_TAP) {
                const int ts = pos >= 2 * SL_TAP ? 2 : pos >= SL_TAP ? 1 : 0, q = pos - ts * SL_TAP;
                s = q < SL_EXP ? q : SL_EXP + ts * SL_C + (q - SL_EXP);
            } else {
                s = SL_EXP + 3 * SL_C + (pos - 3 * SL_TAP);
            }
            const int off = __builtin_amdgcn_readfirstlane(s * SLAB_FLOATS * 4);
            float *d = smem + stage * SLAB_FLOATS + lw * 1024;
#if !(R3D_CHAIN_ABL & 1)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void *)(d + i * 256), 16, lane * 16,
                                                         off + (lw * 1024 + i * 256) * 4, 0, 0);
#else
            (void)off; (void)d;
#endif
            pos = pos + 1 == SLABS_PER_TILE ? 0 : pos + 1;
            stage = stage + 1 == NSTAGE ? 0 : stage + 1;
        };
        // Slab m lives in stage m % NSTAGE.  Slabs 0 .. NSTAGE - 2 requested, 0 and 1 landed before the opening barrier; then slab
        // t + NSTAGE - 1 is requested between barriers t - 1 and t - into the stage slab t - 1 has just left - and vmcnt says slab
        // t + 2 has landed: a request has NSTAGE - 3 slab times (2.5 us) to land.  With four stages it had ONE, less than an L2
        // round trip under load: every barrier of the tile waited 0.15 us for the loaders (11 us per tile, in-tile stamps of the
        // timing build: profiles/r06_chain_probe/).  Nothing is requested past the run's last slab: the ring belongs to the next tile kind once
        // this function returns.
        static_assert(NSTAGE == 6, "the vmcnt immediates below are for three slabs (12 pieces) in flight");
#pragma unroll
        for (int i = 0; i < NSTAGE - 1; ++i) dma();
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");      // slabs 0, 1 landed
        wg_barrier();                                           // opening barrier
        int in_tile = 0;
        for (int t = 0; t < total; ++t) {
            const int left = total - 1 - (t + 2);               // slabs behind t + 2 that have been (or are now) requested and may stay in flight
            if (t + NSTAGE - 1 < total) {
                dma();
                asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            } else if (left == 2) {
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            } else if (left == 1) {
                asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            wg_barrier();
            if (++in_tile == SLABS_PER_TILE) {                  // the MFMA wavefronts' end-of-tile barrier (stores drained -> ready counters)
                in_tile = 0;
                wg_barrier();
            }
        }
        return;
    }
    // -------------------------------------------------------------------- MFMA wavefront: 16 rows x 256 channels
    const int j = lane & 15, g = lane >> 4;
    const int M = P.M, res_tap = P.res_tap;
    const float slope0 = P.slope, slope1 = P.slope2, slope2 = P.slope3;
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.x), 0, P.enc_bytes, 0x00020000);
    auto rsrc_of = [](const float *p) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p), 0, 0x7fffffff, 0x00020000); };
    const __amdgpu_buffer_rsrc_t rb0 = rsrc_of(P.bias), rb1 = rsrc_of(P.bias2), rb2 = rsrc_of(P.bias3);
    auto bias_quad = [&](__amdgpu_buffer_rsrc_t rs, int cb) {      // channels 16 cb + 4 g .. + 3: the lane's registers of block cb
#if R3D_CHAIN_ABL & 8
        return f32x4{0.f, 0.f, 0.f, 0.f};
#else
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, 16 * g, cb * 64, 0));
#endif
    };
    f32x4 wq[2][4];                                             // weight fragments of the current / the next half group
    f32x4 DA[NCB], DB[NCB], O[2][4];     // expand_conv activations (the residual after the last tap) | 3-tap sums | 1x1 outputs of two block groups
    float xv[K0 / 4];                    // this lane's operand values of the next expand_conv: columns 4 i + g
    // ---- gather state: the raw element of column k of expand_conv row 3 r + tap lies lut1[k] bytes behind the row's first frame -
    // or the window's current frame when lutk[k / 4] says so (r3d_internal.hpp, ENC_INVALID: past the descriptor's bound, reads 0)
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    unsigned b_first = 0, b_cur = 0;     // byte offsets of the lane's row: first frame of tap 0's triple, current frame of its window
    auto row_bases = [&](int row0) {
        const int orow = row0 + wave * 16 + j;
        const int e = 3 * (orow < M ? orow : M - 1);
        const int win = e / P.enc_rows, t3 = e - win * P.enc_rows;          // (enc_rows is a multiple of 3: the triple stays in its window)
        const unsigned wbase = (unsigned)win * (unsigned)P.enc_ws;
        b_first = (wbase + (unsigned)(t3 * 3 * P.enc_jf)) * 4u;
        b_cur = (wbase + (unsigned)P.enc_cur) * 4u;
    };
    i32x4 lq[K0 / 16];                   // the lane's element offsets (columns 4 i + g), re-read from LDS for every gather (16 registers for a moment)
    auto lut_fetch = [&](int q) { lq[q] = *reinterpret_cast<const i32x4 *>(perm_lds + (K0 / 4) * g + 4 * q); };
    // which chunks (operand registers) are relative to the window's current frame: the same for every lane - a bit mask in an SGPR
    unsigned cur_mask = 0;
#pragma unroll
    for (int i = 0; i < K0 / 4; ++i) cur_mask |= (lut_lds[K0 + i] != 0 ? 1u : 0u) << i;
    cur_mask = __builtin_amdgcn_readfirstlane(cur_mask);
    auto gather_one = [&](int tap, int i) {
        const unsigned base = (cur_mask >> i) & 1u ? b_cur : b_first + (unsigned)(tap * 3 * P.enc_jf) * 4u;
#if R3D_CHAIN_ABL & 4
        xv[i] = __builtin_bit_cast(float, base);
#else
        xv[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrsrc, base + (unsigned)(lq[i >> 2][i & 3] & ~3), 0, 0));
#endif
    };
    static_assert(K0 == 64, "the gather below walks 16 columns per lane in four table quads");
    auto tap_of = [&](int ts) { return ts == 0 ? 0 : ts == 2 ? res_tap : 3 - res_tap; };   // the residual tap comes last

    // One slab of 64 MFMAs in 4 half groups of 16 (a 16-byte fragment read per lane = the A operands of 4 MFMAs): MFMA n of half
    // group hg takes element e = n >> 2 of fragment f = n & 3.
    //   WIDE   (NARROW = 0): 4 K steps x 16 channel blocks: accumulator 4 hg + f, K step e of the slab's 4;
    //   NARROW (NARROW = 1): 16 K steps x 4 channel blocks: accumulator f, K step 4 hg + e of the slab's 16.
    // ZERO: the slab's first K step starts its accumulators (C = 0).
    long long bar_ticks = 0;             // (timing builds: time this wavefront spent in the slab barriers of the current tile)
    (void)bar_ticks;
    int st_cur = 0;                      // ring stage of the slab being multiplied (run-time: 76 slabs per tile, six stages)
    auto run_slab = [&](auto narrow_tag, auto zero_tag, auto bsrc, auto &acc, auto side) {
        constexpr bool NARROW = decltype(narrow_tag)::value != 0, ZERO = decltype(zero_tag)::value != 0;
        const int st_next = st_cur + 1 == NSTAGE ? 0 : st_cur + 1;
        const float *cur = smem + st_cur * SLAB_FLOATS + lane * 4, *nxt = smem + st_next * SLAB_FLOATS + lane * 4;
        st_cur = st_next;
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
                if (hg < 3 && n < 4) wq[(hg + 1) & 1][n] = *reinterpret_cast<const f32x4 *>(cur + (hg + 1) * 1024 + n * 256);
                if (hg == 3) {
                    // (the next slab landed a whole slab ago: its first fragments are read early in the last half group, away from the barrier)
                    if (n < 4) wq[0][n] = *reinterpret_cast<const f32x4 *>(nxt + n * 256);
                    if (n == 10) {
#if defined(R3D_TIMING) && defined(R3D_TIMING_BARRIERS)   // (two clock reads per slab in every wavefront: ~7 us per tile of their own - off unless asked for)
                        const long long tb = wall_clock64();
                        wg_barrier();
                        bar_ticks += wall_clock64() - tb;
#else
                        wg_barrier();
#endif
                    }
                }
                side(hg * 16 + n);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    // a wide layer of NS slabs: K step 4 M + st of slab M
    auto run_layer = [&](auto ns_tag, auto zero_tag, auto bsrc, f32x4 (&acc)[NCB], auto side) {
        constexpr int NS = decltype(ns_tag)::value;
        auto rec = [&](auto self, auto m_tag) {
            constexpr int MS = decltype(m_tag)::value;
            if constexpr (MS < NS) {
                run_slab(ChainIC<0>{}, ChainIC<(decltype(zero_tag)::value && MS == 0) ? 1 : 0>{},
                         [&](int st) { return bsrc(MS * 4 + st); }, acc, [&](int p) { side(MS, p); });
                self(self, ChainIC<MS + 1>{});
            }
        };
        rec(rec, ChainIC<0>{});
    };
    auto d_elem = [&](const f32x4 (&D)[NCB], int step) { return D[step >> 2][step & 3]; };
    f32x4 bqr[3];                                              // bias quads in flight
    f32x4 &bq = bqr[0], &bq2 = bqr[1];
#ifdef R3D_TIMING
    const long long t_entry = wall_clock64();
#endif

    // the first tile's first operand values go out before the opening barrier: both wait for memory once
    int row0 = __builtin_amdgcn_readfirstlane(tile_list[0].y);
    row_bases(row0);
#pragma unroll
    for (int q = 0; q < K0 / 16; ++q) lut_fetch(q);
#pragma unroll
    for (int i = 0; i < K0 / 4; ++i) gather_one(0, i);
    wg_barrier();                                               // opening barrier: slabs 0, 1 are in the ring
#pragma unroll
    for (int f = 0; f < 4; ++f) wq[0][f] = *reinterpret_cast<const f32x4 *>(smem + lane * 4 + f * 256);

#pragma unroll 1
    for (int ti = 0; ti < ntiles; ++ti) {
        const int next_row0 = ti + 1 < ntiles ? __builtin_amdgcn_readfirstlane(tile_list[(ti + 1) * tstride].y) : row0;
        const int lrow = wave * 16 + j;                        // this lane's row of the tile
        const __amdgpu_buffer_rsrc_t crs = act_rsrc(P.c + (size_t)row0 * P.ldc);
        const int o_voff = (lrow * P.ldc + 4 * g) * 4;         // channels 16 cb + 4 g .. + 3 of the lane's row at byte 64 cb behind it
        const bool row_ok = row0 + lrow < M;
#ifdef R3D_TIMING
        // phase stamps of the run's first tiles (workgroups 0-15): [0] tile start (the run's entry for its first tile) [1] first expand_conv
        // done [2] taps done [3] 1x1 done [4] tile end; [5] - [0] = time in the slab barriers, [6] - [5] = opening barrier -> first MFMA
        long long *dbg = dbg_base && ti < 8 ? dbg_base + ti * 8 : nullptr;
        if (dbg && tid == 0) { dbg[0] = ti == 0 ? t_entry : wall_clock64(); if (ti == 0) dbg[6] = wall_clock64() - t_entry; }
        bar_ticks = 0;
#endif
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) DB[cb] = f32x4{0, 0, 0, 0};
#pragma unroll 1
        for (int ts = 0; ts < 3; ++ts) {
            run_layer(ChainIC<SL_EXP>{}, ChainIC<1>{}, [&](int st) { return xv[st]; }, DA, [&](int MS, int p) {
                if (MS == SL_EXP - 1 && p == 40) bq = bias_quad(rb0, 0);
                if (MS == SL_EXP - 1 && p == 41) bq2 = bias_quad(rb0, 1);
            });
#ifdef R3D_TIMING
            if (dbg && tid == 0 && ts == 0) dbg[1] = wall_clock64();
#endif
            // activations (+ bias): block 0 now, block cb + 1 behind the MFMAs of the slab that reads block cb
#pragma unroll
            for (int r = 0; r < 4; ++r) DA[0][r] = lrelu(DA[0][r] + bq[r], slope0);
            // the next expand_conv's operand rows: the next tap's, or - behind the last tap - the next tile's first tap (the run's last
            // tile requests its own again: never a branch around a load, the wait for it would land in the MFMA stream)
            const int next_tap = ts == 2 ? 0 : tap_of(ts + 1);
            if (ts == 2) row_bases(next_row0);
            run_layer(ChainIC<SL_C>{}, ChainIC<0>{}, [&](int st) { return d_elem(DA, st); }, DB, [&](int MS, int p) {
                if (MS + 1 < NCB) {
                    if (p == 4) { bq = bq2; if (MS + 2 < NCB) bq2 = bias_quad(rb0, MS + 2); }
                    if (p == 8) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) DA[MS + 1][r] = lrelu(DA[MS + 1][r] + bq[r], slope0);
                    }
                }
                if (MS == 8) {
                    if (p >= 24 && p < 24 + K0 / 16) lut_fetch(p - 24);
                    if (p >= 32 && p < 32 + K0 / 4) gather_one(next_tap, p - 32);
                }
                if (MS == SL_C - 1 && p == 40) bq = bias_quad(rb1, 0);
                if (MS == SL_C - 1 && p == 41) bq2 = bias_quad(rb1, 1);
                if (R3D_CHAIN_V >= 2 && MS == SL_C - 1 && p == 42) bqr[2] = bias_quad(rb1, 2);
            });
        }
#ifdef R3D_TIMING
        if (dbg && tid == 0) dbg[2] = wall_clock64();
#endif
        // ---- the 1x1 convolution, four output blocks at a time over all K steps; B operand = lrelu(DB + b1), applied block by block
#pragma unroll
        for (int r = 0; r < 4; ++r) DB[0][r] = lrelu(DB[0][r] + bq[r], slope1);
        // epilogue of block group G (from O[G & 1]): out = lrelu(O + b2) + residual, one block per filler slot
        f32x4 ob[4];                                            // the bias quads of the group whose epilogue comes next
        auto out_block = [&](int G, int f) {
            const int cb = 4 * G + f;
            f32x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = lrelu(O[G & 1][f][r] + ob[f][r], slope2) + DA[cb][r];
            // (write-through: the consumer is another workgroup, mostly on another XCD)
#if R3D_CHAIN_ABL & 2
            asm volatile("" :: "v"(v));
#else
            if (row_ok) act_store4(crs, o_voff + cb * 64, v);
#endif
        };
        auto group_rec = [&](auto self, auto g_tag) {
            constexpr int G = decltype(g_tag)::value;
            if constexpr (G < NCB / 4) {
                auto slab_rec = [&](auto self2, auto m_tag) {
                    constexpr int MS = decltype(m_tag)::value;
                    if constexpr (MS < 4) {
                        run_slab(ChainIC<1>{}, ChainIC<MS == 0 ? 1 : 0>{}, [&](int st) { return d_elem(DB, MS * 16 + st); }, O[G & 1], [&](int p) {
                            // the first group activates the B operand as it goes: half group hg of slab MS reads block 4 MS + hg; the next block behind it
                            if (G == 0) {
                                const int hg = p >> 4, nb = 4 * MS + hg + 1;
                                if (nb < NCB) {
#if R3D_CHAIN_V >= 2
                                    // block nb's quad was requested two half groups ago (slot nb % 3: 36 MFMAs = 0.5 us; one half group -
                                    // 0.2 us, an L2 round trip on an idle chip - stalled the matrix pipe inside the forward)
                                    if ((p & 15) == 2 && nb + 2 < NCB) bqr[(nb + 2) % 3] = bias_quad(rb1, nb + 2);
                                    if ((p & 15) == 6) {
#pragma unroll
                                        for (int r = 0; r < 4; ++r) DB[nb][r] = lrelu(DB[nb][r] + bqr[nb % 3][r], slope1);
                                    }
#else
                                    if ((p & 15) == 2) { bq = bq2; if (nb + 1 < NCB) bq2 = bias_quad(rb1, nb + 1); }
                                    if ((p & 15) == 6) {
#pragma unroll
                                        for (int r = 0; r < 4; ++r) DB[nb][r] = lrelu(DB[nb][r] + bq[r], slope1);
                                    }
#endif
                                }
                            }
                            // the previous group's epilogue (its bias quads were requested in that group's last slab).  On gfx9 loads and
                            // stores share vmcnt and may complete out of order with each other: a wait for a LOAD with a store pending is
                            // compiled as a wait for the store too - a write-through store's trip to memory inside the MFMA stream, twice per
                            // group.  So: every load this wavefront has in flight is waited for BEFORE the group's first store.
                            if (R3D_CHAIN_V >= 1 && G > 0 && MS == 0 && p == 8) CHAIN_WAIT_VM0();
                            if (G > 0 && MS == 0 && p >= 8 && p < 40 && (p & 7) == 2) out_block(G - 1, (p - 8) >> 3);
                            if (R3D_CHAIN_V >= 1) { if (MS == 3 && p >= 16 && p < 20) ob[p - 16] = bias_quad(rb2, 4 * G + (p - 16)); }
                            else if (MS == 3 && p >= 48 && p < 52) ob[p - 48] = bias_quad(rb2, 4 * G + (p - 48));
                        });
                        self2(self2, ChainIC<MS + 1>{});
                    }
                };
                slab_rec(slab_rec, ChainIC<0>{});
                self(self, ChainIC<G + 1>{});
            }
        };
        group_rec(group_rec, ChainIC<0>{});
#ifdef R3D_TIMING
        if (dbg && tid == 0) dbg[3] = wall_clock64();
#endif
        if (R3D_CHAIN_V >= 1) CHAIN_WAIT_VM0();               // (as above: the last group's quads, before its stores)
#pragma unroll
        for (int f = 0; f < 4; ++f) out_block(NCB / 4 - 1, f);
        // ---- the tile is finished when its write-through stores have left the CU: drain, everyone, one add per 32-row unit
        if (cnt) tile_drain();
        wg_barrier();
        if (cnt) {
            const int4 te = tile_list[ti * tstride + 1];     // {dependencies (none), first ready counter, granules, -}
            tile_signal(cnt, __builtin_amdgcn_readfirstlane(te.y), __builtin_amdgcn_readfirstlane(te.z), 2);
        }
#ifdef R3D_TIMING
        if (dbg && tid == 0) { dbg[4] = wall_clock64(); dbg[5] = dbg[0] + bar_ticks; dbg[7] = dbg[5]; if (ti == 0) dbg[6] += dbg[5]; else dbg[6] = dbg[5]; }
#endif
        row0 = next_row0;
    }
}
