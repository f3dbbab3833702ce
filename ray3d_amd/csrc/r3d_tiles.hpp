// Device code of the gfx950 kernels: every tile kind and the persistent tile loop (gemm_persistent).  Included by the kernel
// translation units (r3d_k_*.hip: one per kernel family, so that `make -j` compiles them side by side - the kernels are
// 4 minutes of hipcc between them) and by r3d_kernels.hip (bind / decode kernels, launchers).  Header-only: everything
// here is a template or __forceinline__.
#pragma once
// gfx950 (MI355X, CDNA4) kernels of the lifting forward pass.  Written for 64-lane wavefronts and
// the fp32-input matrix cores (v_mfma_f32_32x32x2_f32: exact fp32, 64 FLOP/clk/SIMD, 157 TFLOP/s
// chip peak) - the path is compute-bound, and the 1e-4 parity budget rules out
// plain bf16.  Opt-in (r3d_config.bf16x3): the same fp32 results on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16)
// through exact three-term splits of both operands - the *_b3 tile kinds below.
//
//  r3d_gemm_f32        persistent grouped GEMM, one launch per level of the plan's DAG.  Tile kinds:
//                        gemm_tile         C = res + lrelu(A W^T + b): every Conv1d / Linear of TemporalBlock /
//                                          FCBlock / Embedding (rie.py:85-105, :122-135, :159-169) with eval
//                                          BatchNorm folded; split-K variants for the small launches;
//                        gemm_tile<PAIR>   a pyramid level's 3-tap and 1x1 convolutions (rie.py:94-97), the
//                                          intermediate tile staying in LDS;
//                        first_level_taps  expand_conv on the gathered input (window gather,
//                                          lib/train_val/trainer.py:47-58; body-part grouping and the
//                                          positional / temporal differences of rie.py:290-357 folded into the
//                                          weights) + the first pyramid level, tap by tap, for 32 / 64 output rows;
//                        enc_tile          GlobalInfo's input (the windows' current frames) gathered the same way;
//                        gemm_tile_b3 (gemm_tile_b3p for single-unit tiles), gemm_tile_b3t, first_level_taps_b3
//                                          the 1024-wide Linears, the fused pairs and the first level on the bf16 matrix cores.
//  r3d_gemm_enc_f32    expand_conv / GlobalInfo.fc_1 with the gather fused, where first_level_taps is not used: one-level
//                      architectures, more than 256 channels, the dense ablation - and the un-fused plan of calls of
//                      <= 48 windows (r3d_plan.cpp, plan_kind).
//  r3d_gemm_uv_f32, r3d_gemm_enc_uv_f32
//                      the same two kernels for launches that gather pixel keypoints (UV input mode).
//  r3d_decode_f32      last Linear of the decoders + joint reassembly (rie.py:409-432) + trajectory
//                      add (lib/train_val/trainer.py:353).
//  UV input mode (pixel keypoints + per-window camera rows) has no kernel of its own: the gathers of first_level_taps
//  and enc_tile encode each value they stage - ray = ((u-cx)/fx, c*y+s, -s*y+c), float64 like the reference's NumPy
//  (lib/camera/camera.py:423-471) - with the camera of the window the operand row belongs to.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <type_traits>

#include "r3d_internal.hpp"

namespace r3d {

// development instrumentation (tools/gemm_probe -DR3D_TIMING): wall-clock stamps of a tile's phases
#ifndef R3D_TS
#define R3D_TS 0          // which of a first-level tile's three tap phases gets the fine stamps (timing builds)
#endif
#ifdef R3D_TIMING
#define R3D_TSTAMP(slot) do { if (dbg && threadIdx.x == 0) dbg[slot] = wall_clock64(); } while (0)
#else
#define R3D_TSTAMP(slot) do { } while (0)
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Explicit global-address-space accesses.  Pointers that reach a kernel through a descriptor table
// are "generic" to the compiler, which then emits flat_load/flat_store: those tick BOTH vmcnt and
// lgkmcnt, so every `s_waitcnt lgkmcnt(0)` in front of an MFMA (meant for ds_read) would also wait
// for the HBM loads in flight.  Casting to address space 1 gives global_load/global_store.
#define R3D_AS1 __attribute__((address_space(1)))
__device__ __forceinline__ f32x4 gload4(const float *p) { return *(const R3D_AS1 f32x4 *)p; }
__device__ __forceinline__ float gload1(const float *p) { return *(const R3D_AS1 float *)p; }
__device__ __forceinline__ void gstore1(float *p, float v) { *(R3D_AS1 float *)p = v; }
// Activations are handed from tile to tile INSIDE a launch (r3d_forward_f32: the whole forward is one launch, tiles
// ordered by ready counters), possibly across XCDs whose L2s are not coherent with each other and always across CUs
// whose L1s are never refreshed: every activation store is write-through (sc1) and every activation load bypasses the
// L1 (sc1) - MI355X_MICROARCH.md, inter-workgroup visibility: "sc1 payload, every storing wave drains, one flag".
// Weights, biases, tables and the raw input are read-only for the whole launch: plain loads.
constexpr int ACT_AUX = 16;                  // aux bits of the buffer builtins: sc1
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned R3D_AS1 *gu32;
// Single-launch form: a tile's wait for its producers, carried INTO the tile: the tile requests its first weight
// fragments - which depend on no producer - and only then asks for the ready counters (r3d_tiles.hpp: wait_deps_reg)
struct LateWait { int dw, ndep; gu32 cnt, abort_flag; long long spin_ticks; };
__device__ __forceinline__ void wait_deps_reg(const int dw, const int ndep, const gu32 cnt, const gu32 abort_flag, const long long spin_ticks);
__device__ __forceinline__ __amdgpu_buffer_rsrc_t act_rsrc(const float *base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ f32x4 act_load4(__amdgpu_buffer_rsrc_t r, int byte_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, ACT_AUX));
}
__device__ __forceinline__ float act_load1(__amdgpu_buffer_rsrc_t r, int byte_off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, ACT_AUX));
}
__device__ __forceinline__ void act_store4(__amdgpu_buffer_rsrc_t r, int byte_off, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, byte_off, 0, ACT_AUX);
}
__device__ __forceinline__ void act_store1(__amdgpu_buffer_rsrc_t r, int byte_off, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, byte_off, 0, ACT_AUX);
}
// A tile is finished when its write-through stores have left the CU: every storing wavefront drains, a barrier, then
// one relaxed agent-scope add per 32-row unit on the unit's ready counter (granules of 32 columns).  The callers'
// barrier is the one that ends the tile anyway.
__device__ __forceinline__ void tile_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void tile_signal(const gu32 cnt, const int sig_base, const int sig_add, const int units) {
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));      // (opaque: `cnt + 4 t` would be hoisted out of the persistent loop and stay live - and spill - across every tile)
    if (t < units) __hip_atomic_fetch_add(cnt + sig_base + t, (unsigned)sig_add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// LeakyReLU for slopes in (0, 1] (0.2, 0.01; 1 = linear layer): max(v, slope v) - a multiply and a max instead of
// multiply, compare, select
__device__ __forceinline__ float lrelu(const float v, const float slope) { return __builtin_fmaxf(v, v * slope); }

// ------------------------------------------------------------------------------------ GEMM
//
// One persistent launch per DAG level: grid = #CUs, one 512-thread workgroup (8 wavefronts, two per
// SIMD) per CU.  The host cuts the level's work - all (problem, 256-column block, 32-row unit)
// triples - into one contiguous, cost-balanced chunk per workgroup (r3d_schedule.cpp); a chunk is
// executed as a few tiles of BM = 32*MI rows (MI = 1..6) by 256 columns.  Wavefront w owns columns
// [32w, 32w+32) of the tile and all MI row blocks, so any MI is perfectly balanced across the 8
// wavefronts and the only waste is the 32-row MFMA granularity.
//
// Operand paths (BK = 32 per K tile):
//  * A (activations, BM x 32) is shared by all 8 wavefronts: global -> VGPR -> LDS ring of three
//    stages.  Tile t+4 is loaded from HBM while tile t feeds the matrix cores, tile t+2 is written
//    to LDS at the top of iteration t (two iterations after its loads were issued), so neither the HBM latency nor the LDS write sits between a
//    barrier and the next MFMA.  Rows are padded to 36 floats: the 16 lanes a ds_read_b128 services
//    together hit 16 distinct 16-byte slots (SQ_LDS_BANK_CONFLICT = 0).
//  * W (weights) never touches LDS: the host packs every layer in MFMA fragment order
//    (r3d_model.cpp) so that a wavefront's B fragments of one K tile are four fully coalesced
//    1 KiB loads straight into VGPRs, issued one K tile ahead.  Each wavefront reads only its own
//    32 columns - there is nothing to share.
//  * With MI <= 4 the first A fragments of tile t+1 are read before the end-of-tile barrier, so the
//    first MFMA after the barrier issues immediately.
// MFMA operand mapping (v_mfma_f32_32x32x2_f32): lane l supplies A[i = l&31][k = l>>5] and
// B[k = l>>5][j = l&31].  Lane (i, h) holds 4 consecutive floats k = 16h + 4q .. +3 of its row /
// column per fragment register quad and feeds them to 4 MFMAs; A and W use the same k permutation
// so the sum over k is unchanged.

constexpr int LDS_LD = BK + 4;                       // 36 floats = 144 B per staged row
constexpr int GEMM_THREADS = 512;
constexpr int GEMM_BN = 256;
constexpr int GEMM_MAX_MI = 6;
constexpr int GEMM_STAGES = 3;
constexpr int STAGE_FLOATS = GEMM_MAX_MI * 32 * LDS_LD;                  // one A tile: 27,648 B
constexpr int LUT_LDS_INTS = 1152;                                       // fused-prologue tables: K + K/4 ints (K <= 480 with room to spare)
constexpr int RING_LDS_BYTES = GEMM_STAGES * STAGE_FLOATS * 4;                      // 82,944 B
constexpr int GEMM_LDS_BYTES = 157952;   // the bf16x3 first level: three H planes + two gather buffers of three planes + tables (fp32 first level 134,400; fused pairs 133,120)
static_assert(GEMM_LDS_BYTES >= RING_LDS_BYTES, "the ring and the intermediate tile share the allocation");

typedef const LaunchArgs __attribute__((address_space(4))) *LaunchArgsPtr;
typedef const GemmProb __attribute__((address_space(4))) &ProbRef;

// Epilogue of a tile of COLS = 256 / KS columns: C = res + lrelu(acc + bias), written in wide rows.
// The MFMA leaves each (phase-0) wavefront with a 32-column slab (C/D layout of v_mfma_f32_32x32x2_f32:
// col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)); storing that directly means 4-byte
// accesses in 128-byte pieces, and the short-K layers become store-issue bound.  Instead the slabs of 32
// rows are transposed through LDS (idle after the K loop) so that every lane of all eight wavefronts
// moves 16 bytes and a wavefront instruction covers whole output rows (1 KiB at COLS = 256) - for
// the residual read too.
constexpr int EPI_LD = GEMM_BN + 4;       // 260 floats: the two 32-lane halves of a ds_write_b32 hit different banks

template <int MI, int KS>
__device__ __forceinline__ void store_tile(ProbRef P, const f32x16 (&acc)[MI], const int row0, const int col0, float *lds,
                                           const bool second = false) {
    constexpr int COLS = GEMM_BN / KS, WN = 8 / KS;
    constexpr int TPR = COLS / 4;                 // threads per output row (16 bytes each)
    constexpr int RPP = GEMM_THREADS / TPR;       // rows per pass: 8 / 16 / 32
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));                 // (opaque: keeps this function's per-thread constants out of the persistent loop's preheader)
    const int lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int M = P.M, N = P.N;
    const float slope = second ? P.slope2 : P.slope;
    const float *res = P.res;
    const int ldc = P.ldc, ldr = P.ldr;
    // (descriptors based at the tile's first element: per-lane offsets stay small and the accesses carry sc1)
    const __amdgpu_buffer_rsrc_t crs = act_rsrc(P.c + (size_t)row0 * ldc + col0);
    const __amdgpu_buffer_rsrc_t rrs = act_rsrc(res ? res + (size_t)row0 * ldr + col0 : P.c);
    const bool writer = wave < WN;                // the wavefronts of K phase 0 hold the sums
    const float bias = writer ? gload1((second ? P.bias2 : P.bias) + col0 + wave * 32 + li) : 0.0f;
    float *wr = lds + (4 * lh) * EPI_LD + wave * 32 + li;
    const int rd_row = tid / TPR, rd_c4 = (tid % TPR) * 4;
    const bool vec = (col0 + COLS <= N);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        __syncthreads();                                            // LDS free: K loop / reduction / previous slab done
        if (writer) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[mi][r] + bias;
                wr[((r & 3) + 8 * (r >> 2)) * EPI_LD] = lrelu(v, slope);
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 32 / RPP; ++j) {
            const int lr = rd_row + RPP * j;
            const int row = row0 + mi * 32 + lr;
            if (row >= M) continue;
            f32x4 v = *reinterpret_cast<const f32x4 *>(lds + lr * EPI_LD + rd_c4);
            const int col = col0 + rd_c4;
            const int lrow = mi * 32 + lr;
            if (vec) {
                if (res) v += act_load4(rrs, (lrow * ldr + rd_c4) * 4);
                // (write-through: the consumer is another workgroup, mostly on another XCD - no use for the line in this L2)
                act_store4(crs, (lrow * ldc + rd_c4) * 4, v);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (col + e < N) act_store1(crs, (lrow * ldc + rd_c4 + e) * 4, v[e] + (res ? act_load1(rrs, (lrow * ldr + rd_c4 + e) * 4) : 0.0f));
            }
        }
    }
    __syncthreads();                                                // the next tile's staging may overwrite the slab
}

// KS = 2 / 4 ("split-K inside the workgroup") serves problems with too few row units to occupy the
// chip or to balance a launch: the tile is 256/KS columns wide, wavefront w multiplies column block
// w % (8/KS) with every KS-th 32-wide K tile (phase w / (8/KS)), and the KS partial sums are added
// through LDS at the end.  KS times as many tiles, each with 1/KS of the K-loop iterations.
//
// PAIR: the problem is a fused pair (GemmProb::w2): the tile of the first layer, all N <= 256 columns of it, stays
// in LDS in the layout an MFMA loop reads and the second layer (K2 = N) runs on it at once - no barriers, no
// staging, weights streaming - before the one epilogue with the residual.  This is a level of the conv pyramid
// (lib/model/rie.py:94-97): the 3-tap stride-3 convolution and the 1x1 convolution that follows it.  Unfused, the
// 1x1 layer is a K = 256 launch whose prologue, residual epilogue and launch cost rival its 8 K tiles of MFMAs,
// and its input makes a round trip through HBM.
constexpr int PAIR_LD = GEMM_BN + 4;                                     // 260 floats per row of the intermediate tile
constexpr int PAIR_MAX_MI = 4;                                           // 128 x 260 floats = 133,120 B of LDS
template <int MI, int KS, bool PAIR = false>
__device__ __forceinline__ void gemm_tile(ProbRef P, const int row0, const int col0, float *smem, long long *dbg, const LateWait late = LateWait{0, 0, nullptr, nullptr, 0}) {
    static_assert(!PAIR || (KS == 1 && MI <= PAIR_MAX_MI), "fused pairs are whole tiles of at most 128 rows");
    constexpr int SF = STAGE_FLOATS;        // floats per LDS ring stage
    R3D_TSTAMP(0);
    static_assert(KS == 1 || (KS == 2 && MI <= 2) || (KS == 4 && MI == 1), "split-K tiles are small tiles");
    constexpr int WN = 8 / KS;              // 32-column blocks per tile
    constexpr int VR = KS * MI * 32;        // staged rows per iteration (KS sub-tiles of MI*32 rows x 32 k)
    constexpr int NA = (VR + 63) / 64;      // A staging slots per thread (64 staged rows per slot)
    constexpr bool PRE = MI <= 3;           // pre-read next tile's first A fragments before the barrier
    // Weight prefetch distance.  A K tile of a small tile is short (MI = 1: ~2k cycles for the two wavefronts
    // of a SIMD) - shorter than an L2 miss - so W runs two tiles ahead there, with three register sets
    // rotating (and the A staging registers likewise, loop unrolled by three).  MI >= 5 keeps distance one:
    // its K tile is long enough and the registers are needed for accumulators.
    constexpr bool WD2 = MI <= 4;
    constexpr int AD = WD2 ? 5 : 4;         // A tile t+AD is issued in iteration t, committed to LDS in iteration t+AD-2
    int tid = threadIdx.x;
    // (opaque to the optimiser: otherwise the per-thread constants of every tile shape are hoisted out of the
    // persistent loop and stay live across the 6-unit tiles, which have no register to spare)
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int wn = wave % WN;                       // 32-column block of the tile this wavefront owns
    const int wk = wave / WN;                       // which 32-wide K tile of an iteration it multiplies
    const int M = P.M, K = P.K;
    const int nk32 = K / BK;                        // 32-wide K tiles
    const int nk = (nk32 + KS - 1) / KS;            // K-loop iterations
    const int srow = tid >> 3, a_kq = (tid & 7) * 4;

    // ---- A staging state
    // A may be a virtual concatenation of up to MAX_SEG buffers (the torch.cat of rie.py:371-407 is never
    // materialised); a plain operand is the one-segment case.  K tiles are issued in increasing order,
    // so the segment state (buffer, leading dimension, first/last K) only ever advances and the
    // descriptor table is touched at segment boundaries only (<= 3 times per tile).  No scalar loads in
    // the steady-state loop: an SMEM load in flight would also degrade every counted
    // `s_waitcnt lgkmcnt(N)` in front of the MFMAs to (0).  (Split-K tiles: the scheduler only splits a
    // concatenated operand when every interior boundary is a multiple of 32*KS, so the KS sub-tiles of
    // an iteration always lie in one buffer.)
    // Operands are read through a buffer descriptor whose base (first row of the tile, first column
    // of the segment) and K-tile offset are scalars: a staging load is ONE instruction with no vector
    // address arithmetic.  That matters because a wavefront's VALU instructions crawl (about one per
    // 64 cycles) while its SIMD partner streams MFMAs, whereas memory instructions issue freely.  The
    // descriptor is bounded at the segment's last valid element: the K tiles a short last split-K
    // iteration has no use for read zeros instead of memory past the buffer.
    const bool multi = P.kend[0] < K;
    int seg_ld = P.lda[0], seg_k0 = 0, seg_end = P.kend[0], seg_i = 0;
    int a_voff[NA];                         // byte offset of the slot's 16 bytes inside the segment's tile rows
    __amdgpu_buffer_rsrc_t arsrc;
    auto open_seg = [&]() {
        const float *base = P.a[seg_i] + (size_t)row0 * seg_ld;
        const long long b = ((long long)(M - 1 - row0) * seg_ld + (seg_end < K ? seg_end : K) - seg_k0) * 4;
        arsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, b < 0x7fffffffLL ? (int)b : 0x7fffffff, 0x00020000);
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            // staged row -> (K sub-tile, row of the tile); staged rows past the tile (odd MI) re-read valid data
            const int vr = srow + 64 * i, sub = KS == 1 ? 0 : (vr / (MI * 32) < KS ? vr / (MI * 32) : KS - 1);
            const int gr = row0 + vr - sub * (MI * 32);
            a_voff[i] = (((gr < M ? gr : M - 1) - row0) * seg_ld + a_kq + sub * BK) * 4;
        }
    };
    open_seg();
    auto prep_seg = [&](int kt) {
        if (!multi) return;
        while (kt * (BK * KS) >= seg_end) {     // uniform
            ++seg_i;
            seg_k0 = seg_end;
            seg_ld = P.lda[seg_i];
            seg_end = P.kend[seg_i];
            open_seg();
        }
    };
    struct Staged {                         // one A tile on its way from HBM to LDS
        f32x4 a[NA];
    };
    Staged ra, ra2, ra3;                    // tiles in flight (ra3: three-set rotation only)
    auto issue_a = [&](int kt, Staged &R) {
        // every slot loads unconditionally (rows past the tile are clamped and never consumed): a
        // predicated load would make the compiler wait for ALL outstanding loads at the merge point.
        // The K segment of this tile was looked up one iteration ago (prep_seg), so that scalar-load
        // round trip is off the critical path of short K tiles.
        const int kb = kt * (BK * KS) - seg_k0;
#pragma unroll
        for (int i = 0; i < NA; ++i)
            R.a[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(arsrc, a_voff[i], kb * 4, ACT_AUX));
    };
    const int st_off = srow * LDS_LD + a_kq;
    auto commit_a = [&](int stage, const Staged &R) {
        // (stage is uniform: three copies of the stores with immediate offsets, no address arithmetic)
        float *s = stage == 0 ? smem + st_off : stage == 1 ? smem + SF + st_off : smem + 2 * SF + st_off;
        {
#pragma unroll
            for (int i = 0; i < NA; ++i) *reinterpret_cast<f32x4 *>(s + i * 64 * LDS_LD) = R.a[i];
        }
    };

    // ---- W fragments: [(n/32)][k tile][q][lane][4] in HBM, this wavefront's 32 columns
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);   // make the descriptor provably wave-uniform
    // (the wavefront's K-slice of an iteration is part of the base, not of the scalar offset: the compiler
    // kept wave_u / WN in a VGPR and wrapped every W load in a waterfall loop)
    __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(P.w + ((size_t)((col0 >> 5) + wave_u % WN) * nk32 + wave_u / WN) * 1024), 0,
        (nk32 - wave_u / WN) * 4096, 0x00020000);   // (K tiles past the end read as zeros through the descriptor's bound)
    const int w_voff = lane * 16;
    f32x4 rb[4], rbn[4], rbn2[4];           // W fragments of the current and the next K tile(s)
    auto load_w = [&](int kt, f32x4 (&dst)[4]) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            dst[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                wrsrc, w_voff + q * 1024, kt * KS * 4096, 0));
    };

    f32x16 acc[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][r] = 0.0f;

    const int a_frag = (wk * MI * 32 + li) * LDS_LD + lh * 16;
    f32x4 av0[PRE ? MI : 1];

    // ---- prologue: tiles 0 and 1 into LDS, tile 2 in flight, W(0) in registers
    // (tile indices past the end are clamped instead of predicated: a redundant reload of the last
    // tile is free, while a branch around a load makes the compiler's s_waitcnt placement pessimistic)
    const int last = nk - 1;
    load_w(0, rb);
    if (WD2) load_w(1 < last ? 1 : last, rbn);
    if (late.ndep > 0) wait_deps_reg(late.dw, late.ndep, late.cnt, late.abort_flag, late.spin_ticks);   // (uniform; the activations below are the producers' outputs)
    {
        Staged r0, r1;                      // all prologue tiles in flight at once: one HBM latency, not four
        issue_a(0, r0);
        prep_seg(1 < last ? 1 : last);
        issue_a(1 < last ? 1 : last, r1);
        prep_seg(2 < last ? 2 : last);
        issue_a(2 < last ? 2 : last, ra);
        prep_seg(3 < last ? 3 : last);
        issue_a(3 < last ? 3 : last, ra2);
        prep_seg(4 < last ? 4 : last);
        if (WD2) {
            issue_a(4 < last ? 4 : last, ra3);
            prep_seg(5 < last ? 5 : last);
        }
        commit_a(0, r0);
        commit_a(1, r1);
    }
    __syncthreads();
    if (PRE) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) av0[mi] = *reinterpret_cast<const f32x4 *>(smem + a_frag + mi * 32 * LDS_LD);
    }
    R3D_TSTAMP(1);
    int st_cur = 0;                              // kt % 3 without a division
    // One K tile.  `w_use` holds this tile's W fragments, `w_load` receives the next tile's: the two
    // register sets swap roles every tile (loop unrolled by two) instead of being copied - a copy
    // would force `s_waitcnt vmcnt(0)` at the end of EVERY tile and with it the HBM latency of the
    // A loads issued in the same tile.  Staging comes first for every wavefront and is a handful of
    // memory instructions (no VALU): non-MFMA instructions of one wavefront issue at about one per
    // MFMA of its SIMD partner, so anything that is not an MFMA belongs in the gap after the barrier
    // (an asymmetric compute-first/stage-first split between the partners was measured slower).
    auto k_tile = [&](int kt, f32x4 (&w_use)[4], f32x4 (&w_load)[4], Staged &stg) {
        const int st_next = st_cur == 2 ? 0 : st_cur + 1, st_next2 = st_next == 2 ? 0 : st_next + 1;
        const float *s = smem + st_cur * SF + a_frag;
        const bool active = KS == 1 || kt * KS + wave_u / WN < nk32;   // K-tile count not a multiple of KS: short last iteration
        auto mfma_q = [&](int q) {
            if (!active) return;
            f32x4 av[MI];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                if (PRE && q == 0) av[mi] = av0[mi];
                else av[mi] = *reinterpret_cast<const f32x4 *>(s + mi * 32 * LDS_LD + q * 4);
            }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
                    acc[mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mi][kk], w_use[q][kk], acc[mi], 0, 0, 0);
        };
        // Where the staging instructions go.  MI >= 3: after the first MFMAs, in the shadow of this
        // wavefront's OWN matrix work (a 32x32x2 MFMA occupies the pipe for 64 cycles but issues in a few).
        // MI <= 2: in front - with one or two accumulators the MFMAs form a dependent chain and any
        // instruction slipped between them costs ~43 cycles (measured: +10 % on the M=256 MLP levels).
        constexpr bool INTERLEAVE = MI >= 3;
        if (!INTERLEAVE) {
            commit_a(st_next2, stg);                 // tile kt+2, issued two iterations ago (before newer loads: vmcnt order)
            load_w(kt + (WD2 ? 2 : 1) < last ? kt + (WD2 ? 2 : 1) : last, w_load);
            issue_a(kt + AD < last ? kt + AD : last, stg);
            prep_seg(kt + AD + 1 < last ? kt + AD + 1 : last);
        }
        mfma_q(0);
        if (INTERLEAVE) {
            commit_a(st_next2, stg);
            load_w(kt + (WD2 ? 2 : 1) < last ? kt + (WD2 ? 2 : 1) : last, w_load);
        }
        mfma_q(1);
        if (INTERLEAVE) {
            issue_a(kt + AD < last ? kt + AD : last, stg);
            prep_seg(kt + AD + 1 < last ? kt + AD + 1 : last);
        }
        mfma_q(2);
        mfma_q(3);
        if (PRE) {
            const float *sn = smem + st_next * SF + a_frag;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) av0[mi] = *reinterpret_cast<const f32x4 *>(sn + mi * 32 * LDS_LD);
        }
        __syncthreads();
        st_cur = st_next;
    };
    int kt = 0;
    if (WD2) {
        // tile t multiplies with W set t % 3 while tile t+2's fragments land in set (t+2) % 3
        for (; kt + 2 < nk; kt += 3) {
            k_tile(kt, rb, rbn2, ra);
            k_tile(kt + 1, rbn, rb, ra2);
            k_tile(kt + 2, rbn2, rbn, ra3);
        }
        if (kt < nk) {
            k_tile(kt, rb, rbn2, ra);
            if (kt + 1 < nk) k_tile(kt + 1, rbn, rb, ra2);
        }
    } else {
        for (; kt + 1 < nk; kt += 2) {
            k_tile(kt, rb, rbn, ra);
            k_tile(kt + 1, rbn, rb, ra2);
        }
        if (kt < nk) k_tile(kt, rb, rbn, ra);
    }

    // ---- epilogue
    R3D_TSTAMP(2);
    if (KS > 1) {
        // add the partial sums of the K phases 1..KS-1 to phase 0's through LDS (the ring is idle now)
        // (slot of phase 0 stays unused: non-negative offsets fold into the ds instructions' immediates)
        constexpr int PHASE_FLOATS = WN * MI * 16 * 64;
        float *red = smem + ((wn * MI) * 16) * 64 + lane;
        if (wk > 0) {
            float *mine = red + wk * PHASE_FLOATS;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) mine[(mi * 16 + r) * 64] = acc[mi][r];
        }
        __syncthreads();
        if (wk == 0) {
#pragma unroll
            for (int ph = 1; ph < KS; ++ph)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mi][r] += red[ph * PHASE_FLOATS + (mi * 16 + r) * 64];
        }
    }
    R3D_TSTAMP(3);
    if constexpr (PAIR) {
        // ---- first layer's activations -> LDS (the staging ring is dead), as the A operand of the second
        const float slope1 = P.slope;
        const float bias1 = gload1(P.bias + wave * 32 + li);
        __syncthreads();                                            // every wavefront is done with the ring
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            float *wr = smem + (mi * 32 + 4 * lh) * PAIR_LD + wave * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = acc[mi][r] + bias1;
                wr[((r & 3) + 8 * (r >> 2)) * PAIR_LD] = lrelu(v, slope1);
            }
        }
        __syncthreads();
        // ---- second layer: barrier-free MFMA loop over K2 = N, weight fragments two K tiles ahead
        const int nk2 = P.K2 / BK;
        __amdgpu_buffer_rsrc_t w2rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(P.w2 + ((size_t)wave_u * nk2) * 1024), 0, nk2 * 4096, 0x00020000);
        auto load_w2 = [&](int kt, f32x4 (&dst)[4]) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                dst[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w2rsrc, w_voff + q * 1024, kt * 4096, 0));
        };
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][r] = 0.0f;
        const float *h_frag = smem + li * PAIR_LD + lh * 16;
        const int last2 = nk2 - 1;
        auto k_tile2 = [&](int kt2, f32x4 (&w_use)[4], f32x4 (&w_load)[4]) {
            load_w2(kt2 + 2 < last2 ? kt2 + 2 : last2, w_load);
            // (pinned, as in first_level_taps: the scheduler would sink the request to its use)
            __builtin_amdgcn_sched_barrier(0);
            const float *s = h_frag + kt2 * BK;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 av[MI];
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) av[mi] = *reinterpret_cast<const f32x4 *>(s + mi * 32 * PAIR_LD + q * 4);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
                        acc[mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mi][kk], w_use[q][kk], acc[mi], 0, 0, 0);
            }
        };
        load_w2(0, rb);
        load_w2(1 < last2 ? 1 : last2, rbn);
        int kt2 = 0;
        for (; kt2 + 2 < nk2; kt2 += 3) {
            k_tile2(kt2, rb, rbn2);
            k_tile2(kt2 + 1, rbn, rb);
            k_tile2(kt2 + 2, rbn2, rbn);
        }
        if (kt2 < nk2) {
            k_tile2(kt2, rb, rbn2);
            if (kt2 + 1 < nk2) k_tile2(kt2 + 1, rbn, rb);
        }
        store_tile<MI, 1>(P, acc, row0, col0, smem, true);      // (begins with a barrier: the tile is dead)
    } else {
        store_tile<MI, KS>(P, acc, row0, col0, smem);
    }
    R3D_TSTAMP(4);
}

// ------------------------------------------------------------------------------------ single-unit tiles of NB x 32 columns
//
// gemm_tile_nb<NB>: C = res + lrelu(A W^T + b) for ONE 32-row unit and NB in {5, 6, 7} 32-column blocks - the tile of the
// M = B levels of calls whose levels are one tile deep (256 windows: a FuseBlock / Integration level is 160 - 224 whole
// 32 x 256 tiles for 256 CUs, a level cannot be shorter than its longest tile, and a quarter of the chip idles).  Cutting
// a row of 32 column blocks into five tiles of 7 / 6 instead of four of 8 puts every CU to work IF the narrower tile is
// proportionally shorter, which the 32 x 256 tile's wavefront = column block mapping cannot give (six blocks on four
// SIMDs take as long as eight).  Here wavefronts 0-3 (one per SIMD) own column blocks 0-3 whole, and the 4 (NB - 4)
// quarter-blocks that remain - column block b, k = 4 q .. 4 q + 3 of each 16-deep half of a K tile: one q of the fragment
// packing, 4 of a block's 16 MFMAs per K tile - are dealt NB - 4 apiece to wavefronts 4-7, the SIMD partners: every
// SIMD issues 16 + 4 (NB - 4) MFMAs per K tile instead of 32.  The partial sums of a split block are added through LDS
// before the epilogue (as the split-K tiles do).  Operand paths as in gemm_tile<1, 1>: A through the three-stage ring,
// weights fragment-ordered straight into VGPRs two K tiles ahead (a split block's wavefront requests its own q only).
template <int NB>
__device__ __forceinline__ void gemm_tile_nb(ProbRef P, const int row0, const int col0, float *smem, long long *dbg, const LateWait late = LateWait{0, 0, nullptr, nullptr, 0}) {
    static_assert(NB >= 4 && NB <= 7, "NB = 8 is gemm_tile<1, 1>");   // (NB = 4: the narrow end of an uneven row - wavefronts 4-7 only stage)
    constexpr int SF = STAGE_FLOATS;
    constexpr int NX = NB - 4;               // quarter-blocks per extra wavefront
    constexpr int AD = 5;
    R3D_TSTAMP(0);
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const bool main_w = wave_u < 4;
    const int M = P.M, K = P.K;
    const int nk = K / BK;
    const int srow = tid >> 3, a_kq = (tid & 7) * 4;
    // ---- this wavefront's share, as slots (column block of the tile, q): wavefront w < 4 the four q of block w; wavefront
    // 4 + j the quarter-blocks g = j NX .. j NX + NX - 1 of the list (block 4, q 0..3), (block 5, q 0..3), ...  One
    // accumulator per slot for the extra wavefronts (a share may straddle two blocks); slot counts are compile-time per
    // role, so each role's K loop is straight-line code (a branch per slot would pin every LDS read behind it).
    const int xg0 = (wave_u - 4) * NX;
    int s_blk[4], s_q[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int sc = s < NX ? s : 0;
        s_blk[s] = main_w ? wave_u : (NX > 0 ? 4 + (xg0 + sc) / 4 : wave_u - 4);
        s_q[s] = main_w ? s : (NX > 0 ? (xg0 + sc) & 3 : 0);
    }
    // ---- A staging (one segment list, as gemm_tile)
    const bool multi = P.kend[0] < K;
    int seg_ld = P.lda[0], seg_k0 = 0, seg_end = P.kend[0], seg_i = 0;
    int a_voff;
    __amdgpu_buffer_rsrc_t arsrc;
    auto open_seg = [&]() {
        const float *base = P.a[seg_i] + (size_t)row0 * seg_ld;
        const long long b = ((long long)(M - 1 - row0) * seg_ld + (seg_end < K ? seg_end : K) - seg_k0) * 4;
        arsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, b < 0x7fffffffLL ? (int)b : 0x7fffffff, 0x00020000);
        const int gr = row0 + (srow & 31);                               // (staged rows 32 .. 63 re-read the tile's rows)
        a_voff = (((gr < M ? gr : M - 1) - row0) * seg_ld + a_kq) * 4;
    };
    open_seg();
    auto prep_seg = [&](int kt) {
        if (!multi) return;
        while (kt * BK >= seg_end) {
            ++seg_i;
            seg_k0 = seg_end;
            seg_ld = P.lda[seg_i];
            seg_end = P.kend[seg_i];
            open_seg();
        }
    };
    f32x4 ra, ra2, ra3;
    auto issue_a = [&](int kt, f32x4 &R) {
        R = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(arsrc, a_voff, (kt * BK - seg_k0) * 4, ACT_AUX));
    };
    const int st_off = srow * LDS_LD + a_kq;
    auto commit_a = [&](int stage, const f32x4 &R) {
        float *sp = stage == 0 ? smem + st_off : stage == 1 ? smem + SF + st_off : smem + 2 * SF + st_off;
        *reinterpret_cast<f32x4 *>(sp) = R;
    };
    // ---- W fragments: one descriptor over the tile's NB column blocks; slot s of K tile kt at scalar offset
    // (s_blk * nk + kt) * 4096 + s_q * 1024 bytes
    __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(P.w + (size_t)(col0 >> 5) * nk * 1024), 0, NB * nk * 4096, 0x00020000);
    const int w_voff = lane * 16;
    int w_soff[4], a_off[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        w_soff[s] = __builtin_amdgcn_readfirstlane((s_blk[s] * nk) * 4096 + s_q[s] * 1024);
        a_off[s] = li * LDS_LD + lh * 16 + s_q[s] * 4;                   // lane (i, h) reads A[row i][k = 16 h + 4 q ..] of a stage
    }
    const int last = nk - 1;
    // One role's whole K loop: NS slots per K tile, slot s into acc[s % NACC] (main: one accumulator, extra: one per slot)
    auto k_loop = [&](auto ns_tag, auto nacc_tag, f32x16 *acc) {
        constexpr int NS = decltype(ns_tag)::value, NACC = decltype(nacc_tag)::value;
        constexpr int NSL = NS > 0 ? NS : 1;                             // (a wavefront without a share still stages A)
        f32x4 rb[NSL], rbn[NSL], rbn2[NSL];
        auto load_w = [&](int kt, f32x4 (&dst)[NSL]) {
            if (NS == 0) return;
#pragma unroll
            for (int s = 0; s < NSL; ++s)
                dst[s] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, w_voff, w_soff[s] + kt * 4096, 0));
        };
        load_w(0, rb);
        load_w(1 < last ? 1 : last, rbn);
        if (late.ndep > 0) wait_deps_reg(late.dw, late.ndep, late.cnt, late.abort_flag, late.spin_ticks);
        {
            f32x4 r0, r1;
            issue_a(0, r0);
            prep_seg(1 < last ? 1 : last);
            issue_a(1 < last ? 1 : last, r1);
            prep_seg(2 < last ? 2 : last);
            issue_a(2 < last ? 2 : last, ra);
            prep_seg(3 < last ? 3 : last);
            issue_a(3 < last ? 3 : last, ra2);
            prep_seg(4 < last ? 4 : last);
            issue_a(4 < last ? 4 : last, ra3);
            prep_seg(5 < last ? 5 : last);
            commit_a(0, r0);
            commit_a(1, r1);
        }
        __syncthreads();
        f32x4 av0 = *reinterpret_cast<const f32x4 *>(smem + a_off[0]);
        R3D_TSTAMP(1);
        int st_cur = 0;
        auto k_tile = [&](int kt, f32x4 (&w_use)[NSL], f32x4 (&w_load)[NSL], f32x4 &stg) {
            const int st_next = st_cur == 2 ? 0 : st_cur + 1, st_next2 = st_next == 2 ? 0 : st_next + 1;
            const float *sp = smem + st_cur * SF;
            commit_a(st_next2, stg);
            load_w(kt + 2 < last ? kt + 2 : last, w_load);
            issue_a(kt + AD < last ? kt + AD : last, stg);
            prep_seg(kt + AD + 1 < last ? kt + AD + 1 : last);
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const f32x4 av = s == 0 ? av0 : *reinterpret_cast<const f32x4 *>(sp + a_off[s]);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
                    acc[s % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk], w_use[s][kk], acc[s % NACC], 0, 0, 0);
            }
            if (NS > 0) av0 = *reinterpret_cast<const f32x4 *>(smem + st_next * SF + a_off[0]);
            __syncthreads();
            st_cur = st_next;
        };
        int kt = 0;
        for (; kt + 2 < nk; kt += 3) {
            k_tile(kt, rb, rbn2, ra);
            k_tile(kt + 1, rbn, rb, ra2);
            k_tile(kt + 2, rbn2, rbn, ra3);
        }
        if (kt < nk) {
            k_tile(kt, rb, rbn2, ra);
            if (kt + 1 < nk) k_tile(kt + 1, rbn, rb, ra2);
        }
    };
    // ---- the two roles (the same number of barriers on both sides)
    constexpr int PART = 16 * 64;                                        // floats of one accumulator in LDS
    f32x16 fin;                                                          // this wavefront's finished 32 x 32 block (writers)
    if (main_w) {
        f32x16 acc[1];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][r] = 0.0f;
        k_loop(std::integral_constant<int, 4>{}, std::integral_constant<int, 1>{}, acc);
        fin = acc[0];
        R3D_TSTAMP(2);
        if (NX > 0) __syncthreads();                                     // (the extra wavefronts publish their partial sums)
    } else {
        constexpr int NA_ = NX > 0 ? NX : 1;
        f32x16 acc[NA_];
#pragma unroll
        for (int a = 0; a < NA_; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][r] = 0.0f;
        k_loop(std::integral_constant<int, NX>{}, std::integral_constant<int, NA_>{}, acc);
        // the split blocks' partial sums -> LDS (the ring is idle: the K loop ended on a barrier), quarter-block g at g * PART;
        // then wavefront 4 + e adds up block 4 + e: quarter-blocks 4 e .. 4 e + 3
        if (NX > 0) {
            float *red = smem + lane;
#pragma unroll
            for (int a = 0; a < NX; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[(xg0 + a) * PART + r * 64] = acc[a][r];
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 16; ++r) fin[r] = 0.0f;
            if (wave_u < NB) {
                const float *src = red + (wave_u - 4) * 4 * PART;
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int r = 0; r < 16; ++r) fin[r] += src[g * PART + r * 64];
            }
        }
    }
    R3D_TSTAMP(3);
    // ---- epilogue: the NB slabs of 32 columns transposed through LDS, 16-byte accesses along the rows (store_tile)
    {
        constexpr int COLS = NB * 32, TPR = COLS / 4;                    // threads per output row
        const int N = P.N;
        const float slope = P.slope;
        const float *res = P.res;
        const int ldc = P.ldc, ldr = P.ldr;
        const __amdgpu_buffer_rsrc_t crs = act_rsrc(P.c + (size_t)row0 * ldc + col0);
        const __amdgpu_buffer_rsrc_t rrs = act_rsrc(res ? res + (size_t)row0 * ldr + col0 : P.c);
        const bool writer = wave_u < NB;
        const float bias = writer ? gload1(P.bias + col0 + wave * 32 + li) : 0.0f;
        __syncthreads();                                                 // the partial sums have been read
        if (writer) {
            float *wr = smem + (4 * lh) * EPI_LD + wave * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) wr[((r & 3) + 8 * (r >> 2)) * EPI_LD] = lrelu(fin[r] + bias, slope);
        }
        __syncthreads();
        const bool vec = col0 + COLS <= N;
#pragma unroll
        for (int it = 0; it < (32 * TPR + GEMM_THREADS - 1) / GEMM_THREADS; ++it) {
            const int c = tid + it * GEMM_THREADS;
            if (c >= 32 * TPR) break;
            const int lr = c / TPR, c4 = (c % TPR) * 4;
            if (row0 + lr >= M) continue;
            f32x4 v = *reinterpret_cast<const f32x4 *>(smem + lr * EPI_LD + c4);
            if (vec) {
                if (res) v += act_load4(rrs, (lr * ldr + c4) * 4);
                act_store4(crs, (lr * ldc + c4) * 4, v);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (col0 + c4 + e < N) act_store1(crs, (lr * ldc + c4 + e) * 4, v[e] + (res ? act_load1(rrs, (lr * ldr + c4 + e) * 4) : 0.0f));
            }
        }
        __syncthreads();
    }
    R3D_TSTAMP(4);
}

// ------------------------------------------------------------------------------------ fp32 on the bf16 matrix cores
//
// gemm_tile_b3: C = res + lrelu(A W^T + b) like gemm_tile, evaluated by v_mfma_f32_32x32x16_bf16 (16x the FLOP
// rate of the fp32 MFMA).  Every fp32 operand is split EXACTLY into three bf16 terms, x = x0 + x1 + x2 (each the
// bf16 rounding of what the previous ones left: 8 + 8 + 8 mantissa bits), and the six products a0b0, a0b1, a1b0,
// a0b2, a1b1, a2b0 are accumulated in fp32, smallest first; the three dropped products are below 2^-24 of the
// leading one.  Measured (tools/bf16x3_probe.cpp, tools/bf16x3_error_table.py): the error against float64 of an fp32
// dot product, 1.1-1.4x the fp32 path's over the whole network (the two dropped cross terms a1b2, a2b1 are each the size
// of one fp32 rounding).
// Weights stay fp32 in memory, packed so that a lane's eight consecutive k of a 16-deep MFMA step are two b128
// loads (r3d_model.cpp: [32-col block][K tile][k16 half][4-float group][lane][4]) and are split in registers by the
// wavefront that owns the column block - each weight is split once per tile, 5.5 VALU instructions per value, hidden
// behind the matrix work - so the weight stream is 4 bytes per value, not the 6 of pre-split planes: a 32-row tile
// is bound by that stream.  Activations are split when their fp32 staging registers are written to the LDS ring
// (a few VALU instructions per thread and K tile), which then holds three bf16 planes.  Used for the FCBlocks'
// 1024-wide Linears; opt-in (r3d_api.cpp, R3D_BF16X3).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int B3_LD = 20;                  // floats per staged row of one plane: 32 bf16 = 64 B + 16 B pad (conflict-free b128 reads)

__device__ __forceinline__ unsigned b3_pack(float a, float b) {
    const f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float b3_lo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float b3_hi(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }

// eight fp32 values -> their three bf16 terms, packed as MFMA operands
__device__ __forceinline__ void b3_split8(const f32x4 &a, const f32x4 &b, bf16x8 (&pl)[3]) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const float x[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    u32x4 p0, p1, p2;
#ifdef R3D_EXP_NOSPLIT_W     // (tools/b3_split_bound.sh: what the weights' split costs - the three planes are the leading term; results are wrong)
#pragma unroll
    for (int i = 0; i < 4; ++i) p0[i] = p1[i] = p2[i] = b3_pack(x[2 * i], x[2 * i + 1]);
    pl[0] = __builtin_bit_cast(bf16x8, p0);
    pl[1] = __builtin_bit_cast(bf16x8, p1);
    pl[2] = __builtin_bit_cast(bf16x8, p2);
    return;
#endif
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned h = b3_pack(x[2 * i], x[2 * i + 1]);
        const float r0 = x[2 * i] - b3_lo(h), r1 = x[2 * i + 1] - b3_hi(h);
        const unsigned m = b3_pack(r0, r1);
        p0[i] = h;
        p1[i] = m;
        p2[i] = b3_pack(r0 - b3_lo(m), r1 - b3_hi(m));
    }
    pl[0] = __builtin_bit_cast(bf16x8, p0);
    pl[1] = __builtin_bit_cast(bf16x8, p1);
    pl[2] = __builtin_bit_cast(bf16x8, p2);
}

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void b3_split4(const f32x4 &x, u32x2 (&pl)[3]);

template <int MI>
__device__ __forceinline__ void gemm_tile_b3(ProbRef P, const int row0, const int col0, float *smem, long long *dbg) {
    R3D_TSTAMP(0);
    constexpr int VR = MI * 32, NA = (VR + 63) / 64;
    constexpr int PLANE = VR * B3_LD, SFB = 3 * PLANE;      // floats per plane / per ring stage
    static_assert(3 * SFB * 4 <= GEMM_LDS_BYTES, "three stages of three planes must fit");
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int M = P.M, K = P.K;
    const int nk = K / BK;
    const int srow = tid >> 3, a_kq = (tid & 7) * 4;
    // ---- A staging (fp32 from HBM, as in gemm_tile)
    const bool multi = P.kend[0] < K;
    int seg_ld = P.lda[0], seg_k0 = 0, seg_end = P.kend[0], seg_i = 0;
    int a_voff[NA];
    __amdgpu_buffer_rsrc_t arsrc;
    auto open_seg = [&]() {
        const float *base = P.a[seg_i] + (size_t)row0 * seg_ld;
        const long long b = ((long long)(M - 1 - row0) * seg_ld + (seg_end < K ? seg_end : K) - seg_k0) * 4;
        arsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, b < 0x7fffffffLL ? (int)b : 0x7fffffff, 0x00020000);
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int gr = row0 + srow + 64 * i;
            a_voff[i] = (((gr < M ? gr : M - 1) - row0) * seg_ld + a_kq) * 4;
        }
    };
    open_seg();
    auto prep_seg = [&](int kt) {
        if (!multi) return;
        while (kt * BK >= seg_end) {
            ++seg_i;
            seg_k0 = seg_end;
            seg_ld = P.lda[seg_i];
            seg_end = P.kend[seg_i];
            open_seg();
        }
    };
    struct Staged { f32x4 a[NA]; };
    Staged ra, ra2, ra3;
    auto issue_a = [&](int kt, Staged &R) {
        const int kb = kt * BK - seg_k0;
#pragma unroll
        for (int i = 0; i < NA; ++i)
            R.a[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(arsrc, a_voff[i], kb * 4, ACT_AUX));
    };
    // registers -> ring stage: split into three bf16 planes (exact: each remainder is representable in fp32)
    const int st_off = srow * B3_LD + (a_kq >> 1);
    auto commit_a = [&](int stage, const Staged &R) {
        float *s = smem + stage * SFB + st_off;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            if (srow + 64 * i >= VR) continue;
            const f32x4 x = R.a[i];
            u32x2 pl[3];
            b3_split4(x, pl);
            float *d = s + i * 64 * B3_LD;
            *reinterpret_cast<u32x2 *>(d) = pl[0];
            *reinterpret_cast<u32x2 *>(d + PLANE) = pl[1];
            *reinterpret_cast<u32x2 *>(d + 2 * PLANE) = pl[2];
        }
    };
    // ---- W fragments: fp32, [(n/32)][K tile][k16 half][4-float group][lane][4]
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(P.wb3 + ((size_t)((col0 >> 5) + wave_u) * nk) * 1024), 0, nk * 4096, 0x00020000);
    const int w_voff = lane * 16;
    struct WFrag { f32x4 f[2][2]; };
    WFrag wa, wb, wc;                        // three sets rotating: weights run two K tiles ahead (an iteration of
                                             // a 32-row tile is shorter than an L2 miss)
    auto load_w = [&](int kt, WFrag &dst) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                dst.f[h][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, w_voff + (h * 2 + j) * 1024, kt * 4096, 0));
    };
    f32x16 acc[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][r] = 0.0f;
    const int a_frag = li * B3_LD + lh * 4;      // + h * 8 floats per k16 half, + mi * 32 rows, + plane
    const int last = nk - 1;
    load_w(0, wa);
    load_w(1 < last ? 1 : last, wb);
    {
        Staged r0, r1;
        issue_a(0, r0);
        prep_seg(1 < last ? 1 : last);
        issue_a(1 < last ? 1 : last, r1);
        prep_seg(2 < last ? 2 : last);
        issue_a(2 < last ? 2 : last, ra);
        prep_seg(3 < last ? 3 : last);
        issue_a(3 < last ? 3 : last, ra2);
        prep_seg(4 < last ? 4 : last);
        issue_a(4 < last ? 4 : last, ra3);
        prep_seg(5 < last ? 5 : last);
        commit_a(0, r0);
        commit_a(1, r1);
    }
    __syncthreads();
    R3D_TSTAMP(1);
    int st_cur = 0;
    // (pinning the split's VALU instructions between the MFMAs of the previous 16-deep step with
    // sched_group_barrier is no faster than what the scheduler does by itself: measured)
    auto k_tile = [&](int kt, const WFrag &w_use, WFrag &w_load, Staged &stg) {
        const int st_next = st_cur == 2 ? 0 : st_cur + 1, st_next2 = st_next == 2 ? 0 : st_next + 1;
        commit_a(st_next2, stg);                 // tile kt+2
        load_w(kt + 2 < last ? kt + 2 : last, w_load);
        issue_a(kt + 5 < last ? kt + 5 : last, stg);
        prep_seg(kt + 6 < last ? kt + 6 : last);
        const float *s = smem + st_cur * SFB + a_frag;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            bf16x8 av[MI][3], wp[3];
            b3_split8(w_use.f[h][0], w_use.f[h][1], wp);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    av[mi][p] = *reinterpret_cast<const bf16x8 *>(s + p * PLANE + mi * 32 * B3_LD + h * 8);
            // (product-major: consecutive MFMAs go to different accumulators - no back-to-back dependent pair when MI > 1)
            constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PW[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
                    acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[mi][PA[t]], wp[PW[t]], acc[mi], 0, 0, 0);
        }
        __syncthreads();
        st_cur = st_next;
    };
    int kt = 0;
    for (; kt + 2 < nk; kt += 3) {
        k_tile(kt, wa, wc, ra);
        k_tile(kt + 1, wb, wa, ra2);
        k_tile(kt + 2, wc, wb, ra3);
    }
    if (kt < nk) {
        k_tile(kt, wa, wc, ra);
        if (kt + 1 < nk) k_tile(kt + 1, wb, wa, ra2);
    }
    R3D_TSTAMP(2);
    R3D_TSTAMP(3);
    store_tile<MI, 1>(P, acc, row0, col0, smem);
    R3D_TSTAMP(4);
}


// ---- pieces shared by the tiles that keep D[channel][row] accumulators (weights as the MFMA's A operand, activations
// as its B operand): a lane then owns one row and, per register quad q, the four consecutive channels
// ch0 + 8 q .. + 3 (ch0 = 32 * wavefront + 4 * (lane / 32)), so activations go to LDS as packed bf16 planes and
// output rows as float4 - no 2- or 4-byte scatter.
struct WFragB3 { f32x4 f[2][2]; };                           // one K tile of a wavefront's 32 channels, fp32: [k16 half][4-float group]
constexpr int B3T_H_PITCH = 264;                             // bf16 per row of an activation plane: 528 B (conflict-free b128 reads)

// four fp32 values -> their three bf16 terms, packed (exact: every remainder is representable in fp32)
__device__ __forceinline__ void b3_split4(const f32x4 &x, u32x2 (&pl)[3]) {
    const unsigned h0 = b3_pack(x[0], x[1]), h1 = b3_pack(x[2], x[3]);
#ifdef R3D_EXP_NOSPLIT_A     // (tools/b3_split_bound.sh: what the activations' split costs; results are wrong)
    pl[0] = pl[1] = pl[2] = u32x2{h0, h1};
    return;
#endif
    const float r0 = x[0] - b3_lo(h0), r1 = x[1] - b3_hi(h0), r2 = x[2] - b3_lo(h1), r3 = x[3] - b3_hi(h1);
    const unsigned m0 = b3_pack(r0, r1), m1 = b3_pack(r2, r3);
    pl[0] = u32x2{h0, h1};
    pl[1] = u32x2{m0, m1};
    pl[2] = u32x2{b3_pack(r0 - b3_lo(m0), r1 - b3_hi(m0)), b3_pack(r2 - b3_lo(m1), r3 - b3_hi(m1))};
}

// gemm_tile_b3 for single-unit tiles, software-pipelined ACROSS the per-K-tile barrier: the operand fragments of tile
// kt+1 (weights split into their three bf16 terms, activation planes read from the ring) are prepared in registers while
// tile kt's MFMAs run, so that the matrix pipe has work from the first instruction after a barrier.  Worth 4-5 % of a
// single-unit tile.  What bounds that tile is elsewhere (tools/coexec_probe2.cpp): a K tile's 24 MFMAs per SIMD take
// 0.40 us by themselves, 0.49 with 8 VALU instructions each (only ~4 per MFMA issue for free), 0.69 with the weight
// loads (4 x b128 per wavefront), 0.79 with the LDS operand reads, 1.02 with the barrier - operand data arriving in
// the VGPRs and the matrix pipe do not overlap, so a tile with one row block per wavefront pays ~0.4 us per K tile for
// its 20 KB of operands per SIMD whatever the order of the instructions.
template <int MI>
__device__ __forceinline__ void gemm_tile_b3p(ProbRef P, const int row0, const int col0, float *smem, long long *dbg) {
    static_assert(MI >= 1 && MI <= 2, "register budget: two sets of operand fragments");
    R3D_TSTAMP(0);
    constexpr int VR = MI * 32, NA = (VR + 63) / 64;
    constexpr int PLANE = VR * B3_LD, SFB = 3 * PLANE;      // floats per plane / per ring stage
    static_assert(3 * SFB * 4 <= GEMM_LDS_BYTES, "three stages of three planes must fit");
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int M = P.M, K = P.K;
    const int nk = K / BK;
    const int srow = tid >> 3, a_kq = (tid & 7) * 4;
    const bool multi = P.kend[0] < K;
    int seg_ld = P.lda[0], seg_k0 = 0, seg_end = P.kend[0], seg_i = 0;
    int a_voff[NA];
    __amdgpu_buffer_rsrc_t arsrc;
    auto open_seg = [&]() {
        const float *base = P.a[seg_i] + (size_t)row0 * seg_ld;
        const long long b = ((long long)(M - 1 - row0) * seg_ld + (seg_end < K ? seg_end : K) - seg_k0) * 4;
        arsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, b < 0x7fffffffLL ? (int)b : 0x7fffffff, 0x00020000);
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int gr = row0 + srow + 64 * i;
            a_voff[i] = (((gr < M ? gr : M - 1) - row0) * seg_ld + a_kq) * 4;
        }
    };
    open_seg();
    auto prep_seg = [&](int kt) {
        if (!multi) return;
        while (kt * BK >= seg_end) {
            ++seg_i;
            seg_k0 = seg_end;
            seg_ld = P.lda[seg_i];
            seg_end = P.kend[seg_i];
            open_seg();
        }
    };
    struct Staged { f32x4 a[NA]; };
    const int last = nk - 1;
    int a_next = 0;                                          // K tiles are requested strictly in order (the segments advance with them)
    auto next_a = [&](Staged &R) {
        const int kt = a_next < last ? a_next : last;
        const int kb = kt * BK - seg_k0;
#pragma unroll
        for (int i = 0; i < NA; ++i)
            R.a[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(arsrc, a_voff[i], kb * 4, ACT_AUX));
        ++a_next;
        prep_seg(a_next < last ? a_next : last);
    };
    const int st_off = srow * B3_LD + (a_kq >> 1);
    auto commit_a = [&](int stage, const Staged &R) {
        float *s = smem + stage * SFB + st_off;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            if (srow + 64 * i >= VR) continue;
            u32x2 pl[3];
            b3_split4(R.a[i], pl);
            float *d = s + i * 64 * B3_LD;
            *reinterpret_cast<u32x2 *>(d) = pl[0];
            *reinterpret_cast<u32x2 *>(d + PLANE) = pl[1];
            *reinterpret_cast<u32x2 *>(d + 2 * PLANE) = pl[2];
        }
    };
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(P.wb3 + ((size_t)((col0 >> 5) + wave_u) * nk) * 1024), 0, nk * 4096, 0x00020000);
    const int w_voff = lane * 16;
    auto load_w = [&](int kt, WFragB3 &dst) {
        const int k = kt < last ? kt : last;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                dst.f[h][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, w_voff + (h * 2 + j) * 1024, k * 4096, 0));
    };
    struct Frags { bf16x8 w[2][3]; bf16x8 a[2][MI][3]; };   // one K tile's MFMA operands: [k16 half][plane]
    auto split_w = [&](const WFragB3 &raw, Frags &F) {
#pragma unroll
        for (int h = 0; h < 2; ++h) b3_split8(raw.f[h][0], raw.f[h][1], F.w[h]);
    };
    const int a_frag = li * B3_LD + lh * 4;
    auto read_a = [&](int stage, Frags &F) {
        const float *s = smem + stage * SFB + a_frag;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    F.a[h][mi][p] = *reinterpret_cast<const bf16x8 *>(s + p * PLANE + mi * 32 * B3_LD + h * 8);
    };
    f32x16 acc[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][r] = 0.0f;
    auto mma = [&](const Frags &F) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F.a[h][mi][2], F.w[h][0], acc[mi], 0, 0, 0);
                acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F.a[h][mi][1], F.w[h][1], acc[mi], 0, 0, 0);
                acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F.a[h][mi][0], F.w[h][2], acc[mi], 0, 0, 0);
                acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F.a[h][mi][1], F.w[h][0], acc[mi], 0, 0, 0);
                acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F.a[h][mi][0], F.w[h][1], acc[mi], 0, 0, 0);
                acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F.a[h][mi][0], F.w[h][0], acc[mi], 0, 0, 0);
            }
    };
    WFragB3 raw0, raw1;                                      // fp32 weights in flight: tile kt + 1 (being split) and kt + 2
    Staged s0, s1, s2, s3;                                   // A tiles kt + 2 .. kt + 5 on their way to the ring
    Frags f0, f1;                                            // operands of tile kt (in use) and kt + 1 (being prepared)
    load_w(0, raw0);
    load_w(1, raw1);
    {
        Staged t0, t1;
        next_a(t0);
        next_a(t1);
        next_a(s0);
        next_a(s1);
        next_a(s2);
        next_a(s3);
        commit_a(0, t0);
        commit_a(1, t1);
    }
    split_w(raw0, f0);
    load_w(2, raw0);
    __syncthreads();
    read_a(0, f0);
    R3D_TSTAMP(1);
    int st_nxt = 1;                                          // ring stage of tile kt + 1
    // iteration kt: the MFMAs of tile kt on `cur`; meanwhile `nxt` <- tile kt + 1 (weights from `raw`, which then
    // takes tile kt + 3; activations from the ring), stage of tile kt + 2 <- `stg`, which then takes tile kt + 6
    auto iter = [&](int kt, const Frags &cur, Frags &nxt, WFragB3 &raw, Staged &stg) {
        const int st_wr = st_nxt == 2 ? 0 : st_nxt + 1;
        mma(cur);
        split_w(raw, nxt);
        load_w(kt + 3, raw);
        read_a(st_nxt, nxt);
        commit_a(st_wr, stg);
        next_a(stg);
        __syncthreads();
        st_nxt = st_wr;
    };
    int kt = 0;
    for (; kt + 3 < nk; kt += 4) {
        iter(kt, f0, f1, raw1, s0);
        iter(kt + 1, f1, f0, raw0, s1);
        iter(kt + 2, f0, f1, raw1, s2);
        iter(kt + 3, f1, f0, raw0, s3);
    }
    if (kt < nk) {
        iter(kt, f0, f1, raw1, s0);
        if (kt + 1 < nk) {
            iter(kt + 1, f1, f0, raw0, s1);
            if (kt + 2 < nk) iter(kt + 2, f0, f1, raw1, s2);
        }
    }
    R3D_TSTAMP(2);
    R3D_TSTAMP(3);
    store_tile<MI, 1>(P, acc, row0, col0, smem);
    R3D_TSTAMP(4);
}

// one 32-deep K tile: weights `w` (fp32, split here) x activations in three planes at `xb` (byte pointer to the K tile's
// first column of row 0 of plane 0; `pitch` bf16 per row, `plane_bytes` between planes); six products per term pair,
// smallest first
template <int MI>
__device__ __forceinline__ void b3t_mma_ktile(const char *xb, const int pitch, const int plane_bytes, const WFragB3 &w,
                                              f32x16 (&acc)[MI], const int li, const int lh) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        bf16x8 wp[3], av[MI][3];
        b3_split8(w.f[h][0], w.f[h][1], wp);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int p = 0; p < 3; ++p)
                av[mi][p] = *reinterpret_cast<const bf16x8 *>(xb + p * plane_bytes + ((mi * 32 + li) * pitch + h * 16 + lh * 8) * 2);
        // (product-major, smallest product first: consecutive MFMAs go to different accumulators)
        constexpr int PW[6] = {0, 1, 2, 0, 1, 0}, PX[6] = {2, 1, 0, 1, 0, 0};
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
                acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wp[PW[t]], av[mi][PX[t]], acc[mi], 0, 0, 0);
    }
}

// accumulators of a D[channel][row] tile start at the layer's bias (one add per value less in the epilogues)
template <int MI>
__device__ __forceinline__ void b3t_init_bias(f32x16 (&acc)[MI], const float *bias, const int ch0) {
    int c = ch0;
    asm volatile("" : "+v"(c));                   // (opaque: the loads stay here instead of being hoisted out of the tile loops, where
                                                  //  the bias vectors of three layers would stay live across all matrix phases)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 bq = gload4(bias + c + 8 * q);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[mi][4 * q + e] = bq[e];
    }
}

// acc <- lrelu(acc) (kept, fp32; the bias is in the sums already) and, split, into the three activation planes at `Hb`
// (pitch B3T_H_PITCH)
template <int MI>
__device__ __forceinline__ void b3t_activate_to_planes(f32x16 (&acc)[MI], const float slope, char *Hb,
                                                       const int plane_bytes, const int li, const int ch0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float t = lrelu(acc[mi][4 * q + e], slope);
                acc[mi][4 * q + e] = t;
                v[e] = t;
            }
            u32x2 pl[3];
            b3_split4(v, pl);
            char *d = Hb + ((mi * 32 + li) * B3T_H_PITCH + ch0 + 8 * q) * 2;
#pragma unroll
            for (int p = 0; p < 3; ++p) *reinterpret_cast<u32x2 *>(d + p * plane_bytes) = pl[p];
        }
    }
}

// gemm_tile_b3t<MI>: a fused pair (GemmProb::w2: a pyramid level's 3-tap and 1x1 convolutions, lib/model/rie.py:94-97)
// on the bf16 matrix cores with D[channel][row] accumulators.  First layer: A from HBM as in gemm_tile_b3 (fp32 staged
// through a three-stage LDS ring as three bf16 planes), weights fp32 in bf16-MFMA operand order split in registers.
// Its activations go to three [32 MI x C] bf16 planes over the dead ring, the 1x1 convolution runs on them barrier-free,
// and the epilogue stages fp32 rows over the planes: + residual, 1 KiB stores.  N <= 256 (one column tile), MI <= 3 (three planes of 96 rows: 152 KB).
template <int MI>
__device__ __forceinline__ void gemm_tile_b3t(ProbRef P, const int row0, float *smem, long long *dbg) {
    static_assert(MI >= 1 && MI <= 3, "the activation planes of a pair tile hold 96 rows (152 KB)");
    R3D_TSTAMP(0);
    constexpr int VR = MI * 32, NA = (VR + 63) / 64;
    constexpr int PLANE = VR * B3_LD, SFB = 3 * PLANE;      // floats per ring plane / per ring stage
    constexpr int H_PLANE = VR * B3T_H_PITCH * 2;           // bytes per activation plane
    static_assert(3 * SFB * 4 <= GEMM_LDS_BYTES && 3 * H_PLANE <= GEMM_LDS_BYTES && VR * PAIR_LD * 4 <= 3 * H_PLANE, "LDS overlays");
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int M = P.M, K = P.K;
    const int nk = K / BK;
    const int srow = tid >> 3, a_kq = (tid & 7) * 4;
    const int ch0 = wave * 32 + 4 * lh;
    // ---- A staging (fp32 from HBM -> three bf16 planes per ring stage), one segment (pair operands are plain)
    __amdgpu_buffer_rsrc_t arsrc;
    int a_voff[NA];
    {
        const int ld = P.lda[0];
        const float *base = P.a[0] + (size_t)row0 * ld;
        const long long b = ((long long)(M - 1 - row0) * ld + K) * 4;
        arsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, b < 0x7fffffffLL ? (int)b : 0x7fffffff, 0x00020000);
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int gr = row0 + srow + 64 * i;
            a_voff[i] = (((gr < M ? gr : M - 1) - row0) * ld + a_kq) * 4;
        }
    }
    struct Staged { f32x4 a[NA]; };
    Staged ra, ra2, ra3;
    auto issue_a = [&](int kt, Staged &R) {
#pragma unroll
        for (int i = 0; i < NA; ++i)
            R.a[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(arsrc, a_voff[i], kt * BK * 4, ACT_AUX));
    };
    const int st_off = srow * B3_LD + (a_kq >> 1);
    auto commit_a = [&](int stage, const Staged &R) {
        float *s = smem + stage * SFB + st_off;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            if (srow + 64 * i >= VR) continue;
            u32x2 pl[3];
            b3_split4(R.a[i], pl);
            float *d = s + i * 64 * B3_LD;
            *reinterpret_cast<u32x2 *>(d) = pl[0];
            *reinterpret_cast<u32x2 *>(d + PLANE) = pl[1];
            *reinterpret_cast<u32x2 *>(d + 2 * PLANE) = pl[2];
        }
    };
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int w_voff = lane * 16;
    auto load_w = [&](__amdgpu_buffer_rsrc_t rs, int kt, WFragB3 &dst) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                dst.f[h][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, w_voff + (h * 2 + j) * 1024, kt * 4096, 0));
    };
    __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(P.wb3 + ((size_t)wave_u * nk) * 1024), 0, nk * 4096, 0x00020000);
    WFragB3 wa, wb, wc;
    f32x16 acc[MI];
    b3t_init_bias<MI>(acc, P.bias, ch0);
    const int last = nk - 1;
    load_w(wrsrc, 0, wa);
    load_w(wrsrc, 1 < last ? 1 : last, wb);
    {
        Staged r0, r1;
        issue_a(0, r0);
        issue_a(1 < last ? 1 : last, r1);
        issue_a(2 < last ? 2 : last, ra);
        issue_a(3 < last ? 3 : last, ra2);
        issue_a(4 < last ? 4 : last, ra3);
        commit_a(0, r0);
        commit_a(1, r1);
    }
    __syncthreads();
    R3D_TSTAMP(1);
    int st_cur = 0;
    auto k_tile = [&](int kt, const WFragB3 &w_use, WFragB3 &w_load, Staged &stg) {
        const int st_next = st_cur == 2 ? 0 : st_cur + 1, st_next2 = st_next == 2 ? 0 : st_next + 1;
        commit_a(st_next2, stg);                 // tile kt+2
        load_w(wrsrc, kt + 2 < last ? kt + 2 : last, w_load);
        issue_a(kt + 5 < last ? kt + 5 : last, stg);
        // ring planes: row pitch B3_LD floats = 2 * B3_LD bf16
        b3t_mma_ktile<MI>(reinterpret_cast<const char *>(smem + st_cur * SFB), 2 * B3_LD, PLANE * 4, w_use, acc, li, lh);
        __syncthreads();
        st_cur = st_next;
    };
    int kt = 0;
    for (; kt + 2 < nk; kt += 3) {
        k_tile(kt, wa, wc, ra);
        k_tile(kt + 1, wb, wa, ra2);
        k_tile(kt + 2, wc, wb, ra3);
    }
    if (kt < nk) {
        k_tile(kt, wa, wc, ra);
        if (kt + 1 < nk) k_tile(kt + 1, wb, wa, ra2);
    }
    R3D_TSTAMP(2);
    // ---- first layer's activations -> planes (the ring is dead: the last k_tile ended with a barrier)
    const int nk2 = P.K2 / BK, last2 = nk2 - 1;
    __amdgpu_buffer_rsrc_t w2rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(P.w2b3 + ((size_t)wave_u * nk2) * 1024), 0, nk2 * 4096, 0x00020000);
    load_w(w2rsrc, 0, wa);
    load_w(w2rsrc, 1 < last2 ? 1 : last2, wb);
    char *Hb = reinterpret_cast<char *>(smem);
    b3t_activate_to_planes<MI>(acc, P.slope, Hb, H_PLANE, li, ch0);
    __syncthreads();
    R3D_TSTAMP(3);
    b3t_init_bias<MI>(acc, P.bias2, ch0);
    {
        auto k_tile2 = [&](int k2, const WFragB3 &w_use, WFragB3 &w_load) {
            load_w(w2rsrc, k2 + 2 < last2 ? k2 + 2 : last2, w_load);
            b3t_mma_ktile<MI>(Hb + k2 * BK * 2, B3T_H_PITCH, H_PLANE, w_use, acc, li, lh);
        };
        int k2 = 0;
        for (; k2 + 2 < nk2; k2 += 3) {
            k_tile2(k2, wa, wc);
            k_tile2(k2 + 1, wb, wa);
            k_tile2(k2 + 2, wc, wb);
        }
        if (k2 < nk2) {
            k_tile2(k2, wa, wc);
            if (k2 + 1 < nk2) k_tile2(k2 + 1, wb, wa);
        }
    }
    // ---- epilogue: lrelu(acc + bias2) as fp32 rows over the planes, then + residual and 1 KiB stores
    __syncthreads();                                         // every wavefront is done reading the planes
    {
        const float slope2 = P.slope2;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = lrelu(acc[mi][4 * q + e], slope2);
                *reinterpret_cast<f32x4 *>(smem + (mi * 32 + li) * PAIR_LD + ch0 + 8 * q) = v;
            }
        }
    }
    __syncthreads();
    {
        const int rd_row = tid >> 6, rd_c4 = (tid & 63) * 4;
        const int N = P.N;
        const float *res = P.res;
        const int ldc = P.ldc, ldr = P.ldr;
        const __amdgpu_buffer_rsrc_t crs = act_rsrc(P.c + (size_t)row0 * ldc);
        const __amdgpu_buffer_rsrc_t rrs = act_rsrc(res ? res + (size_t)row0 * ldr : P.c);
#pragma unroll
        for (int j = 0; j < 4 * MI; ++j) {
            const int lr = rd_row + 8 * j, row = row0 + lr;
            if (row >= M) continue;
            f32x4 v = *reinterpret_cast<const f32x4 *>(smem + lr * PAIR_LD + rd_c4);
            if (rd_c4 + 4 <= N) {
                if (res) v += act_load4(rrs, (lr * ldr + rd_c4) * 4);
                act_store4(crs, (lr * ldc + rd_c4) * 4, v);
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (rd_c4 + c < N) act_store1(crs, (lr * ldc + rd_c4 + c) * 4, v[c] + (res ? act_load1(rrs, (lr * ldr + rd_c4 + c) * 4) : 0.0f));
            }
        }
    }
    __syncthreads();
    R3D_TSTAMP(4);
}

// ------------------------------------------------------------------------------------ UV input mode
//
// get_cam_ray_given_uv (lib/camera/camera.py:460-471) applied to a gathered value on its way into LDS: the operand
// column says which ray component it is (two low bits of its table entry), the operand row which window - hence
// which camera row {fx, fy, cx, cy, cos(pitch), sin(pitch)} - it belongs to.  float64 then cast, exactly as the
// reference encodes on the host (NumPy float64) and casts at lib/train_val/trainer.py:298: the result is bit-identical
// to feeding the host-encoded rays.
struct CamRow { double fx, fy, cx, cy, c, s; };
typedef double f64x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ CamRow load_cam_row(const double *row) {
    const f64x2 a = *(const R3D_AS1 f64x2 *)row, b = *(const R3D_AS1 f64x2 *)(row + 2), c = *(const R3D_AS1 f64x2 *)(row + 4);
    return CamRow{a[0], a[1], b[0], b[1], c[0], c[1]};
}
__device__ __forceinline__ float uv_to_ray(const float px, const int code, const CamRow &k) {
    const int f = code & 3;
    const double t = ((double)px - (f == 0 ? k.cx : k.cy)) / (f == 0 ? k.fx : k.fy);     // x = (u-cx)/fx, y = (v-cy)/fy
    const double r = f == 1 ? k.c * t + k.s : -k.s * t + k.c;                            // [x, y, 1] @ Rx(pitch)^T
    return (float)(f == 0 ? t : r);
}

// ------------------------------------------------------------------------------------ first layers
//
// r3d_gemm_enc_f32: expand_conv of every temporal branch and GlobalInfo.fc_1, with the input encoding
// fused in.  A[row][k] = x[.. + off1] - x[.. + off2] (positional / temporal differences and body-part
// gather, lib/model/rie.py:290-357; window gather from a batch or a sliding clip,
// lib/train_val/trainer.py:47-58) is address arithmetic - VALU work - and VALU instructions of a
// wavefront crawl while its SIMD partner streams MFMAs.  So this kernel does not interleave the two:
// a tile's WHOLE encoded operand (<= 96 rows x K <= 480) is built in LDS first, then a barrier-free
// MFMA loop consumes it with the weight fragments streaming from HBM.  Two workgroups share a CU
// (<= 66 KiB LDS, <= 128 VGPRs each), so one workgroup's encoding overlaps the other's MFMAs.

constexpr int ENC_TILE_BYTES = 64 * 1024;                  // encoded tile: rows * (K + 4) floats
constexpr int ENC_LDS_BYTES = ENC_TILE_BYTES + LUT_LDS_INTS * 4;

template <int MI, bool UV>
__device__ __forceinline__ void enc_tile(ProbRef P, const int row0, const int col0, const bool new_prob, float *smem, long long *dbg) {
    R3D_TSTAMP(0);
    constexpr int R = MI * 32;
    constexpr int NA = (R + 63) / 64;
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int M = P.M, K = P.K;
    const int nk = K / BK;
    const int ldt = K + 4;                                  // (K+4)*4 B = odd multiple of 16 B: conflict-free b128 rows
    const int srow = tid >> 3, a_kq = (tid & 7) * 4;
    int *lut_lds = reinterpret_cast<int *>(smem + ENC_TILE_BYTES / 4);

    __syncthreads();                                        // the previous tile's MFMA loop is done with LDS
    const int *lut1 = lut_lds, *lutk = lut_lds + K;
    if (new_prob) {
        for (int i = tid; i < K + K / 4; i += GEMM_THREADS) lut_lds[i] = *(const R3D_AS1 int *)(P.lut + i);
        __syncthreads();
    }
    // ---- build the operand tile: one gathered input element per column (the differences of the reference's
    // encoding live in the folded weights, r3d_internal.hpp).  Columns are grouped by base, so the four columns of
    // a staging thread share it; padding columns read 0 through the descriptor.
    __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.x), 0, P.enc_bytes, 0x00020000);
    unsigned b_first[NA], b_cur[NA];                 // byte offsets into the raw input
    bool on[NA];
    CamRow camr[UV ? NA : 1];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int vr = srow + 64 * i;
        on[i] = vr < R;
        const int gr = row0 + vr;
        const int row = gr < M ? gr : M - 1;
        const int win = row / P.enc_rows, t3 = row - win * P.enc_rows;
        const unsigned wbase = (unsigned)win * (unsigned)P.enc_ws;
        b_first[i] = (wbase + (unsigned)(t3 * P.enc_step * P.enc_jf)) * 4;   // first frame of the row (rows step by 3 frames; 1 for the dense ablation)
        b_cur[i] = (wbase + (unsigned)P.enc_cur) * 4;               // the window's "current" frame (quirk Q1)
        if constexpr (UV) camr[i] = load_cam_row(P.cam + (long long)win * P.cam_stride);
    }
    struct Raw { f32x4 a[NA]; };
    auto issue = [&](int kt, Raw &r) {
        const int k = kt * BK + a_kq;
        const int4 o1 = *reinterpret_cast<const int4 *>(lut1 + k);
        const bool cur = lutk[k >> 2] != 0;
        const int c1[4] = {o1.x & ~3, o1.y & ~3, o1.z & ~3, o1.w & ~3};   // (UV tables: ray component in the low bits)
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            if (!on[i]) continue;
            const unsigned b = cur ? b_cur[i] : b_first[i];
#pragma unroll
            for (int e = 0; e < 4; ++e)
                r.a[i][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrsrc, b + (unsigned)c1[e], 0, 0));
        }
    };
    auto commit = [&](int kt, const Raw &r) {
        int4 code = make_int4(0, 0, 0, 0);
        if constexpr (UV) code = *reinterpret_cast<const int4 *>(lut1 + kt * BK + a_kq);
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            if (!on[i]) continue;
            f32x4 v = r.a[i];
            if constexpr (UV) {
                v[0] = uv_to_ray(v[0], code.x, camr[i]);
                v[1] = uv_to_ray(v[1], code.y, camr[i]);
                v[2] = uv_to_ray(v[2], code.z, camr[i]);
                v[3] = uv_to_ray(v[3], code.w, camr[i]);
            }
            *reinterpret_cast<f32x4 *>(smem + (srow + 64 * i) * ldt + kt * BK + a_kq) = v;
        }
    };
    {
        Raw r0, r1;
        issue(0, r0);
        int kt = 0;
        for (; kt + 1 < nk; kt += 2) {
            issue(kt + 1, r1);
            commit(kt, r0);
            if (kt + 2 < nk) issue(kt + 2, r0);
            commit(kt + 1, r1);
        }
        if (kt < nk) commit(kt, r0);
    }
    __syncthreads();
    R3D_TSTAMP(1);

    // ---- barrier-free MFMA loop: A fragments from the LDS tile, W fragments straight from HBM
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(P.w + ((size_t)((col0 >> 5) + wave_u) * nk) * 1024), 0, nk * 4096, 0x00020000);
    const int w_voff = lane * 16;
    f32x4 rb[4], rbn[4];
    auto load_w = [&](int kt, f32x4 (&dst)[4]) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            dst[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, w_voff + q * 1024, kt * 4096, 0));
    };
    f32x16 acc[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][r] = 0.0f;
    const float *a_frag = smem + li * ldt + lh * 16;
    const int last = nk - 1;
    auto k_tile = [&](int kt, f32x4 (&w_use)[4], f32x4 (&w_load)[4]) {
        load_w(kt + 1 < last ? kt + 1 : last, w_load);
        __builtin_amdgcn_sched_barrier(0);                    // (pinned, as in first_level_taps: the scheduler would sink the request to its use)
        const float *s = a_frag + kt * BK;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 av[MI];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) av[mi] = *reinterpret_cast<const f32x4 *>(s + mi * 32 * ldt + q * 4);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
                    acc[mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mi][kk], w_use[q][kk], acc[mi], 0, 0, 0);
        }
    };
    load_w(0, rb);
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {
        k_tile(kt, rb, rbn);
        k_tile(kt + 1, rbn, rb);
    }
    if (kt < nk) k_tile(kt, rb, rbn);

    // ---- epilogue: C = lrelu(acc + bias) through the LDS transpose (the encoded tile is dead by now)
    R3D_TSTAMP(2);
    R3D_TSTAMP(3);
    store_tile<MI, 1>(P, acc, row0, col0, smem);
    R3D_TSTAMP(4);
}

constexpr int FL_LUT_INTS = 320;                           // first-layer tables in LDS: K0 + K0/4 ints, K0 <= 256

// ------------------------------------------------------------------------------------ first level, tap by tap
//
// first_level_taps: expand_conv (on the gathered input), the level-1 3-tap convolution and its 1x1 convolution
// (lib/model/rie.py:85-97 up to the end of the first loop iteration) for 32 * MI output rows (MI = 1, 2), without the
// expand_conv output - the largest activation of the network, 127 MB at 256 windows - ever leaving the CU.  The tile is
// organised around the taps of the 3-tap convolution (round 1's form held the 96 expand_conv rows of a 32-row tile at once).  Output row r of the level reads the
// expand_conv rows 3r, 3r+1, 3r+2 - one per tap - so the tile walks the taps: gather the raw elements of the rows
// {3r + tap}, run expand_conv on them (activations -> H, one [32 MI x C] buffer in MFMA operand order), multiply H
// with the tap's third of the 3-tap weights into the level's accumulators, next tap.  Only ONE tap's activations
// are alive at a time, which is what lets a tile hold 64 output rows in 67 KB where the row-major form needed 100 KB
// for 32: every weight fragment of the two big layers now feeds two row blocks (a single-row-block K loop is bound
// by the weight stream), and half as many weight bytes cross the L2.  The residual tap (centre;
// the last one for causal models, rie.py:92-94) is visited LAST and its activations simply stay in the expand
// accumulators: they have the C layout of the final accumulators (same wavefront, same columns, same rows), so the
// residual add of the epilogue is register + register.  (Summation order over the taps therefore differs from the
// reference's k-major order; the result is the same to fp32 rounding.)
//   per tap:  raw values (requested one phase earlier) -> G (encoded on the way in UV mode) | barrier |
//             request the next phase's raw values | expand_conv MFMAs on G          [K0 > 64: in chunks of 64 columns]
//             activations -> H | barrier | 3-tap partial: C/32 K tiles, weights streaming, barrier-free
//   then:     level activations -> H | barrier | 1x1 convolution | epilogue (+ residual from registers) through H.
// G is double buffered, so a phase costs one barrier.
constexpr int FLT_MAX_MI = 2;
constexpr int FLT_H_FLOATS = FLT_MAX_MI * 32 * PAIR_LD;                  // 16,640
constexpr int FLT_G_LD = 64 + 4;                                         // a 64-column chunk, conflict-free b128 rows
constexpr int FLT_G_FLOATS = FLT_MAX_MI * 32 * FLT_G_LD;                 //  4,352 per buffer
constexpr int FLT_LUT_OFF = FLT_H_FLOATS + 2 * FLT_G_FLOATS;
static_assert((FLT_LUT_OFF + FL_LUT_INTS) * 4 <= GEMM_LDS_BYTES, "the tap-wise first level fits the GEMM kernel's LDS allocation");
static_assert((2 * FLT_H_FLOATS + 256) * 4 <= GEMM_LDS_BYTES, "first_level_shared: two activation tiles and the row tables");

template <int MI, bool MULTI, bool UV>   // MULTI: K0 > 64 (several 64-column chunks per tap: the trajectory model)
__device__ __forceinline__ void first_level_taps(ProbRef P, const int4 *tile_list, const int tstride, const int ntiles, const bool new_prob, float *smem, const gu32 cnt,
                                                 long long *dbg_base) {
    static_assert(MI >= 1 && MI <= FLT_MAX_MI, "tile height");
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int w_voff = lane * 16;
    const int srow = tid >> 3, a_kq = (tid & 7) * 4;         // staging: 64 rows x 8 threads x 4 columns per K tile
    const int K0 = P.K, nk0 = K0 / BK, nch = (nk0 + 1) >> 1;
    const int M = P.M, M0 = 3 * M;
    const int res_tap = P.res_tap;
    float *H = smem, *G0 = smem + FLT_H_FLOATS;
    int *lut_lds = reinterpret_cast<int *>(smem + FLT_LUT_OFF);
    const int *lut1 = lut_lds, *lutk = lut_lds + K0;
    if (new_prob) {
        for (int i = tid; i < K0 + K0 / 4; i += GEMM_THREADS) lut_lds[i] = *(const R3D_AS1 int *)(P.lut + i);
    }
    __syncthreads();                                         // (also: the previous tile is done with LDS)
    __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.x), 0, P.enc_bytes, 0x00020000);
    auto load_frag = [&](__amdgpu_buffer_rsrc_t rs, int kt, f32x4 (&dst)[4]) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            dst[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, w_voff + q * 1024, kt * 4096, 0));
    };
    const int nk1 = P.K2 / BK, tiles_per_tap = nk1 / 3, nk2 = P.K3 / BK;
    __amdgpu_buffer_rsrc_t w0rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(P.w + ((size_t)wave_u * nk0) * 1024), 0, nk0 * 4096, 0x00020000);
    __amdgpu_buffer_rsrc_t w1rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(P.w2 + ((size_t)wave_u * nk1) * 1024), 0, nk1 * 4096, 0x00020000);
    __amdgpu_buffer_rsrc_t w2rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(P.w3 + ((size_t)wave_u * nk2) * 1024), 0, nk2 * 4096, 0x00020000);
    const float slope0 = P.slope, slope1 = P.slope2, slope2 = P.slope3;
    // a layer's bias (this lane's column), loaded where its accumulators are initialised: the opaque index keeps the
    // loads from being hoisted out of the tile loop, where three more values would be live across every matrix phase
    auto bias_at = [&](const float *b) {
        int c = wave * 32 + li;
        asm volatile("" : "+v"(c));
        return gload1(b + c);
    };

    // ---- gather state of the phase whose raw values are in flight / in registers
    struct Raw { f32x4 a[2]; };                              // the two K tiles of a 64-column chunk
    Raw gq;
    unsigned b_first, b_cur;
    const bool on = srow < MI * 32;
    CamRow camr;
    auto issue_phase = [&](int row0, int tap, int ch) {      // tile rows [row0, row0 + 32 MI), expand_conv rows 3r + tap
        const int orow = row0 + srow;
        const int e = 3 * (orow < M ? orow : M - 1) + tap;
        const int win = e / P.enc_rows, t3 = e - win * P.enc_rows;
        const unsigned wbase = (unsigned)win * (unsigned)P.enc_ws;
        b_first = (wbase + (unsigned)(t3 * 3 * P.enc_jf)) * 4;
        b_cur = (wbase + (unsigned)P.enc_cur) * 4;
        if constexpr (UV) camr = load_cam_row(P.cam + (long long)win * P.cam_stride);
        if (!on) return;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int k = (ch * 2 + h) * BK + a_kq;
            if (MULTI && k >= K0) break;                     // (uniform: the last chunk may hold one K tile)
            const int4 o1 = *reinterpret_cast<const int4 *>(lut1 + k);
            const unsigned b = lutk[k >> 2] != 0 ? b_cur : b_first;
            const int c1[4] = {o1.x & ~3, o1.y & ~3, o1.z & ~3, o1.w & ~3};
#pragma unroll
            for (int c = 0; c < 4; ++c)
                gq.a[h][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrsrc, b + (unsigned)c1[c], 0, 0));
        }
    };
    auto commit_phase = [&](int ch, float *G) {
        if (!on) return;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int k = (ch * 2 + h) * BK + a_kq;
            if (MULTI && k >= K0) break;
            f32x4 v = gq.a[h];
            if constexpr (UV) {
                const int4 code = *reinterpret_cast<const int4 *>(lut1 + k);
                v[0] = uv_to_ray(v[0], code.x, camr);
                v[1] = uv_to_ray(v[1], code.y, camr);
                v[2] = uv_to_ray(v[2], code.z, camr);
                v[3] = uv_to_ray(v[3], code.w, camr);
            }
            *reinterpret_cast<f32x4 *>(G + srow * FLT_G_LD + h * BK + a_kq) = v;
        }
    };

    f32x4 rb[4], rbn[4], rbn2[4];                            // streaming weight fragments (three sets rotating)
    // expand_conv fragments: two K tiles per 64-column chunk.  One chunk (K0 <= 64): resident in w0a/w0b for the whole
    // run.  Several: chunks alternate between the sets (w0a, w0b) and (w0c, w0d), the next chunk's fragments requested
    // in front of the current chunk's matrix work; chunk 0 returns to the first set after every tap's 3-tap loop.
    f32x4 w0a[4], w0b[4], w0c[4], w0d[4];
    load_frag(w0rsrc, 0, w0a);
    load_frag(w0rsrc, nk0 > 1 ? 1 : 0, w0b);
    int phase = 0;                                           // parity selects the G buffer
    auto tap_of = [&](int ts) { return ts == 0 ? 0 : ts == 2 ? res_tap : 3 - res_tap; };   // the residual tap comes last
    issue_phase(__builtin_amdgcn_readfirstlane(tile_list[0].y), 0, 0);
#pragma unroll 1
    for (int ti = 0; ti < ntiles; ++ti) {
        const int row0 = __builtin_amdgcn_readfirstlane(tile_list[ti * tstride].y);
        const int next_row0 = ti + 1 < ntiles ? __builtin_amdgcn_readfirstlane(tile_list[(ti + 1) * tstride].y) : -1;
#ifdef R3D_TIMING
        long long *dbg = dbg_base && ti < 8 ? dbg_base + ti * 8 : nullptr;
#else
        long long *dbg = nullptr;
        (void)dbg;
#endif
        R3D_TSTAMP(0);
        f32x16 acc0[MI], acc1[MI];                           // (accumulators start at the layer's bias)
        const float b1v = bias_at(P.bias2);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[mi][r] = b1v;
#pragma unroll 1
        for (int ts = 0; ts < 3; ++ts) {
            const int tap = tap_of(ts);
            const float b0v = bias_at(P.bias);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc0[mi][r] = b0v;
            // ---- expand_conv on the rows {3r + tap}, one 64-column chunk of the operand per phase
            auto chunk = [&](int ch, f32x4 (&ua)[4], f32x4 (&ub)[4], f32x4 (&la)[4], f32x4 (&lb)[4]) {
                float *G = G0 + (phase & 1) * FLT_G_FLOATS;
                commit_phase(ch, G);
                __syncthreads();
                if constexpr (MULTI) {
                    if (ch + 1 < nch) {
                        load_frag(w0rsrc, (ch + 1) * 2, la);
                        load_frag(w0rsrc, (ch + 1) * 2 + 1 < nk0 ? (ch + 1) * 2 + 1 : (ch + 1) * 2, lb);
                    }
                }
                // the next phase's raw values, in front of this phase's matrix work
                if (ch + 1 < nch) issue_phase(row0, tap, ch + 1);
                else if (ts < 2) issue_phase(row0, tap_of(ts + 1), 0);
                else if (next_row0 >= 0) issue_phase(next_row0, 0, 0);
                const float *a_frag = G + li * FLT_G_LD + lh * 16;
                const int nkc = ch * 2 + 1 < nk0 ? 2 : 1;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    if (h >= nkc) break;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        f32x4 av[MI];
#pragma unroll
                        for (int mi = 0; mi < MI; ++mi) av[mi] = *reinterpret_cast<const f32x4 *>(a_frag + mi * 32 * FLT_G_LD + h * BK + q * 4);
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                            for (int mi = 0; mi < MI; ++mi)
                                acc0[mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mi][kk], h == 0 ? ua[q][kk] : ub[q][kk], acc0[mi], 0, 0, 0);
                    }
                }
                ++phase;
            };
            if constexpr (MULTI) {
#pragma unroll 1
                for (int ch = 0; ch < nch; ch += 2) {
                    chunk(ch, w0a, w0b, w0c, w0d);
                    if (ch + 1 < nch) chunk(ch + 1, w0c, w0d, w0a, w0b);
                }
            } else {
                chunk(0, w0a, w0b, w0c, w0d);
            }
            if (ts == R3D_TS) R3D_TSTAMP(5);
            // ---- activations (in place: the residual tap's stay in acc0 for the epilogue) -> H
            load_frag(w1rsrc, tap * tiles_per_tap, rb);
            load_frag(w1rsrc, __builtin_amdgcn_readfirstlane(tap * tiles_per_tap + (tiles_per_tap > 1 ? 1 : 0)), rbn);   // (else: four waterfall loops)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                float *wr = H + (mi * 32 + 4 * lh) * PAIR_LD + wave * 32 + li;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = lrelu(acc0[mi][r], slope0);
                    acc0[mi][r] = v;
                    wr[((r & 3) + 8 * (r >> 2)) * PAIR_LD] = v;
                }
            }
            __syncthreads();
            if (ts == R3D_TS) R3D_TSTAMP(6);
            // ---- this tap's third of the 3-tap convolution: K = C, barrier-free, weights two K tiles ahead
            {
                const float *h_frag = H + li * PAIR_LD + lh * 16;
                const int kbase = tap * tiles_per_tap, lastk = tiles_per_tap - 1;
                auto k_tile1 = [&](int kin, f32x4 (&w_use)[4], f32x4 (&w_load)[4]) {
                    load_frag(w1rsrc, kbase + (kin + 2 < lastk ? kin + 2 : lastk), w_load);
                    // (pinned: left to itself the scheduler sinks these requests to their uses - one group of 8 MFMAs ahead instead of two K tiles, and
                    //  `s_waitcnt vmcnt(0)` four times per iteration; same registers, same results, -0.6 .. 0.9 % at 1024 windows: profiles/r06_pin_prefetch/)
                    __builtin_amdgcn_sched_barrier(0);
                    const float *sp = h_frag + kin * BK;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        f32x4 av[MI];
#pragma unroll
                        for (int mi = 0; mi < MI; ++mi) av[mi] = *reinterpret_cast<const f32x4 *>(sp + mi * 32 * PAIR_LD + q * 4);
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                            for (int mi = 0; mi < MI; ++mi)
                                acc1[mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mi][kk], w_use[q][kk], acc1[mi], 0, 0, 0);
                    }
                };
                int kin = 0;
                for (; kin + 2 < tiles_per_tap; kin += 3) {
                    k_tile1(kin, rb, rbn2);
                    k_tile1(kin + 1, rbn, rb);
                    k_tile1(kin + 2, rbn2, rbn);
                }
                if (kin < tiles_per_tap) {
                    k_tile1(kin, rb, rbn2);
                    if (kin + 1 < tiles_per_tap) k_tile1(kin + 1, rbn, rb);
                }
            }
            // (no barrier here: the next tap's first chunk phase has one between this loop and the next write of H)
            if (ts == R3D_TS) R3D_TSTAMP(7);
            if constexpr (MULTI) {
                if (ts < 2) {                                // chunk 0 again for the next tap (the streaming sets are dead here)
                    load_frag(w0rsrc, 0, w0a);
                    load_frag(w0rsrc, nk0 > 1 ? 1 : 0, w0b);
                }
            }
        }
        R3D_TSTAMP(1);
        // ---- level activations -> H; the 1x1 convolution on them
        load_frag(w2rsrc, 0, rb);
        load_frag(w2rsrc, nk2 > 1 ? 1 : 0, rbn);
        __syncthreads();                                     // every wavefront is done reading the last tap's H
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            float *wr = H + (mi * 32 + 4 * lh) * PAIR_LD + wave * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) wr[((r & 3) + 8 * (r >> 2)) * PAIR_LD] = lrelu(acc1[mi][r], slope1);
        }
        __syncthreads();
        R3D_TSTAMP(2);
        const float b2v = bias_at(P.bias3);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[mi][r] = b2v;
        {
            const float *h_frag = H + li * PAIR_LD + lh * 16;
            const int last2 = nk2 - 1;
            auto k_tile2 = [&](int kt, f32x4 (&w_use)[4], f32x4 (&w_load)[4]) {
                load_frag(w2rsrc, kt + 2 < last2 ? kt + 2 : last2, w_load);
                // (pinned: left to itself the scheduler sinks these requests to their uses - one group of 8 MFMAs ahead instead of two K tiles, and
                //  `s_waitcnt vmcnt(0)` four times per iteration; same registers, same results, -0.6 .. 0.9 % at 1024 windows: profiles/r06_pin_prefetch/)
                __builtin_amdgcn_sched_barrier(0);
                const float *sp = h_frag + kt * BK;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 av[MI];
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) av[mi] = *reinterpret_cast<const f32x4 *>(sp + mi * 32 * PAIR_LD + q * 4);
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                        for (int mi = 0; mi < MI; ++mi)
                            acc1[mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mi][kk], w_use[q][kk], acc1[mi], 0, 0, 0);
                }
            };
            int kt = 0;
            for (; kt + 2 < nk2; kt += 3) {
                k_tile2(kt, rb, rbn2);
                k_tile2(kt + 1, rbn, rb);
                k_tile2(kt + 2, rbn2, rbn);
            }
            if (kt < nk2) {
                k_tile2(kt, rb, rbn2);
                if (kt + 1 < nk2) k_tile2(kt + 1, rbn, rb);
            }
        }
        R3D_TSTAMP(3);
        if constexpr (MULTI) {
            if (next_row0 >= 0) {                            // chunk 0 for the next tile: lands behind the epilogue
                load_frag(w0rsrc, 0, w0a);
                load_frag(w0rsrc, nk0 > 1 ? 1 : 0, w0b);
            }
        }
        // ---- epilogue: + the residual tap's activations (registers), rows transposed through H, 1 KiB stores
        __syncthreads();                                     // every wavefront is done reading H
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            float *wr = H + (mi * 32 + 4 * lh) * PAIR_LD + wave * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) wr[((r & 3) + 8 * (r >> 2)) * PAIR_LD] = lrelu(acc1[mi][r], slope2) + acc0[mi][r];
        }
        __syncthreads();
        {
            const int rd_row = tid >> 6, rd_c4 = (tid & 63) * 4;
            const int N = P.N, ldc = P.ldc;
            const __amdgpu_buffer_rsrc_t crs = act_rsrc(P.c + (size_t)row0 * ldc);
#pragma unroll
            for (int j = 0; j < 4 * MI; ++j) {
                const int lr = rd_row + 8 * j, row = row0 + lr;
                if (row >= M) continue;
                const f32x4 v = *reinterpret_cast<const f32x4 *>(H + lr * PAIR_LD + rd_c4);
                if (rd_c4 + 4 <= N) {
                    act_store4(crs, (lr * ldc + rd_c4) * 4, v);
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (rd_c4 + c < N) act_store1(crs, (lr * ldc + rd_c4 + c) * 4, v[c]);
                }
            }
        }
        if (cnt) tile_drain();
        __syncthreads();                                     // H is free for the next tile's activations
        if (cnt) {
            const int4 te = tile_list[ti * tstride + 1];     // {dependencies (none), first ready counter, granules, -}
            tile_signal(cnt, __builtin_amdgcn_readfirstlane(te.y), __builtin_amdgcn_readfirstlane(te.z), MI);
        }
        R3D_TSTAMP(4);
    }
    (void)M0;
}


// ------------------------------------------------------------------------------------ first level of a clip call
//
// first_level_shared: first_level_taps for calls whose windows slide over a clip one frame at a time
// (lib/train_val/trainer.py:47-58).  expand_conv is linear in its operand, so the pre-activation of expand_conv row t of
// window w is E[f] + V[c]: E of the row's first frame f = w * stride + 3 t, V of the window's current frame c (quirk Q1) -
// two C-vectors per input FRAME that a launch of gathered GEMMs ahead of the forward left in the per-frame buffer
// (r3d_api.cpp: Plan::frame_probs; row = frame, this branch's block [E | V] at P.x, P.enc_jf floats per row).  The 81
// windows that contain a frame share them: the tile neither gathers nor multiplies for expand_conv - every lane loads the
// 16 (row, column) values of its accumulator registers straight into them (two 128-byte row segments per wavefront
// instruction), one tap ahead of their use and behind the previous tap's matrix work; V once per tile.  The rest - the
// tap-wise 3-tap convolution, the 1x1 convolution, the residual tap's activations kept in registers - is first_level_taps.
template <int MI>
__device__ __forceinline__ void first_level_shared(ProbRef P, const int4 *tile_list, const int tstride, const int ntiles, float *smem, const gu32 cnt,
                                                   long long *dbg_base) {
    static_assert(MI >= 1 && MI <= FLT_MAX_MI, "tile height");
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int w_voff = lane * 16;
    const int M = P.M;
    const int res_tap = P.res_tap;
    const int rpw = P.enc_rows / 3;                           // output rows per window
    const int ld4 = P.enc_jf * 4;                             // bytes per row of the per-frame buffer
    float *H = smem;
    int *rowtab = reinterpret_cast<int *>(smem + FLT_H_FLOATS);   // [2][2][64]: (E row offset, V row offset) of a tile's rows, double buffered
    float *HR = smem + FLT_H_FLOATS + 256;                    // the residual tap's activations: operand of its matrix phase AND kept for the
                                                              // epilogue (32 MI registers less than holding them, which is what spilled)
    __syncthreads();                                          // (the previous tile is done with LDS)
    const __amdgpu_buffer_rsrc_t frs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.x), 0, P.enc_bytes, 0x00020000);
    const int col = wave * 32 + li;
    const unsigned ecol = col < P.N ? (unsigned)col * 4u : 0x7ffffff0u;             // (columns past C: beyond the descriptor's bound - zeros)
    const unsigned vcol = col < P.N ? (unsigned)(P.N + col) * 4u : 0x7ffffff0u;
    auto load_frag = [&](__amdgpu_buffer_rsrc_t rs, int kt, f32x4 (&dst)[4]) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            dst[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, w_voff + q * 1024, kt * 4096, 0));
    };
    const int nk1 = P.K2 / BK, tiles_per_tap = nk1 / 3, nk2 = P.K3 / BK;
    __amdgpu_buffer_rsrc_t w1rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(P.w2 + ((size_t)wave_u * nk1) * 1024), 0, nk1 * 4096, 0x00020000);
    __amdgpu_buffer_rsrc_t w2rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(P.w3 + ((size_t)wave_u * nk2) * 1024), 0, nk2 * 4096, 0x00020000);
    const float slope0 = P.slope, slope1 = P.slope2, slope2 = P.slope3;
    auto bias_at = [&](const float *b) {
        int c = wave * 32 + li;
        asm volatile("" : "+v"(c));
        return gload1(b + c);
    };
    auto tap_of = [&](int ts) { return ts == 0 ? 0 : ts == 2 ? res_tap : 3 - res_tap; };   // the residual tap comes last
    // rows of a tile -> byte offsets of their E row (tap 0) and of their window's V row
    auto fill_rowtab = [&](int row0, int buf) {
        if (tid < MI * 32) {
            const int orow = row0 + tid < M ? row0 + tid : M - 1;
            const int win = orow / rpw, j = orow - win * rpw;
            const unsigned wb = (unsigned)win * (unsigned)P.enc_ws;
            rowtab[buf * 128 + tid] = (int)(wb * 4u + (unsigned)(9 * j) * (unsigned)ld4);
            rowtab[buf * 128 + 64 + tid] = (int)((wb + (unsigned)P.enc_cur) * 4u);
        }
    };
    f32x16 acc0[MI], vt[MI], acc1[MI];
    auto issue_e = [&](int buf, int tap) {
        const int so = __builtin_amdgcn_readfirstlane(tap * 3 * ld4);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int lr = mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                acc0[mi][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(frs, (unsigned)rowtab[buf * 128 + lr] + ecol, so, 0));
                if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);       // (four row offsets at a time: hoisting all the table reads spills)
            }
    };
    auto issue_v = [&](int buf) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int lr = mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                vt[mi][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(frs, (unsigned)rowtab[buf * 128 + 64 + lr] + vcol, 0, 0));
                if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
    };
    f32x4 rb[4], rbn[4], rbn2[4];                            // streaming weight fragments (three sets rotating: 555 k against 544 k poses/s with two)
    fill_rowtab(__builtin_amdgcn_readfirstlane(tile_list[0].y), 0);
    __syncthreads();
    issue_e(0, 0);
    issue_v(0);
#pragma unroll 1
    for (int ti = 0; ti < ntiles; ++ti) {
        const int row0 = __builtin_amdgcn_readfirstlane(tile_list[ti * tstride].y);
        const int next_row0 = ti + 1 < ntiles ? __builtin_amdgcn_readfirstlane(tile_list[(ti + 1) * tstride].y) : -1;
        const int cur = ti & 1;
#ifdef R3D_TIMING
        long long *dbg = dbg_base && ti < 8 ? dbg_base + ti * 8 : nullptr;
#else
        long long *dbg = nullptr;
        (void)dbg;
        (void)dbg_base;
#endif
        R3D_TSTAMP(0);
        if (next_row0 >= 0) fill_rowtab(next_row0, cur ^ 1);   // (visible behind this tile's first barrier; read at its last tap)
        const float b1v = bias_at(P.bias2);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[mi][r] = b1v;
#pragma unroll 1
        for (int ts = 0; ts < 3; ++ts) {
            const int tap = tap_of(ts);
            // ---- this tap's expand_conv activations (E + V arrived during the previous matrix phase) -> H
            load_frag(w1rsrc, tap * tiles_per_tap, rb);
            load_frag(w1rsrc, __builtin_amdgcn_readfirstlane(tap * tiles_per_tap + (tiles_per_tap > 1 ? 1 : 0)), rbn);
            float *Ht = ts == 2 ? HR : H;                     // (the residual tap comes last)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                float *wr = Ht + (mi * 32 + 4 * lh) * PAIR_LD + wave * 32 + li;
#pragma unroll
                for (int r = 0; r < 16; ++r) wr[((r & 3) + 8 * (r >> 2)) * PAIR_LD] = lrelu(acc0[mi][r] + vt[mi][r], slope0);
            }
            __syncthreads();
            // ---- the next tap's values (the next tile's first tap behind the last one), in front of this tap's matrix work
            if (ts < 2) issue_e(cur, tap_of(ts + 1));
            else if (next_row0 >= 0) { issue_e(cur ^ 1, 0); issue_v(cur ^ 1); }
            // ---- this tap's third of the 3-tap convolution: K = C, barrier-free, weights two K tiles ahead
            {
                const float *h_frag = Ht + li * PAIR_LD + lh * 16;
                const int kbase = tap * tiles_per_tap, lastk = tiles_per_tap - 1;
                auto k_tile1 = [&](int kin, f32x4 (&w_use)[4], f32x4 (&w_load)[4]) {
                    load_frag(w1rsrc, kbase + (kin + 2 < lastk ? kin + 2 : lastk), w_load);
                    // (pinned, as in first_level_taps)
                    __builtin_amdgcn_sched_barrier(0);
                    const float *sp = h_frag + kin * BK;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        f32x4 av[MI];
#pragma unroll
                        for (int mi = 0; mi < MI; ++mi) av[mi] = *reinterpret_cast<const f32x4 *>(sp + mi * 32 * PAIR_LD + q * 4);
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                            for (int mi = 0; mi < MI; ++mi)
                                acc1[mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mi][kk], w_use[q][kk], acc1[mi], 0, 0, 0);
                    }
                };
                int kin = 0;
                for (; kin + 2 < tiles_per_tap; kin += 3) {
                    k_tile1(kin, rb, rbn2);
                    k_tile1(kin + 1, rbn, rb);
                    k_tile1(kin + 2, rbn2, rbn);
                }
                if (kin < tiles_per_tap) {
                    k_tile1(kin, rb, rbn2);
                    if (kin + 1 < tiles_per_tap) k_tile1(kin + 1, rbn, rb);
                }
            }
            if (ts < 2) __syncthreads();                      // every wavefront is done reading this tap's H (the last tap's is HR)
        }
        R3D_TSTAMP(1);
        // ---- level activations -> H; the 1x1 convolution on them
        load_frag(w2rsrc, 0, rb);
        load_frag(w2rsrc, nk2 > 1 ? 1 : 0, rbn);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            float *wr = H + (mi * 32 + 4 * lh) * PAIR_LD + wave * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) wr[((r & 3) + 8 * (r >> 2)) * PAIR_LD] = lrelu(acc1[mi][r], slope1);
        }
        __syncthreads();
        R3D_TSTAMP(2);
        const float b2v = bias_at(P.bias3);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[mi][r] = b2v;
        {
            const float *h_frag = H + li * PAIR_LD + lh * 16;
            const int last2 = nk2 - 1;
            auto k_tile2 = [&](int kt, f32x4 (&w_use)[4], f32x4 (&w_load)[4]) {
                load_frag(w2rsrc, kt + 2 < last2 ? kt + 2 : last2, w_load);
                // (pinned, as in first_level_taps)
                __builtin_amdgcn_sched_barrier(0);
                const float *sp = h_frag + kt * BK;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 av[MI];
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) av[mi] = *reinterpret_cast<const f32x4 *>(sp + mi * 32 * PAIR_LD + q * 4);
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                        for (int mi = 0; mi < MI; ++mi)
                            acc1[mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mi][kk], w_use[q][kk], acc1[mi], 0, 0, 0);
                }
            };
            int kt = 0;
            for (; kt + 2 < nk2; kt += 3) {
                k_tile2(kt, rb, rbn2);
                k_tile2(kt + 1, rbn, rb);
                k_tile2(kt + 2, rbn2, rbn);
            }
            if (kt < nk2) {
                k_tile2(kt, rb, rbn2);
                if (kt + 1 < nk2) k_tile2(kt + 1, rbn, rb);
            }
        }
        R3D_TSTAMP(3);
        // ---- epilogue: + the residual tap's activations (registers), rows transposed through H, 1 KiB stores
        __syncthreads();                                     // every wavefront is done reading H
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            float *wr = H + (mi * 32 + 4 * lh) * PAIR_LD + wave * 32 + li;
            const float *rr = HR + (mi * 32 + 4 * lh) * PAIR_LD + wave * 32 + li;        // (this lane's own values of the residual tap)
#pragma unroll
            for (int r = 0; r < 16; ++r) wr[((r & 3) + 8 * (r >> 2)) * PAIR_LD] = lrelu(acc1[mi][r], slope2) + rr[((r & 3) + 8 * (r >> 2)) * PAIR_LD];
        }
        __syncthreads();
        {
            const int rd_row = tid >> 6, rd_c4 = (tid & 63) * 4;
            const int N = P.N, ldc = P.ldc;
            const __amdgpu_buffer_rsrc_t crs = act_rsrc(P.c + (size_t)row0 * ldc);
#pragma unroll
            for (int j = 0; j < 4 * MI; ++j) {
                const int lr = rd_row + 8 * j, row = row0 + lr;
                if (row >= M) continue;
                const f32x4 v = *reinterpret_cast<const f32x4 *>(H + lr * PAIR_LD + rd_c4);
                if (rd_c4 + 4 <= N) {
                    act_store4(crs, (lr * ldc + rd_c4) * 4, v);
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (rd_c4 + c < N) act_store1(crs, (lr * ldc + rd_c4 + c) * 4, v[c]);
                }
            }
        }
        if (cnt) tile_drain();
        __syncthreads();                                     // H is free for the next tile's activations
        if (cnt) {
            const int4 te = tile_list[ti * tstride + 1];     // {dependencies (none), first ready counter, granules, -}
            tile_signal(cnt, __builtin_amdgcn_readfirstlane(te.y), __builtin_amdgcn_readfirstlane(te.z), MI);
        }
        R3D_TSTAMP(4);
    }
}


// ------------------------------------------------------------------------------------ first level on the bf16 matrix cores
//
// first_level_taps_b3: first_level_taps with every product evaluated by v_mfma_f32_32x32x16_bf16 on exact three-term
// bf16 splits of both operands (gemm_tile_b3 above has the arithmetic: six products per fp32 product, fp32 accumulate,
// the error of an fp32 dot product) - 6/16 of the fp32 MFMA's matrix time.  What changes around the matrix work:
//  * operands swap roles - the weights are the MFMA's A operand, the activations its B operand - so an accumulator
//    holds D[channel][row]: a lane owns ONE row and four consecutive channels per register quad, which is what lets
//    the activations go to LDS as packed bf16 (one ds_write_b64 per plane and quad) and the output rows as float4;
//  * the activations (G: the gathered chunk; H: a tap's / the level's activations) live in LDS as three bf16 planes,
//    split when they are written - each value once per tile - and are read as ready MFMA operands (b128 per plane);
//  * the weights stream as fp32 in bf16-MFMA operand order (GemmProb::wb3 / w2b3 / w3b3: the same bytes per K tile
//    as the fp32 path) and are split in registers by the wavefront that owns the 32 channels, behind the matrix work.
// Opt-in with the rest of the bf16x3 mode (R3D_BF16X3=1 at r3d_create).
constexpr int FLB_H_PITCH = B3T_H_PITCH;                                 // bf16 per H row: 528 B (conflict-free b128 operand reads)
constexpr int FLB_G_PITCH = 72;                                          // bf16 per G row: 144 B
constexpr int FLB_H_PLANE = FLT_MAX_MI * 32 * FLB_H_PITCH * 2;           // bytes per H plane: 33,792
constexpr int FLB_G_PLANE = FLT_MAX_MI * 32 * FLB_G_PITCH * 2;           // bytes per G plane:  9,216
constexpr int FLB_G_OFF = 3 * FLB_H_PLANE;                               // two G buffers of three planes each
constexpr int FLB_LUT_OFF = FLB_G_OFF + 6 * FLB_G_PLANE;                 // 156,672
static_assert(FLB_LUT_OFF + FL_LUT_INTS * 4 <= GEMM_LDS_BYTES, "the bf16x3 first level fits the GEMM kernel's LDS allocation");
static_assert(FLT_MAX_MI * 32 * PAIR_LD * 4 <= FLB_G_OFF, "the fp32 output rows are staged over the H planes");

template <int MI, bool MULTI, bool UV>
__device__ __forceinline__ void first_level_taps_b3(ProbRef P, const int4 *tile_list, const int tstride, const int ntiles, const bool new_prob, float *smem, const gu32 cnt,
                                                    long long *dbg_base) {
    static_assert(MI >= 1 && MI <= FLT_MAX_MI, "tile height");
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int w_voff = lane * 16;
    const int srow = tid >> 3, a_kq = (tid & 7) * 4;
    const int K0 = P.K, nk0 = K0 / BK, nch = (nk0 + 1) >> 1;
    const int M = P.M;
    const int res_tap = P.res_tap;
    char *lds = reinterpret_cast<char *>(smem);
    char *Hb = lds, *Gb = lds + FLB_G_OFF;
    int *lut_lds = reinterpret_cast<int *>(lds + FLB_LUT_OFF);
    const int *lut1 = lut_lds, *lutk = lut_lds + K0;
    if (new_prob) {
        for (int i = tid; i < K0 + K0 / 4; i += GEMM_THREADS) lut_lds[i] = *(const R3D_AS1 int *)(P.lut + i);
    }
    __syncthreads();
    __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.x), 0, P.enc_bytes, 0x00020000);
    typedef WFragB3 WFrag;
    auto load_w = [&](__amdgpu_buffer_rsrc_t rs, int kt, WFrag &dst) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                dst.f[h][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, w_voff + (h * 2 + j) * 1024, kt * 4096, 0));
    };
    const int nk1 = P.K2 / BK, tiles_per_tap = nk1 / 3, nk2 = P.K3 / BK;
    __amdgpu_buffer_rsrc_t w0rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(P.wb3 + ((size_t)wave_u * nk0) * 1024), 0, nk0 * 4096, 0x00020000);
    __amdgpu_buffer_rsrc_t w1rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(P.w2b3 + ((size_t)wave_u * nk1) * 1024), 0, nk1 * 4096, 0x00020000);
    __amdgpu_buffer_rsrc_t w2rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(P.w3b3 + ((size_t)wave_u * nk2) * 1024), 0, nk2 * 4096, 0x00020000);
    const float slope0 = P.slope, slope1 = P.slope2, slope2 = P.slope3;
    const int ch0 = wave * 32 + 4 * lh;                      // this lane's channels: ch0 + 8 q + e

    // ---- gather (as first_level_taps), committed as three bf16 planes
    struct Raw { f32x4 a[2]; };
    Raw gq;
    unsigned b_first, b_cur;
    const bool on = srow < MI * 32;
    CamRow camr;
    auto issue_phase = [&](int row0, int tap, int ch) {
        const int orow = row0 + srow;
        const int e = 3 * (orow < M ? orow : M - 1) + tap;
        const int win = e / P.enc_rows, t3 = e - win * P.enc_rows;
        const unsigned wbase = (unsigned)win * (unsigned)P.enc_ws;
        b_first = (wbase + (unsigned)(t3 * 3 * P.enc_jf)) * 4;
        b_cur = (wbase + (unsigned)P.enc_cur) * 4;
        if constexpr (UV) camr = load_cam_row(P.cam + (long long)win * P.cam_stride);
        if (!on) return;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int k = (ch * 2 + h) * BK + a_kq;
            if (MULTI && k >= K0) break;
            const int4 o1 = *reinterpret_cast<const int4 *>(lut1 + k);
            const unsigned b = lutk[k >> 2] != 0 ? b_cur : b_first;
            const int c1[4] = {o1.x & ~3, o1.y & ~3, o1.z & ~3, o1.w & ~3};
#pragma unroll
            for (int c = 0; c < 4; ++c)
                gq.a[h][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrsrc, b + (unsigned)c1[c], 0, 0));
        }
    };
    auto commit_phase = [&](int ch, char *G) {
        if (!on) return;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int k = (ch * 2 + h) * BK + a_kq;
            if (MULTI && k >= K0) break;
            f32x4 v = gq.a[h];
            if constexpr (UV) {
                const int4 code = *reinterpret_cast<const int4 *>(lut1 + k);
                v[0] = uv_to_ray(v[0], code.x, camr);
                v[1] = uv_to_ray(v[1], code.y, camr);
                v[2] = uv_to_ray(v[2], code.z, camr);
                v[3] = uv_to_ray(v[3], code.w, camr);
            }
            u32x2 pl[3];
            b3_split4(v, pl);
            char *d = G + (srow * FLB_G_PITCH + h * BK + a_kq) * 2;
#pragma unroll
            for (int p = 0; p < 3; ++p) *reinterpret_cast<u32x2 *>(d + p * FLB_G_PLANE) = pl[p];
        }
    };
    WFrag wa, wb, wc;                                        // streaming weight fragments (three sets rotating)
    // expand_conv fragments.  One chunk: resident in (w0a, w0b).  Several chunks: the streaming sets are idle during
    // the expand phases, so the chunks alternate between (wa, wb) and (wc, w0a) and chunk 0 returns to (wa, wb) once
    // a tap's 3-tap loop is done with them.
    WFrag w0a, w0b;
    if constexpr (MULTI) {
        load_w(w0rsrc, 0, wa);
        load_w(w0rsrc, nk0 > 1 ? 1 : 0, wb);
    } else {
        load_w(w0rsrc, 0, w0a);
        load_w(w0rsrc, nk0 > 1 ? 1 : 0, w0b);
    }
    int phase = 0;
    auto tap_of = [&](int ts) { return ts == 0 ? 0 : ts == 2 ? res_tap : 3 - res_tap; };
    issue_phase(__builtin_amdgcn_readfirstlane(tile_list[0].y), 0, 0);
#pragma unroll 1
    for (int ti = 0; ti < ntiles; ++ti) {
        const int row0 = __builtin_amdgcn_readfirstlane(tile_list[ti * tstride].y);
        const int next_row0 = ti + 1 < ntiles ? __builtin_amdgcn_readfirstlane(tile_list[(ti + 1) * tstride].y) : -1;
#ifdef R3D_TIMING
        long long *dbg = dbg_base && ti < 8 ? dbg_base + ti * 8 : nullptr;
#else
        long long *dbg = nullptr;
        (void)dbg;
#endif
        R3D_TSTAMP(0);
        f32x16 acc0[MI], acc1[MI];                           // (accumulators start at the layer's bias)
        b3t_init_bias<MI>(acc1, P.bias2, ch0);
#pragma unroll 1
        for (int ts = 0; ts < 3; ++ts) {
            const int tap = tap_of(ts);
            b3t_init_bias<MI>(acc0, P.bias, ch0);
            auto chunk = [&](int ch, const WFrag &ua, const WFrag &ub, WFrag &la, WFrag &lb) {
                char *G = Gb + (phase & 1) * 3 * FLB_G_PLANE;
                commit_phase(ch, G);
                __syncthreads();
                if constexpr (MULTI) {
                    if (ch + 1 < nch) {
                        load_w(w0rsrc, (ch + 1) * 2, la);
                        load_w(w0rsrc, (ch + 1) * 2 + 1 < nk0 ? (ch + 1) * 2 + 1 : (ch + 1) * 2, lb);
                    }
                }
                if (ch + 1 < nch) issue_phase(row0, tap, ch + 1);
                else if (ts < 2) issue_phase(row0, tap_of(ts + 1), 0);
                else if (next_row0 >= 0) issue_phase(next_row0, 0, 0);
                b3t_mma_ktile<MI>(G, FLB_G_PITCH, FLB_G_PLANE, ua, acc0, li, lh);
                if (ch * 2 + 1 < nk0) b3t_mma_ktile<MI>(G + BK * 2, FLB_G_PITCH, FLB_G_PLANE, ub, acc0, li, lh);
                ++phase;
            };
            if constexpr (MULTI) {
#pragma unroll 1
                for (int ch = 0; ch < nch; ch += 2) {
                    chunk(ch, wa, wb, wc, w0a);
                    if (ch + 1 < nch) chunk(ch + 1, wc, w0a, wa, wb);
                }
            } else {
                chunk(0, w0a, w0b, wc, wc);
            }
            if (ts == R3D_TS) R3D_TSTAMP(5);
            // ---- activations -> H planes (the residual tap's stay in acc0, fp32, for the epilogue)
            load_w(w1rsrc, tap * tiles_per_tap, wa);
            load_w(w1rsrc, __builtin_amdgcn_readfirstlane(tap * tiles_per_tap + (tiles_per_tap > 1 ? 1 : 0)), wb);
            b3t_activate_to_planes<MI>(acc0, slope0, Hb, FLB_H_PLANE, li, ch0);
            __syncthreads();
            if (ts == R3D_TS) R3D_TSTAMP(6);
            // ---- this tap's third of the 3-tap convolution
            {
                const int kbase = tap * tiles_per_tap, lastk = tiles_per_tap - 1;
                auto k_tile1 = [&](int kin, const WFrag &w_use, WFrag &w_load) {
                    load_w(w1rsrc, kbase + (kin + 2 < lastk ? kin + 2 : lastk), w_load);
                    b3t_mma_ktile<MI>(Hb + kin * BK * 2, FLB_H_PITCH, FLB_H_PLANE, w_use, acc1, li, lh);
                };
                int kin = 0;
                for (; kin + 2 < tiles_per_tap; kin += 3) {
                    k_tile1(kin, wa, wc);
                    k_tile1(kin + 1, wb, wa);
                    k_tile1(kin + 2, wc, wb);
                }
                if (kin < tiles_per_tap) {
                    k_tile1(kin, wa, wc);
                    if (kin + 1 < tiles_per_tap) k_tile1(kin + 1, wb, wa);
                }
            }
            if (ts == R3D_TS) R3D_TSTAMP(7);
            if constexpr (MULTI) {
                if (ts < 2) {
                    load_w(w0rsrc, 0, wa);
                    load_w(w0rsrc, nk0 > 1 ? 1 : 0, wb);
                }
            }
        }
        R3D_TSTAMP(1);
        // ---- level activations -> H planes; the 1x1 convolution on them
        load_w(w2rsrc, 0, wa);
        load_w(w2rsrc, nk2 > 1 ? 1 : 0, wb);
        __syncthreads();                                     // every wavefront is done reading the last tap's H
        b3t_activate_to_planes<MI>(acc1, slope1, Hb, FLB_H_PLANE, li, ch0);
        __syncthreads();
        R3D_TSTAMP(2);
        b3t_init_bias<MI>(acc1, P.bias3, ch0);
        {
            const int last2 = nk2 - 1;
            auto k_tile2 = [&](int kt, const WFrag &w_use, WFrag &w_load) {
                load_w(w2rsrc, kt + 2 < last2 ? kt + 2 : last2, w_load);
                b3t_mma_ktile<MI>(Hb + kt * BK * 2, FLB_H_PITCH, FLB_H_PLANE, w_use, acc1, li, lh);
            };
            int kt = 0;
            for (; kt + 2 < nk2; kt += 3) {
                k_tile2(kt, wa, wc);
                k_tile2(kt + 1, wb, wa);
                k_tile2(kt + 2, wc, wb);
            }
            if (kt < nk2) {
                k_tile2(kt, wa, wc);
                if (kt + 1 < nk2) k_tile2(kt + 1, wb, wa);
            }
        }
        R3D_TSTAMP(3);
        if constexpr (MULTI) {
            if (next_row0 >= 0) {
                load_w(w0rsrc, 0, wa);
                load_w(w0rsrc, nk0 > 1 ? 1 : 0, wb);
            }
        }
        // ---- epilogue: lrelu(acc + bias) + the residual tap's activations, fp32 rows staged over the H planes
        __syncthreads();                                     // every wavefront is done reading H
        {
            float *S = smem;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = lrelu(acc1[mi][4 * q + e], slope2) + acc0[mi][4 * q + e];
                    *reinterpret_cast<f32x4 *>(S + (mi * 32 + li) * PAIR_LD + ch0 + 8 * q) = v;
                }
            }
        }
        __syncthreads();
        {
            const float *S = smem;
            const int rd_row = tid >> 6, rd_c4 = (tid & 63) * 4;
            const int N = P.N, ldc = P.ldc;
            const __amdgpu_buffer_rsrc_t crs = act_rsrc(P.c + (size_t)row0 * ldc);
#pragma unroll
            for (int j = 0; j < 4 * MI; ++j) {
                const int lr = rd_row + 8 * j, row = row0 + lr;
                if (row >= M) continue;
                const f32x4 v = *reinterpret_cast<const f32x4 *>(S + lr * PAIR_LD + rd_c4);
                if (rd_c4 + 4 <= N) {
                    act_store4(crs, (lr * ldc + rd_c4) * 4, v);
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (rd_c4 + c < N) act_store1(crs, (lr * ldc + rd_c4 + c) * 4, v[c]);
                }
            }
        }
        if (cnt) tile_drain();
        __syncthreads();
        if (cnt) {
            const int4 te = tile_list[ti * tstride + 1];
            tile_signal(cnt, __builtin_amdgcn_readfirstlane(te.y), __builtin_amdgcn_readfirstlane(te.z), MI);
        }
        R3D_TSTAMP(4);
    }
}

#include "r3d_chain.hpp"

// ------------------------------------------------------------------------------------ calls of a few windows: GEMV tiles
//
// A problem of M <= 4 rows (the MLPs and the top of the conv pyramid in calls of up to four windows) has nothing for a
// 32-row MFMA tile to chew on: its time is the latency of a K loop - barriers, LDS round trips, a dependent MFMA chain -
// around a weight stream of a megabyte.  gemv_tile takes one 32-column block of such a problem and makes the stream the
// only thing that takes time: the eight wavefronts split K tile-wise (wavefront w: K tiles w, w + 8, ...), every
// wavefront requests ALL its weight fragments up front (the fragment order of the MFMA path: a lane owns one column and
// sixteen k of a K tile) - one memory round trip - while the workgroup copies the M operand rows into LDS; then plain FMAs
// (16 per row and K tile), the two k-halves of a wavefront and the eight wavefronts' partial sums added through LDS, bias /
// LeakyReLU / residual, one 128-byte row segment per row.  32 tiles per 1024-column layer: the five FuseBlocks' layers
// of a one-window call occupy 160 CUs instead of 80, each for a third of the time.
constexpr int GEMV_MAX_M = 4;
__device__ __forceinline__ void wait_deps(const int4 *tile, const int ndep, const gu32 cnt, const gu32 abort_flag, const long long spin_ticks);
// (single-launch form: these tiles wait for their producers themselves - BEHIND their weight requests, which depend on no
//  producer: in a call of a few windows a layer is one memory round trip, and the wait for the previous layer hides it)
struct TileDeps { const int4 *tile; int ndep; gu32 cnt, abort_flag; bool poll; long long *tstamp; long long spin_ticks; };   // (tstamp: -DR3D_TIMING builds, this tile's four stamps)
__device__ __forceinline__ float act_ld(const float *p) {
    return __builtin_bit_cast(float, __hip_atomic_load((gu32)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));   // 4-byte sc1 load
}
// Data as its own ready flag (calls of a few windows, FwdArgs::poll): the activation bank of the call was filled with
// ACT_SENTINEL - a quiet NaN no arithmetic produces - before the launch, GEMV / latency tiles read their operands until
// no sentinel is left in them instead of waiting for ready counters first, and their producers' write-through stores need
// no drain, counter update and counter poll in between: a dependency hop is one store -> load latency.  Every float is
// its own flag, so no ordering between stores is assumed.  (A producer that computes exactly this NaN - only from an
// input that carries it - stores the canonical quiet NaN instead.)
constexpr unsigned ACT_SENTINEL = 0x7fc5a1e7u;
__device__ __forceinline__ bool act_missing(float v) { return __builtin_bit_cast(unsigned, v) == ACT_SENTINEL; }
__device__ __forceinline__ void act_st(float *p, float v) {
    unsigned u = __builtin_bit_cast(unsigned, v);
    u = u == ACT_SENTINEL ? 0x7fc00000u : u;
    __hip_atomic_store((gu32)p, u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// one round of a bounded poll loop: true = give up (the launch's abort flag is up, or this wavefront has polled for
// `spin_ticks` of the 100 MHz wall clock - r3d_set_option(R3D_OPT_SPIN_TIMEOUT_MS), 1 s by default - and raises it): the
// caller goes on with what it has, the decoder turns the outputs into NaN and raises the handle's status (r3d_status)
__device__ __forceinline__ bool poll_gave_up(unsigned &spins, long long &t_first, const gu32 abort_flag, const long long spin_ticks) {
    __builtin_amdgcn_s_sleep(2);
    if ((++spins & 31) != 0) return false;
    if (__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return true;
    const long long now = wall_clock64();                       // 100 MHz
    if (t_first == 0) { t_first = now; return false; }
    if (now - t_first <= spin_ticks) return false;
    __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return true;
}
__device__ __forceinline__ void gemv_tile(ProbRef P, const int col0, float *smem, const TileDeps &dep) {
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int M = P.M, K = P.K, N = P.N, nk32 = K / BK;
    const int ldA = K + 8;
    float *As = smem, *red = smem + GEMV_MAX_M * ldA;
    __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(P.w + (size_t)(col0 >> 5) * nk32 * 1024), 0, nk32 * 4096, 0x00020000);   // (K tiles past the end read as zeros)
    f32x4 wf[4][4];
    auto load_round = [&](int j0) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                wf[j][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, lane * 16 + q * 1024, (wave_u + 8 * (j0 + j)) * 4096, 0));
    };
    load_round(0);
    const int er = tid >> 5, ecol = col0 + (tid & 31);
    const bool emit = tid < M * 32 && ecol < N;
    const float ebias = emit ? gload1(P.bias + ecol) : 0.0f;       // (the epilogue's bias: requested now, used after the reduction)
    if (dep.ndep > 0 && !dep.poll) wait_deps(dep.tile, dep.ndep, dep.cnt, dep.abort_flag, dep.spin_ticks);
    // the operand rows -> LDS (a virtual concatenation of up to MAX_SEG buffers; every column of such a problem is real)
    // (poll mode: read until no sentinel is left - each wavefront for its own elements, the barrier collects them)
    {
        const int e0 = P.kend[0], e1 = P.kend[1], e2 = P.kend[2];
        const float *a0 = P.a[0], *a1 = P.a[1], *a2 = P.a[2], *a3 = P.a[3];
        const int l0 = P.lda[0], l1 = P.lda[1], l2 = P.lda[2], l3 = P.lda[3];
        unsigned spins = 0;
        long long t_first = 0;
        for (;;) {
            bool missing = false;
            for (int r = 0; r < M; ++r)
                for (int k = tid; k < K; k += GEMM_THREADS) {
                    const float *src = k < e0 ? a0 + (size_t)r * l0 + k : k < e1 ? a1 + (size_t)r * l1 + (k - e0)
                                     : k < e2 ? a2 + (size_t)r * l2 + (k - e1) : a3 + (size_t)r * l3 + (k - e2);
                    const float v = act_ld(src);
                    missing |= act_missing(v);
                    As[r * ldA + k] = v;
                }
            if (!dep.poll || !__any(missing) || poll_gave_up(spins, t_first, dep.abort_flag, dep.spin_ticks)) break;
        }
    }
    // (the residual is an input of the operand's producer chain: whoever sees the operand sees it - polled all the same)
    float eres = 0.0f;
    if (emit && P.res) {
        unsigned spins = 0;
        long long t_first = 0;
        do eres = act_ld(P.res + (size_t)er * P.ldr + ecol);
        while (dep.poll && act_missing(eres) && !poll_gave_up(spins, t_first, dep.abort_flag, dep.spin_ticks));
    }
    __syncthreads();
    float acc[GEMV_MAX_M];
#pragma unroll
    for (int r = 0; r < GEMV_MAX_M; ++r) acc[r] = 0.0f;
    for (int j0 = 0; wave_u + 8 * j0 < nk32; j0 += 4) {
        if (j0) load_round(j0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int kt = wave_u + 8 * (j0 + j);
            if (kt >= nk32) break;                                  // (uniform)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int r = 0; r < GEMV_MAX_M; ++r) {
                    const f32x4 a = *reinterpret_cast<const f32x4 *>(As + (r < M ? r : M - 1) * ldA + kt * BK + lh * 16 + q * 4);
                    acc[r] += a[0] * wf[j][q][0] + a[1] * wf[j][q][1] + a[2] * wf[j][q][2] + a[3] * wf[j][q][3];
                }
        }
    }
#pragma unroll
    for (int r = 0; r < GEMV_MAX_M; ++r) {
        acc[r] += __shfl_xor(acc[r], 32, 64);                       // the two k-halves of the wavefront
        if (lh == 0) red[(wave * GEMV_MAX_M + r) * 32 + li] = acc[r];
    }
    __syncthreads();
    if (emit) {
        float v = 0.0f;
#pragma unroll
        for (int w = 0; w < 8; ++w) v += red[(w * GEMV_MAX_M + er) * 32 + (tid & 31)];
        v = lrelu(v + ebias, P.slope) + eres;
        act_st(P.c + (size_t)er * P.ldc + ecol, v);
    }
    __syncthreads();                                                // the next tile may write LDS
}

// lat_tile: the same shape - one 32-column block of a layer, K split tile-wise over the eight wavefronts, every operand of a
// wavefront's share requested up front (weights: fragment order; activations: straight from memory into MFMA operand
// registers, no LDS ring, no per-K-tile barrier), partial sums added through LDS - for layers of 5 .. 32 rows, on the
// fp32 matrix cores.  A 1024-deep layer is four K tiles per wavefront: one memory round trip and 64 MFMAs where the
// split-K gemm_tile runs eight barrier-separated iterations (12.5 us per M = B stage of a 16-window call against ~7).
__device__ __forceinline__ void lat_tile(ProbRef P, const int col0, float *smem, const TileDeps &dep) {
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int M = P.M, K = P.K, N = P.N, nk32 = K / BK;
    __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(P.w + (size_t)(col0 >> 5) * nk32 * 1024), 0, nk32 * 4096, 0x00020000);
    const int e0 = P.kend[0], e1 = P.kend[1], e2 = P.kend[2];
    const int arow = li < M ? li : M - 1;                       // (rows past the problem re-read its last row: never stored)
    f32x4 wf[4][4], af[4][4];
    auto load_w_round = [&](int j0) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                wf[j][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, lane * 16 + q * 1024, (wave_u + 8 * (j0 + j)) * 4096, 0));
    };
    auto load_a_round = [&](int j0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int kt = wave_u + 8 * (j0 + j), k = kt * BK;
            // the K tile's segment of the (virtual) concatenation: uniform
            const int sg = k < e0 ? 0 : k < e1 ? 1 : k < e2 ? 2 : 3;
            const int k0 = sg == 0 ? 0 : sg == 1 ? e0 : sg == 2 ? e1 : e2;
            const float *base = sg == 0 ? P.a[0] : sg == 1 ? P.a[1] : sg == 2 ? P.a[2] : P.a[3];
            const int ld = sg == 0 ? P.lda[0] : sg == 1 ? P.lda[1] : sg == 2 ? P.lda[2] : P.lda[3];
            const __amdgpu_buffer_rsrc_t ars = act_rsrc(base);
            const bool live = kt < nk32;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                af[j][q] = live ? act_load4(ars, (arow * ld + (k - k0) + lh * 16 + q * 4) * 4) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        }
    };
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    load_w_round(0);
    // the epilogue's bias: requested now, used after the reduction (this thread's two outputs: r3d of the two passes below)
    float ebias[2], eres[2] = {0.0f, 0.0f};
    int erow[2], ecol[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int idx = tid + h * GEMM_THREADS, r = idx >> 6, ln = idx & 63;
        erow[h] = (r & 3) + 8 * (r >> 2) + 4 * (ln >> 5);
        ecol[h] = col0 + (ln & 31);
        ebias[h] = erow[h] < M && ecol[h] < N ? gload1(P.bias + ecol[h]) : 0.0f;
    }
    if (dep.ndep > 0 && !dep.poll) wait_deps(dep.tile, dep.ndep, dep.cnt, dep.abort_flag, dep.spin_ticks);
    // the residual: requested by EVERY emitting thread, ahead of the K loop - a wavefront whose share of a short K is empty
    // (K < 256: wave_u >= nk32) never enters the loop and still emits rows (poll mode: checked again in the epilogue)
    if (P.res) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
            if (erow[h] < M && ecol[h] < N) eres[h] = act_ld(P.res + (size_t)erow[h] * P.ldr + ecol[h]);
    }
    unsigned spins = 0;
    long long t_first = 0;
    for (int j0 = 0; wave_u + 8 * j0 < nk32; j0 += 4) {
        if (j0) load_w_round(j0);
        for (;;) {                                                     // (poll mode: until the round holds no sentinel)
            load_a_round(j0);
            if (!dep.poll) break;
            bool missing = false;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) missing |= act_missing(af[j][q][kk]);
            if (!__any(missing) || poll_gave_up(spins, t_first, dep.abort_flag, dep.spin_ticks)) break;
        }
#ifdef R3D_TIMING
        if (j0 == 0 && dep.tstamp && threadIdx.x == 0) dep.tstamp[1] = wall_clock64();      // (this wavefront's first operand round has arrived)
#endif
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (wave_u + 8 * (j0 + j) >= nk32) break;               // (uniform; a 256-deep layer is ONE K tile per wavefront, not four - three
                                                                    //  quarters of that tile's matrix time went into zeros)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j][q][kk], wf[j][q][kk], acc, 0, 0, 0);
        }
    }
    float *red = smem;                                             // [wave][register][lane]
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[r];
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int idx = tid + h * GEMM_THREADS, r = idx >> 6, ln = idx & 63;
        float v = 0.0f;
#pragma unroll
        for (int w = 0; w < 8; ++w) v += red[(w * 16 + r) * 64 + ln];
        const int row = erow[h], col = ecol[h];
        if (row < M && col < N) {
            v = lrelu(v + ebias[h], P.slope);
            if (P.res) {
                float rv = eres[h];
                unsigned rspins = 0;
                long long rt = 0;
                while (dep.poll && act_missing(rv) && !poll_gave_up(rspins, rt, dep.abort_flag, dep.spin_ticks)) rv = act_ld(P.res + (size_t)row * P.ldr + col);
                v += rv;
            }
            act_st(P.c + (size_t)row * P.ldc + col, v);
        }
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------ tile-level dependencies
//
// r3d_forward_f32 runs the tiles of EVERY level of the network in one launch.  What orders them is data: a tile's
// descriptor lists, per producer problem, the range of 32-row units its windows need (the network is row-local) and how
// many 64-column granules each of them must have finished; one wavefront polls those ready counters - one counter per
// lane, relaxed agent-scope loads, s_sleep between polls - and a barrier releases the workgroup.  No acquire fence
// follows: producers store activations write-through (sc1) and consumers load them with sc1 (ACT_AUX above).
// Spins are bounded: after ~1 s without progress the wavefront raises the launch's abort flag and goes on; every later
// wait sees the flag and returns at once, the decoder kernel turns the outputs into NaN, nothing hangs.
typedef const FwdArgs __attribute__((address_space(4))) *FwdArgsPtr;
// wait_deps on a descriptor a wavefront already holds (lane i = descriptor int i: ONE load for the whole descriptor in front of the tile
// instead of dependent round trips for its header, its dependency list and then the counters)
__device__ __forceinline__ void wait_deps_reg(const int dw, const int ndep, const gu32 cnt, const gu32 abort_flag, const long long spin_ticks) {
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        int total = 0;
        for (int d = 0; d < ndep; ++d) total += __builtin_amdgcn_readlane(dw, 9 + 2 * d) & 0xffff;
        for (int off = 0; off < total; off += 64) {
            int idx = -1, acc = 0;
            unsigned need = 0;
            for (int d = 0; d < ndep; ++d) {
                const int base = __builtin_amdgcn_readlane(dw, 8 + 2 * d), nw = __builtin_amdgcn_readlane(dw, 9 + 2 * d);
                const int n = nw & 0xffff, l = lane + off - acc;
                if (l >= 0 && l < n) { idx = base + l; need = (unsigned)nw >> 16; }
                acc += n;
            }
            long long t_first = 0;
            for (unsigned spins = 1;; ++spins) {
                const unsigned v = idx >= 0 ? __hip_atomic_load(cnt + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : need;
                if (__all(v >= need)) break;
                __builtin_amdgcn_s_sleep(4);
                if ((spins & 31) == 0) {
                    if (__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
                    const long long now = wall_clock64();
                    if (t_first == 0) t_first = now;
                    else if (now - t_first > spin_ticks) {
                        if (lane == 0) __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                }
            }
        }
    }
    __syncthreads();
}
__device__ __forceinline__ void wait_deps(const int4 *tile, const int ndep, const gu32 cnt, const gu32 abort_flag, const long long spin_ticks) {
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        const int *ti = reinterpret_cast<const int *>(tile);
        int total = 0;
        for (int d = 0; d < ndep; ++d) total += __builtin_amdgcn_readfirstlane(ti[9 + 2 * d]) & 0xffff;
        for (int off = 0; off < total; off += 64) {          // (one pass unless a tile needs more than 64 counters)
            int idx = -1, acc = 0;
            unsigned need = 0;
            for (int d = 0; d < ndep; ++d) {
                const int base = __builtin_amdgcn_readfirstlane(ti[8 + 2 * d]), nw = __builtin_amdgcn_readfirstlane(ti[9 + 2 * d]);
                const int n = nw & 0xffff, l = lane + off - acc;
                if (l >= 0 && l < n) { idx = base + l; need = (unsigned)nw >> 16; }
                acc += n;
            }
            long long t_first = 0;
            for (unsigned spins = 1;; ++spins) {
                const unsigned v = idx >= 0 ? __hip_atomic_load(cnt + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : need;
                if (__all(v >= need)) break;
                __builtin_amdgcn_s_sleep(4);
                if ((spins & 31) == 0) {
                    if (__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
                    const long long now = wall_clock64();                       // 100 MHz
                    if (t_first == 0) t_first = now;
                    else if (now - t_first > spin_ticks) {
                        if (lane == 0) __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                }
            }
        }
    }
    __syncthreads();
}

// A workgroup's consecutive GEMV tiles as one run (single-launch form; calls of up to four windows are little else): the
// tile loop of gemv_tile with the NEXT tile's first round of weights requested while this tile computes.  Every layer's
// weights are read once per call, i.e. from HBM, and a CU that has a tile in every layer of the chain would otherwise
// start that 2-3 us round trip only when its previous tile is done - longer than the tile itself.  The request goes out
// behind this tile's operand loads (memory returns in order: in front of them it would delay them) and ahead of its
// arithmetic.  Tiles whose consumers all take data as its own flag (FWD_TILE_NOSIGNAL, poll mode) skip the drain and the
// counter update.
constexpr int FWD_TILE_NOSIGNAL = 1;      // descriptor int 7, bit 0 (r3d_schedule.cpp)
__device__ __forceinline__ void gemv_run(FwdArgsPtr fargs, const int4 *tl, const int TS, const int n, float *smem, const gu32 cnt,
                                         const gu32 abort_flag, int &gemv_seen, long long *dbg_arg, const int t_first_tile) {
    const long long spin_ticks = fargs->spin_ticks;
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const bool poll = fargs->poll != 0;
    typedef const GemmProb __attribute__((address_space(4))) *ProbPtr;
    auto prob_of = [&](int i) -> ProbPtr {
        return (ProbPtr)fargs->probs + (__builtin_amdgcn_readfirstlane(tl[i * TS].x) & 0xff);
    };
    auto request = [&](ProbPtr P, int col0, int j0, f32x4 (&wf)[4][4]) {
        const int nk32 = P->K / BK;
        __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(P->w + (size_t)(col0 >> 5) * nk32 * 1024), 0, nk32 * 4096, 0x00020000);   // (K tiles past the end read as zeros)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                wf[j][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, lane * 16 + q * 1024, (wave_u + 8 * (j0 + j)) * 4096, 0));
    };
    f32x4 wa[4][4], wb[4][4];
    request(prob_of(0), __builtin_amdgcn_readfirstlane(tl[0].z), 0, wa);
    auto body = [&](const int i, f32x4 (&wf)[4][4], f32x4 (&wnext)[4][4]) {
        const int4 *tile = tl + i * TS;
        ProbPtr Pp = prob_of(i);
        ProbRef P = *Pp;
        const int col0 = __builtin_amdgcn_readfirstlane(tile[0].z);
        const int4 te = tile[1];
        const int ndep = __builtin_amdgcn_readfirstlane(te.x), sig_base = __builtin_amdgcn_readfirstlane(te.y);
        const int sig_add = __builtin_amdgcn_readfirstlane(te.z), flags = __builtin_amdgcn_readfirstlane(te.w);
#ifdef R3D_TIMING
        if (dbg_arg && threadIdx.x == 0) dbg_arg[16384 + (long long)(t_first_tile + i) * 4 + 0] = wall_clock64();
#endif
        const int M = P.M, K = P.K, N = P.N, nk32 = K / BK;
        const int ldA = K + 8;
        float *As = smem, *red = smem + GEMV_MAX_M * ldA;
        const int er = tid >> 5, ecol = col0 + (tid & 31);
        const bool emit = tid < M * 32 && ecol < N;
        const float ebias = emit ? gload1(P.bias + ecol) : 0.0f;
        if (ndep > 0 && !poll) wait_deps(tile, ndep, cnt, abort_flag, spin_ticks);
        {
            const int e0 = P.kend[0], e1 = P.kend[1], e2 = P.kend[2];
            const float *a0 = P.a[0], *a1 = P.a[1], *a2 = P.a[2], *a3 = P.a[3];
            const int l0 = P.lda[0], l1 = P.lda[1], l2 = P.lda[2], l3 = P.lda[3];
            unsigned spins = 0;
            long long t_first = 0;
            for (;;) {
                bool missing = false;
                for (int r = 0; r < M; ++r)
                    for (int k = tid; k < K; k += GEMM_THREADS) {
                        const float *src = k < e0 ? a0 + (size_t)r * l0 + k : k < e1 ? a1 + (size_t)r * l1 + (k - e0)
                                         : k < e2 ? a2 + (size_t)r * l2 + (k - e1) : a3 + (size_t)r * l3 + (k - e2);
                        const float v = act_ld(src);
                        missing |= act_missing(v);
                        As[r * ldA + k] = v;
                    }
                if (!poll || !__any(missing) || poll_gave_up(spins, t_first, abort_flag, spin_ticks)) break;
            }
        }
        float eres = 0.0f;
        if (emit && P.res) {
            unsigned spins = 0;
            long long t_first = 0;
            do eres = act_ld(P.res + (size_t)er * P.ldr + ecol);
            while (poll && act_missing(eres) && !poll_gave_up(spins, t_first, abort_flag, spin_ticks));
        }
        if (i + 1 < n) request(prob_of(i + 1), __builtin_amdgcn_readfirstlane(tl[(i + 1) * TS].z), 0, wnext);
        __syncthreads();
#ifdef R3D_TIMING
        if (dbg_arg && threadIdx.x == 0) dbg_arg[16384 + (long long)(t_first_tile + i) * 4 + 1] = wall_clock64();
#endif
        float acc[GEMV_MAX_M];
#pragma unroll
        for (int r = 0; r < GEMV_MAX_M; ++r) acc[r] = 0.0f;
        for (int j0 = 0; wave_u + 8 * j0 < nk32; j0 += 4) {
            if (j0) request(Pp, col0, j0, wf);
            // (row by row, and only the rows there are: one window reads and multiplies a quarter of what four do - 0.098 against
            //  0.108 ms at one window, 0.116 against 0.124 at two, 0.147 against 0.143 at four; two accumulator chains per row: the same)
#pragma unroll
            for (int r = 0; r < GEMV_MAX_M; ++r) {
                if (r >= M) break;                                      // (uniform)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int kt = wave_u + 8 * (j0 + j);
                    if (kt >= nk32) break;                              // (uniform)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 a = *reinterpret_cast<const f32x4 *>(As + r * ldA + kt * BK + lh * 16 + q * 4);
                        acc[r] += a[0] * wf[j][q][0] + a[1] * wf[j][q][1] + a[2] * wf[j][q][2] + a[3] * wf[j][q][3];
                    }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < GEMV_MAX_M; ++r) {
            acc[r] += __shfl_xor(acc[r], 32, 64);                       // the two k-halves of the wavefront
            if (lh == 0) red[(wave * GEMV_MAX_M + r) * 32 + li] = acc[r];
        }
        __syncthreads();
        // (test hook: workgroup 0's n-th tile neither stores nor reports - what the bounded spins are for)
        const bool faulty = blockIdx.x == 0 && gemv_seen++ == fargs->fault_tile1 - 1;
        if (emit && !faulty) {
            float v = 0.0f;
#pragma unroll
            for (int w = 0; w < 8; ++w) v += red[(w * GEMV_MAX_M + er) * 32 + (tid & 31)];
            v = lrelu(v + ebias, P.slope) + eres;
            act_st(P.c + (size_t)er * P.ldc + ecol, v);
        }
        if (!(poll && (flags & FWD_TILE_NOSIGNAL))) {
            tile_drain();
            __syncthreads();
            if (!faulty) tile_signal(cnt, sig_base, sig_add, 1);
        } else {
            __syncthreads();                                            // the next tile may write LDS
        }
#ifdef R3D_TIMING
        if (dbg_arg && threadIdx.x == 0) {
            dbg_arg[16384 + (long long)(t_first_tile + i) * 4 + 2] = wall_clock64();
            dbg_arg[16384 + (long long)(t_first_tile + i) * 4 + 3] = 1;
        }
#endif
    };
    int i = 0;
    for (; i + 1 < n; i += 2) {
        body(i, wa, wb);
        body(i + 1, wb, wa);
    }
    if (i < n) body(i, wa, wb);
}

// B3: the kernel carries the tiles that run fp32 GEMMs on the bf16 matrix cores (r3d_config.bf16x3); NARROW: the GEMV /
// latency tiles of calls of a few windows (and their data-as-its-own-flag hand-off).  The single-launch forward exists in
// three specialisations - r3d_forward_f32 (neither: the fp32 throughput tiles only), r3d_forward_b3, r3d_forward_lat - picked
// on the host by what the schedule's tile lists hold, so that the headline kernel pays neither registers nor scratch for
// code it never runs and a trace names the mode.
// CHAIN: the kernel carries the register-chained first-level tile (r3d_chain.hpp; GemmProb::wchain) - the fp32 kernels of gathered-rays calls.
template <bool ENC, bool UV, bool DEP = false, bool B3 = true, bool NARROW = true, bool CLIP = !DEP, bool CHAIN = false>
__device__ __forceinline__ void gemm_persistent(float *smem) {
    LaunchArgsPtr args = (LaunchArgsPtr)__builtin_amdgcn_kernarg_segment_ptr();
    FwdArgsPtr fargs = (FwdArgsPtr)__builtin_amdgcn_kernarg_segment_ptr();      // (DEP: the same segment holds a FwdArgs)
    constexpr int TS = DEP ? FWD_TILE_INT4 : 1;                                  // int4s per tile descriptor
    // XCD-aware chunk order: workgroup b runs on XCD b % 8 (observed; speed only), so give each XCD a
    // contiguous run of chunks - neighbouring chunks share weights (and A rows) through that XCD's L2.
    // (Single-launch form: the host has applied that order per level when it concatenated the workgroups' lists.)
    int wg = blockIdx.x;
    if constexpr (!DEP) {
        const int n = gridDim.x, q = n >> 3, r = n & 7, xcd = wg & 7, idx = wg >> 3;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int4 *tiles = DEP ? fargs->tiles : args->tiles;
    const int *wg_off = DEP ? fargs->wg_off : args->wg_off;
    const gu32 cnt = DEP ? (gu32)fargs->cnt : (gu32) nullptr;
    const gu32 abort_flag = DEP ? (gu32)(fargs->cnt + fargs->ncnt) : (gu32) nullptr;
    const int t0 = __builtin_amdgcn_readfirstlane(wg_off[wg]);
    const int t1 = __builtin_amdgcn_readfirstlane(wg_off[wg + 1]);
    if constexpr (DEP) {
        // the other bank of ready counters (and its abort flag): zero for the next call, which then runs without r3d_bind_f32
        // (nothing reads that bank before this launch has ended)
        unsigned *nx = fargs->cnt_next;
        if (nx != nullptr)
            for (int j = blockIdx.x * GEMM_THREADS + threadIdx.x; j < fargs->ncnt + 4; j += gridDim.x * GEMM_THREADS) nx[j] = 0u;
        // ... and the other bank of activations (poll mode): sentinels
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        u32x4 *arm = NARROW ? reinterpret_cast<u32x4 *>(fargs->arm) : nullptr;
        if (arm != nullptr) {
            const u32x4 sv = {ACT_SENTINEL, ACT_SENTINEL, ACT_SENTINEL, ACT_SENTINEL};
            for (long long j = blockIdx.x * GEMM_THREADS + threadIdx.x; j < fargs->arm_vec4; j += gridDim.x * GEMM_THREADS) arm[j] = sv;
        }
    }
    // The clock this launch ran at: workgroup 0 stamps its shader-cycle counter and the 100 MHz wall clock at both ends and
    // leaves the two differences behind the abort flag (words ncnt + 2, ncnt + 3 of its counter bank: r3d_last_clock).
    // (the start stamps wait in those two words, not in registers: four values live across the persistent loop spilled)
    if (DEP && blockIdx.x == 0 && threadIdx.x == 0) {
        fargs->cnt[fargs->ncnt + 2] = (unsigned)__builtin_readcyclecounter();
        fargs->cnt[fargs->ncnt + 3] = (unsigned)wall_clock64();
    }
    long long *dbg = nullptr;
#ifdef R3D_TIMING
    long long *dbg_arg = DEP ? fargs->dbg : args->dbg;
    long long *dbg_base = dbg_arg && wg < 16 ? dbg_arg + 6144 + wg * 64 : nullptr;   // 8 tiles x 8 stamps
    if (dbg_arg && threadIdx.x == 0) {
        dbg_arg[1024 + wg * 4 + 0] = __builtin_readcyclecounter();
        dbg_arg[1024 + wg * 4 + 2] = wall_clock64();
    }
#endif
    int prev_pi = -1;
    int plain_seen = 0, gemv_seen = 0;       // (test hook: this workgroup's tiles so far, per kind)
    for (int t = t0; t < t1; ++t) {
        // the whole descriptor (24 ints) in ONE load per wavefront, lane i = int i: header, signal fields and dependency list come out of it
        // with v_readlane - no second and third round trip before the counters can be asked for (-0.2 .. 0.3 % of the forward, six A/B
        // rounds: profiles/r06_desc_one_load/)
        int dw = 0;
        if constexpr (DEP) {
            int dl = threadIdx.x & 63;
            asm volatile("" : "+v"(dl));
            dw = reinterpret_cast<const int *>(tiles + t * TS)[dl < FWD_TILE_INT4 * 4 ? dl : 0];
        }
        int4 td;
        if constexpr (DEP) td = make_int4(__builtin_amdgcn_readlane(dw, 0), __builtin_amdgcn_readlane(dw, 1), __builtin_amdgcn_readlane(dw, 2), __builtin_amdgcn_readlane(dw, 3));
        else td = tiles[t * TS];
        const int pi = __builtin_amdgcn_readfirstlane(td.x & 0xff);
        const bool new_prob = pi != prev_pi;
        prev_pi = pi;
        const int mi = __builtin_amdgcn_readfirstlane(td.x >> 8);
        const int row0 = __builtin_amdgcn_readfirstlane(td.y);
        const int col0 = __builtin_amdgcn_readfirstlane(td.z);
        const int ks = __builtin_amdgcn_readfirstlane(td.w);      // split-K factor of this tile (1, 2 or 4)
        int sig_base = 0, sig_add = 0, tflags = 0;
        LateWait late{0, 0, nullptr, nullptr, 0};
        TileDeps tdep{nullptr, 0, nullptr, nullptr, false, nullptr, 0};
        if constexpr (DEP) {
            const int4 te = make_int4(__builtin_amdgcn_readlane(dw, 4), __builtin_amdgcn_readlane(dw, 5), __builtin_amdgcn_readlane(dw, 6), __builtin_amdgcn_readlane(dw, 7));
            const int ndep = __builtin_amdgcn_readfirstlane(te.x);
            sig_base = __builtin_amdgcn_readfirstlane(te.y);
            sig_add = __builtin_amdgcn_readfirstlane(te.z);
            tflags = __builtin_amdgcn_readfirstlane(te.w);
#ifdef R3D_TIMING
            if (dbg_arg && threadIdx.x == 0) dbg_arg[16384 + (long long)t * 4 + 0] = wall_clock64();      // tile fetched
#endif
            // (GEMV / latency tiles wait themselves, behind their weight requests)
#ifndef R3D_TIMING
            // plain / split-K / pair / narrow-column tiles wait inside, behind their first weight requests: a dependency hop is on the
            // critical path of a 256-window call thirteen times, and the weights' round trip now overlaps the counters' (-1.7 % at 256
            // windows, -0.2 % at 1024, six same-box rounds: profiles/r06_late_wait/).  First-level and gathered tiles have no producers;
            // the bf16x3 tiles and the timing build - whose "producers ready" stamp is taken here - keep the wait in front.
            constexpr bool LATE = !B3 && !ENC && !CHAIN;      // (the chained-tile experiment's kernel as it was measured: its register budget has no room)
#else
            constexpr bool LATE = false;
#endif
            const bool wait_here = ndep > 0 && (!NARROW || !tile_is_narrow(ks));
            if (wait_here && !LATE) wait_deps_reg(dw, ndep, cnt, abort_flag, fargs->spin_ticks);
            if (wait_here && LATE) late = LateWait{dw, ndep, cnt, abort_flag, fargs->spin_ticks};
#ifdef R3D_TIMING
            if (dbg_arg && threadIdx.x == 0) dbg_arg[16384 + (long long)t * 4 + 1] = wall_clock64();      // producers ready
#endif
            tdep = TileDeps{tiles + t * TS, NARROW && tile_is_narrow(ks) ? ndep : 0, cnt, abort_flag, NARROW && fargs->poll != 0, nullptr, fargs->spin_ticks};
#ifdef R3D_TIMING
            if (dbg_arg) tdep.tstamp = dbg_arg + 16384 + (long long)t * 4;
#endif
        }
        ProbRef P = DEP ? *((const GemmProb __attribute__((address_space(4))) *)fargs->probs + pi) : args->p[pi];
#ifdef R3D_TIMING
        dbg = dbg_base && t - t0 < 8 ? dbg_base + (t - t0) * 8 : nullptr;
#endif
        bool signalled = false;
        do {
        if constexpr (ENC) {
            (void)ks;
            switch (mi) {
                case 1: enc_tile<1, UV>(P, row0, col0, new_prob, smem, dbg); break;
                case 2: enc_tile<2, UV>(P, row0, col0, new_prob, smem, dbg); break;
                default: enc_tile<3, UV>(P, row0, col0, new_prob, smem, dbg); break;
            }
        } else {
            if (P.w3 != nullptr) {       // first level of the pyramid, fused (32 output rows per tile): this
                int n = 1;               // workgroup's consecutive tiles of the problem as one run
                while (t + n < t1 && __builtin_amdgcn_readfirstlane(tiles[(t + n) * TS].x) == __builtin_amdgcn_readfirstlane(td.x)) ++n;   // same problem, same height
#ifdef R3D_TIMING
                long long *run_dbg = dbg_base && t - t0 < 8 ? dbg_base + (t - t0) * 8 : nullptr;
#else
                long long *run_dbg = nullptr;
#endif
                const int4 *tl = tiles + t * TS;
                if (CLIP && P.lut == nullptr) {   // a clip call: expand_conv's pre-activations come from the per-frame buffer
                  if constexpr (CLIP) {
                    if (mi >= 2) first_level_shared<2>(P, tl, TS, n, smem, cnt, run_dbg);
                    else first_level_shared<1>(P, tl, TS, n, smem, cnt, run_dbg);
                  }
                } else if (B3 && P.wb3 != nullptr) {   // fp32 on the bf16 matrix cores
                  if constexpr (B3) {
                    if (P.K <= 64) {
                        if (mi >= 2) first_level_taps_b3<2, false, UV>(P, tl, TS, n, new_prob, smem, cnt, run_dbg);
                        else first_level_taps_b3<1, false, UV>(P, tl, TS, n, new_prob, smem, cnt, run_dbg);
                    } else {
                        if (mi >= 2) first_level_taps_b3<2, true, UV>(P, tl, TS, n, new_prob, smem, cnt, run_dbg);
                        else first_level_taps_b3<1, true, UV>(P, tl, TS, n, new_prob, smem, cnt, run_dbg);
                    }
                  }
                } else if (CHAIN && P.wchain != nullptr && mi >= 2) {   // 64-row tiles of a body-part branch: register-chained
                    if constexpr (CHAIN) first_level_chain<4>(P, tl, TS, n, new_prob, smem, cnt, run_dbg);
                } else if (P.K <= 64) {
                    if (mi >= 2) first_level_taps<2, false, UV>(P, tl, TS, n, new_prob, smem, cnt, run_dbg);
                    else first_level_taps<1, false, UV>(P, tl, TS, n, new_prob, smem, cnt, run_dbg);
                } else {
                    if (mi >= 2) first_level_taps<2, true, UV>(P, tl, TS, n, new_prob, smem, cnt, run_dbg);
                    else first_level_taps<1, true, UV>(P, tl, TS, n, new_prob, smem, cnt, run_dbg);
                }
#ifdef R3D_TIMING
                if (DEP && dbg_arg && threadIdx.x == 0) {                    // (a run: the stamps of its first tile stand for all n)
                    dbg_arg[16384 + (long long)t * 4 + 2] = wall_clock64();
                    dbg_arg[16384 + (long long)t * 4 + 3] = n;
                }
#endif
                t += n - 1;
                signalled = true;        // (every tile of the run has raised its own counters)
                break;
            }
            if (P.lut != nullptr) {      // gathered operand without the fused level: GlobalInfo.fc_1's current frames
                switch (mi) {
                    case 1: enc_tile<1, UV>(P, row0, col0, new_prob, smem, dbg); break;
                    case 2: enc_tile<2, UV>(P, row0, col0, new_prob, smem, dbg); break;
                    default: enc_tile<3, UV>(P, row0, col0, new_prob, smem, dbg); break;
                }
                break;
            }
            if constexpr (B3) {
            if (P.wb3 != nullptr && P.w2 != nullptr) {   // a fused pair on the bf16 matrix cores (tiles of <= 96 rows)
                if (mi >= 3) gemm_tile_b3t<3>(P, row0, smem, dbg);
                else if (mi == 2) gemm_tile_b3t<2>(P, row0, smem, dbg);
                else gemm_tile_b3t<1>(P, row0, smem, dbg);
                break;
            }
            if (P.wb3 != nullptr) {      // fp32 on the bf16 matrix cores (whole tiles of <= 128 rows)
                switch (mi) {
                    case 1: gemm_tile_b3p<1>(P, row0, col0, smem, dbg); break;   // (32.9 against 34.3 us for the M = B launch; two-unit
                    case 2: gemm_tile_b3<2>(P, row0, col0, smem, dbg); break;    //  tiles are 2 % slower pipelined: 53.8 against 52.8)
                    case 3: gemm_tile_b3<3>(P, row0, col0, smem, dbg); break;
                    default: gemm_tile_b3<4>(P, row0, col0, smem, dbg); break;
                }
                break;
            }
            }
            if constexpr (NARROW) {
            if (ks == 8) {               // a problem of a few rows: one 32-column block, K split over the wavefronts, no MFMA
                if constexpr (DEP) {     // ... this workgroup's consecutive tiles of the kind as one run (weights requested a tile ahead)
                    int n = 1;
                    while (t + n < t1 && __builtin_amdgcn_readfirstlane(tiles[(t + n) * TS].w) == 8) ++n;
#ifdef R3D_TIMING
                    long long *run_dbg_arg = dbg_arg;
#else
                    long long *run_dbg_arg = nullptr;
#endif
                    gemv_run(fargs, tiles + t * TS, TS, n, smem, cnt, abort_flag, gemv_seen, run_dbg_arg, t);
                    t += n - 1;
                    signalled = true;
                    break;
                }
                gemv_tile(P, col0, smem, tdep);
                break;
            }
            if (ks == 16) {              // ... of up to 32 rows: the same shape on the matrix cores
                lat_tile(P, col0, smem, tdep);
                break;
            }
            }
            if (ks > NB_CODE) {          // a single-unit tile of 4 .. 7 column blocks
                if (ks == NB_CODE + 4) gemm_tile_nb<4>(P, row0, col0, smem, dbg, late);
                else if (ks == NB_CODE + 5) gemm_tile_nb<5>(P, row0, col0, smem, dbg, late);
                else if (ks == NB_CODE + 6) gemm_tile_nb<6>(P, row0, col0, smem, dbg, late);
                else gemm_tile_nb<7>(P, row0, col0, smem, dbg, late);
                break;
            }
            if (ks > 1) {
                if (ks == 4) gemm_tile<1, 4>(P, row0, col0, smem, dbg, late);
                else if (mi == 1) gemm_tile<1, 2>(P, row0, col0, smem, dbg, late);
                else gemm_tile<2, 2>(P, row0, col0, smem, dbg, late);
                break;
            }
            if (P.w2 != nullptr) {       // fused pair (the scheduler caps these tiles at PAIR_MAX_MI units)
                switch (mi) {
                    case 1: gemm_tile<1, 1, true>(P, row0, col0, smem, dbg, late); break;
                    case 2: gemm_tile<2, 1, true>(P, row0, col0, smem, dbg, late); break;
                    case 3: gemm_tile<3, 1, true>(P, row0, col0, smem, dbg, late); break;
                    default: gemm_tile<4, 1, true>(P, row0, col0, smem, dbg, late); break;
                }
                break;
            }
            switch (mi) {
                case 1: gemm_tile<1, 1>(P, row0, col0, smem, dbg, late); break;
                case 2: gemm_tile<2, 1>(P, row0, col0, smem, dbg, late); break;
                case 3: gemm_tile<3, 1>(P, row0, col0, smem, dbg, late); break;
                case 4: gemm_tile<4, 1>(P, row0, col0, smem, dbg, late); break;
                case 5: gemm_tile<5, 1>(P, row0, col0, smem, dbg, late); break;
                default: gemm_tile<6, 1>(P, row0, col0, smem, dbg, late); break;
            }
        }
        } while (false);
        if constexpr (DEP) {
            if (NARROW && !signalled && fargs->poll != 0 && (tflags & FWD_TILE_NOSIGNAL)) {   // (a latency tile nobody counts on)
                signalled = true;
#ifdef R3D_TIMING
                if (dbg_arg && threadIdx.x == 0) {
                    dbg_arg[16384 + (long long)t * 4 + 2] = wall_clock64();
                    dbg_arg[16384 + (long long)t * 4 + 3] = 1;
                }
#endif
            }
            if (!signalled) {            // (the tile functions end on a barrier: drain, one more barrier, raise the counters)
                tile_drain();
                __syncthreads();
                // (test hook: workgroup 0's n-th tile of this kind never reports - what the bounded spins are for)
                if (!(blockIdx.x == 0 && plain_seen++ == fargs->fault_tile1 - 1)) tile_signal(cnt, sig_base, sig_add, mi);
#ifdef R3D_TIMING
                if (dbg_arg && threadIdx.x == 0) {
                    dbg_arg[16384 + (long long)t * 4 + 2] = wall_clock64();                               // tile finished
                    dbg_arg[16384 + (long long)t * 4 + 3] = 1;
                }
#endif
            }
        }
    }
    if (DEP && blockIdx.x == 0 && threadIdx.x == 0) {
        volatile unsigned *ck = fargs->cnt + fargs->ncnt + 2;
        const unsigned c0 = ck[0], w0 = ck[1];
        ck[0] = (unsigned)__builtin_readcyclecounter() - c0;      // (differences of the low words: a forward is far below 2^32 cycles)
        ck[1] = (unsigned)wall_clock64() - w0;
    }
#ifdef R3D_TIMING
    if (dbg_arg && threadIdx.x == 0) {
        dbg_arg[1024 + wg * 4 + 1] = __builtin_readcyclecounter();
        dbg_arg[1024 + wg * 4 + 3] = wall_clock64();
    }
#endif
}

}  // namespace r3d
