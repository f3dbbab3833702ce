// gfx950 (MI355X, CDNA4): the throughput form of the lifting forward pass with TWO independent workgroups per CU.
//
// r3d_kernels.hip runs one 512-thread workgroup per CU: eight wavefronts in lock step, so a tile's prologue, its LDS
// phases, its epilogue and its wait for producers leave the CU's four matrix pipes idle (tiles are ~80 % matrix-busy
// inside).  Here a workgroup is 256 threads - four wavefronts, one per SIMD - with at most 80 KB of LDS and
// 256 VGPRs, so that two of them share a CU: whatever one of them does that is not an MFMA, the other one's MFMAs fill.
// A wavefront owns MI row blocks x NB column blocks of 32 x 32 (column blocks wave, wave + 4), i.e. MI * NB accumulators:
//
//   g4_tile<MI, 1>        plain Conv1d / Linear (rie.py:94-97 un-fused, :122-135, :159-169; embedding.py:15-18), eval
//                         BatchNorm folded: 32 MI rows x 128 columns, MI <= 4 - an M = B layer of 256 windows is 64 x 128
//                         tiles whose every weight fragment feeds two row blocks (a 32 x 256 tile of eight wavefronts:
//                         one), 224 of them for 256 CUs as before;
//   g4_tile<MI, 2>        the same, 32 MI rows x 256 columns (MI <= 2): large batches;
//   g4_tile<MI, 2, PAIR>  a pyramid level's 3-tap and 1x1 convolutions (rie.py:94-97), the intermediate tile in LDS;
//   g4_first              expand_conv on the gathered input + the first pyramid level, tap by tap (rie.py:85-97;
//                         window gather of lib/train_val/trainer.py:47-58; UV mode: lib/camera/camera.py:423-471 in the
//                         gather), 32 output rows x 256 columns;
//   g4_enc                GlobalInfo.fc_1 on the gathered current frames (rie.py:290-292, :362).
// Tiles, problem tables, ready counters and the hand-off protocol are those of r3d_kernels.hip (r3d_device.hpp); the host
// schedules 2 x CUs workgroups (r3d_schedule.cpp, four-wave mode).  fp32 MFMA (v_mfma_f32_32x32x2_f32) only.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "r3d_device.hpp"
#include "r3d_internal.hpp"

namespace r3d {

constexpr int W4_THREADS = 256;
constexpr int W4_LDS_BYTES = 80 * 1024;              // two workgroups per CU (160 KB)
constexpr int W4_PAIR_LD = 256 + 4;                  // floats per row of an intermediate tile (conflict-free b128 rows)
constexpr int W4_G_LD = 64 + 4;                      // a 64-column chunk of the gathered operand
constexpr int W4_H_FLOATS = 32 * W4_PAIR_LD;         // first level: H (32 rows)
constexpr int W4_G_FLOATS = 32 * W4_G_LD;            // ... G, two buffers
constexpr int W4_LUT_OFF = W4_H_FLOATS + 2 * W4_G_FLOATS;
static_assert((W4_LUT_OFF + FL_LUT_INTS) * 4 <= W4_LDS_BYTES, "first level fits");
static_assert(64 * W4_PAIR_LD * 4 <= W4_LDS_BYTES, "a 64-row pair tile fits");
static_assert(3 * 4 * 32 * LDS_LD * 4 <= W4_LDS_BYTES, "a 128-row ring fits");

// Epilogue: C = res + lrelu(acc + bias).  The accumulators (C/D layout of v_mfma_f32_32x32x2_f32: col = lane & 31, row =
// (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)) go through LDS one 32-row slab at a time, so that every lane moves 16 bytes
// and a wavefront instruction covers whole row segments - for the residual read too.
template <int MI, int NB>
__device__ __forceinline__ void store_tile4(ProbRef P, const f32x16 (&acc)[MI][NB], const int row0, const int col0, float *lds,
                                            const bool second = false) {
    constexpr int COLS = 128 * NB, ELD = COLS + 4;
    constexpr int TPR = COLS / 4;                 // threads per output row (16 bytes each)
    constexpr int RPP = W4_THREADS / TPR;         // rows per pass: 8 / 4
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int M = P.M, N = P.N;
    const float slope = second ? P.slope2 : P.slope;
    const float *res = P.res;
    const int ldc = P.ldc, ldr = P.ldr;
    const __amdgpu_buffer_rsrc_t crs = act_rsrc(P.c + (size_t)row0 * ldc + col0);
    const __amdgpu_buffer_rsrc_t rrs = act_rsrc(res ? res + (size_t)row0 * ldr + col0 : P.c);
    float bias[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) bias[j] = gload1((second ? P.bias2 : P.bias) + col0 + (wave + 4 * j) * 32 + li);
    float *wr = lds + (4 * lh) * ELD + wave * 32 + li;
    const int rd_row = tid / TPR, rd_c4 = (tid % TPR) * 4;
    const bool vec = (col0 + COLS <= N);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        __syncthreads();                                            // LDS free: K loop / previous slab done
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) wr[((r & 3) + 8 * (r >> 2)) * ELD + j * 128] = lrelu(acc[mi][j][r] + bias[j], slope);
        __syncthreads();
#pragma unroll
        for (int jj = 0; jj < 32 / RPP; ++jj) {
            const int lr = rd_row + RPP * jj;
            const int row = row0 + mi * 32 + lr;
            if (row >= M) continue;
            f32x4 v = *reinterpret_cast<const f32x4 *>(lds + lr * ELD + rd_c4);
            const int col = col0 + rd_c4;
            const int lrow = mi * 32 + lr;
            if (vec) {
                if (res) v += act_load4(rrs, (lrow * ldr + rd_c4) * 4);
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

// One tile of C = res + lrelu(A W^T + b): 32 MI rows x 128 NB columns.  A (activations; possibly a virtual concatenation
// of up to MAX_SEG buffers, rie.py:371-407) goes global -> VGPR -> LDS ring of three stages through a buffer descriptor
// with a scalar K offset, requested five K tiles ahead and committed two ahead of its use; W never touches LDS: fragment-
// ordered weights (r3d_model.cpp, frag_index) straight into VGPRs two K tiles ahead, three register sets rotating by name.
// One barrier per K tile.  PAIR (NB == 2, N <= 256): the tile of the first layer stays in LDS in MFMA operand order and
// the 1x1 convolution runs on it at once - barrier-free, weights streaming - before the one epilogue with the residual.
template <int MI, int NB, bool PAIR = false>
__device__ __forceinline__ void g4_tile(ProbRef P, const int row0, const int col0, float *smem) {
    static_assert(MI >= 1 && MI * NB <= 4 && (NB == 1 || NB == 2), "a wavefront holds at most four accumulators");
    static_assert(!PAIR || NB == 2, "a fused pair spans all 256 columns");
    constexpr int SF = MI * 32 * LDS_LD;    // floats per ring stage
    constexpr int NA = MI;                  // A staging slots per thread (32 staged rows per slot)
    constexpr bool PRE = MI <= 2;           // the next tile's first A fragments are read before the barrier
    constexpr int AD = 5;                   // A tile t+AD is issued in iteration t, committed to LDS in iteration t+AD-2
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));           // (opaque: per-thread constants stay out of the persistent loop's preheader)
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int M = P.M, K = P.K;
    const int nk = K / BK;
    const int srow = tid >> 3, a_kq = (tid & 7) * 4;

    // ---- A staging (the segment state only ever advances; no scalar loads in the steady-state loop)
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
            const int gr = row0 + srow + 32 * i;
            a_voff[i] = (((gr < M ? gr : M - 1) - row0) * seg_ld + a_kq) * 4;
        }
    };
    open_seg();
    auto prep_seg = [&](int kt) {
        if (!multi) return;
        while (kt * BK >= seg_end) {        // uniform
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
    const int st_off = srow * LDS_LD + a_kq;
    auto commit_a = [&](int stage, const Staged &R) {
        float *s = stage == 0 ? smem + st_off : stage == 1 ? smem + SF + st_off : smem + 2 * SF + st_off;
#pragma unroll
        for (int i = 0; i < NA; ++i) *reinterpret_cast<f32x4 *>(s + i * 32 * LDS_LD) = R.a[i];
    };

    // ---- W fragments: [(n/32)][k tile][q][lane][4]; this wavefront's column blocks wave, wave + 4
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(P.w + ((size_t)((col0 >> 5) + wave_u) * nk) * 1024), 0, ((NB - 1) * 4 + 1) * nk * 4096, 0x00020000);
    const int w_voff = lane * 16;
    const int w_blk4 = 4 * nk * 4096;       // bytes between column blocks b and b + 4
    typedef f32x4 WSet[NB][4];
    WSet rb, rbn, rbn2;
    auto load_w = [&](int kt, WSet &dst) {
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                dst[j][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, w_voff + q * 1024, kt * 4096 + j * w_blk4, 0));
    };

    f32x16 acc[MI][NB];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][j][r] = 0.0f;

    const int a_frag = li * LDS_LD + lh * 16;
    f32x4 av0[PRE ? MI : 1];

    // ---- prologue: tiles 0 and 1 into LDS, tiles 2..4 in flight, W(0), W(1) requested
    const int last = nk - 1;
    load_w(0, rb);
    load_w(1 < last ? 1 : last, rbn);
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
    if (PRE) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) av0[mi] = *reinterpret_cast<const f32x4 *>(smem + a_frag + mi * 32 * LDS_LD);
    }
    int st_cur = 0;
    auto k_tile = [&](int kt, WSet &w_use, WSet &w_load, Staged &stg) {
        const int st_next = st_cur == 2 ? 0 : st_cur + 1, st_next2 = st_next == 2 ? 0 : st_next + 1;
        const float *s = smem + st_cur * SF + a_frag;
        auto mfma_q = [&](int q) {
            f32x4 av[MI];
            if constexpr (PRE) {
                // a wavefront that is alone on its SIMD waits out every LDS round trip it starts just in time: the A fragments of
                // quad q + 1 are requested BEFORE quad q's matrix work (av0 carries quad 0 across the barrier)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) av[mi] = av0[mi];
                // (quad 3: the NEXT K tile's quad 0 - its stage was committed a barrier ago - so that nothing is in flight at the barrier)
                const float *nx = q < 3 ? s + (q + 1) * 4 : smem + st_next * SF + a_frag;
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) av0[mi] = *reinterpret_cast<const f32x4 *>(nx + mi * 32 * LDS_LD);
                __builtin_amdgcn_sched_barrier(0);
            } else {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) av[mi] = *reinterpret_cast<const f32x4 *>(s + mi * 32 * LDS_LD + q * 4);
            }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int j = 0; j < NB; ++j)
                        acc[mi][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mi][kk], w_use[j][q][kk], acc[mi][j], 0, 0, 0);
        };
        // staging behind the first MFMAs when there are accumulators to interleave with, in front otherwise
        constexpr bool INTERLEAVE = MI * NB >= 2;
        if (!INTERLEAVE) {
            commit_a(st_next2, stg);
            load_w(kt + 2 < last ? kt + 2 : last, w_load);
            issue_a(kt + AD < last ? kt + AD : last, stg);
            prep_seg(kt + AD + 1 < last ? kt + AD + 1 : last);
        }
        mfma_q(0);
        if (INTERLEAVE) {
            commit_a(st_next2, stg);
            load_w(kt + 2 < last ? kt + 2 : last, w_load);
        }
        mfma_q(1);
        if (INTERLEAVE) {
            issue_a(kt + AD < last ? kt + AD : last, stg);
            prep_seg(kt + AD + 1 < last ? kt + AD + 1 : last);
        }
        mfma_q(2);
        mfma_q(3);
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

    if constexpr (PAIR) {
        // ---- first layer's activations -> LDS (the staging ring is dead), as the A operand of the second
        const float slope1 = P.slope;
        float bias1[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) bias1[j] = gload1(P.bias + (wave + 4 * j) * 32 + li);
        const int nk2 = P.K2 / BK, last2 = nk2 - 1;
        __amdgpu_buffer_rsrc_t w2rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(P.w2 + ((size_t)wave_u * nk2) * 1024), 0, 5 * nk2 * 4096, 0x00020000);
        const int w2_blk4 = 4 * nk2 * 4096;
        auto load_w2 = [&](int k2, WSet &dst) {
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    dst[j][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w2rsrc, w_voff + q * 1024, k2 * 4096 + j * w2_blk4, 0));
        };
        load_w2(0, rb);
        load_w2(1 < last2 ? 1 : last2, rbn);
        // (the last k_tile ended with a barrier: every wavefront is done with the ring)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            float *wr = smem + (mi * 32 + 4 * lh) * W4_PAIR_LD + wave * 32 + li;
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) wr[((r & 3) + 8 * (r >> 2)) * W4_PAIR_LD + j * 128] = lrelu(acc[mi][j][r] + bias1[j], slope1);
        }
        __syncthreads();
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][j][r] = 0.0f;
        const float *h_frag = smem + li * W4_PAIR_LD + lh * 16;
        auto k_tile2 = [&](int k2, WSet &w_use, WSet &w_load) {
            load_w2(k2 + 2 < last2 ? k2 + 2 : last2, w_load);
            const float *s = h_frag + k2 * BK;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 av[MI];
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) av[mi] = *reinterpret_cast<const f32x4 *>(s + mi * 32 * W4_PAIR_LD + q * 4);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                        for (int j = 0; j < NB; ++j)
                            acc[mi][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mi][kk], w_use[j][q][kk], acc[mi][j], 0, 0, 0);
            }
        };
        int k2 = 0;
        for (; k2 + 2 < nk2; k2 += 3) {
            k_tile2(k2, rb, rbn2);
            k_tile2(k2 + 1, rbn, rb);
            k_tile2(k2 + 2, rbn2, rbn);
        }
        if (k2 < nk2) {
            k_tile2(k2, rb, rbn2);
            if (k2 + 1 < nk2) k_tile2(k2 + 1, rbn, rb);
        }
        store_tile4<MI, NB>(P, acc, row0, col0, smem, true);        // (begins with a barrier: the tile is dead)
    } else {
        store_tile4<MI, NB>(P, acc, row0, col0, smem);
    }
}

// GlobalInfo.fc_1 on the gathered current frames (and any other gathered operand that is not a fused first level): the
// tile's WHOLE encoded operand (32 MI rows x K <= 256, one gathered input element per column: the differences of the
// reference's encoding live in the folded weights) is built in LDS first, then a barrier-free MFMA loop consumes it.
template <int MI, bool UV>
__device__ __forceinline__ void g4_enc(ProbRef P, const int row0, const int col0, const bool new_prob, float *smem) {
    constexpr int R = MI * 32, NA = MI;
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int M = P.M, K = P.K;
    const int nk = K / BK;
    const int ldt = K + 4;
    const int srow = tid >> 3, a_kq = (tid & 7) * 4;
    int *lut_lds = reinterpret_cast<int *>(smem + (W4_LDS_BYTES / 4 - FL_LUT_INTS));
    __syncthreads();
    const int *lut1 = lut_lds, *lutk = lut_lds + K;
    if (new_prob) {
        for (int i = tid; i < K + K / 4; i += W4_THREADS) lut_lds[i] = *(const R3D_AS1 int *)(P.lut + i);
        __syncthreads();
    }
    __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.x), 0, P.enc_bytes, 0x00020000);
    unsigned b_first[NA], b_cur[NA];
    CamRow camr[UV ? NA : 1];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int gr = row0 + srow + 32 * i;
        const int row = gr < M ? gr : M - 1;
        const int win = row / P.enc_rows, t3 = row - win * P.enc_rows;
        const unsigned wbase = (unsigned)win * (unsigned)P.enc_ws;
        b_first[i] = (wbase + (unsigned)(t3 * P.enc_step * P.enc_jf)) * 4;
        b_cur[i] = (wbase + (unsigned)P.enc_cur) * 4;
        if constexpr (UV) camr[i] = load_cam_row(P.cam + (long long)win * P.cam_stride);
    }
    struct Raw { f32x4 a[NA]; };
    auto issue = [&](int kt, Raw &r) {
        const int k = kt * BK + a_kq;
        const int4 o1 = *reinterpret_cast<const int4 *>(lut1 + k);
        const bool cur = lutk[k >> 2] != 0;
        const int c1[4] = {o1.x & ~3, o1.y & ~3, o1.z & ~3, o1.w & ~3};
#pragma unroll
        for (int i = 0; i < NA; ++i) {
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
            f32x4 v = r.a[i];
            if constexpr (UV) {
                v[0] = uv_to_ray(v[0], code.x, camr[i]);
                v[1] = uv_to_ray(v[1], code.y, camr[i]);
                v[2] = uv_to_ray(v[2], code.z, camr[i]);
                v[3] = uv_to_ray(v[3], code.w, camr[i]);
            }
            *reinterpret_cast<f32x4 *>(smem + (srow + 32 * i) * ldt + kt * BK + a_kq) = v;
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
    f32x16 acc[MI][1];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][0][r] = 0.0f;
    const float *a_frag = smem + li * ldt + lh * 16;
    const int last = nk - 1;
    auto k_tile = [&](int kt, f32x4 (&w_use)[4], f32x4 (&w_load)[4]) {
        load_w(kt + 1 < last ? kt + 1 : last, w_load);
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
                    acc[mi][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mi][kk], w_use[q][kk], acc[mi][0], 0, 0, 0);
        }
    };
    load_w(0, rb);
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {
        k_tile(kt, rb, rbn);
        k_tile(kt + 1, rbn, rb);
    }
    if (kt < nk) k_tile(kt, rb, rbn);
    store_tile4<MI, 1>(P, acc, row0, col0, smem);
}

// The first pyramid level, tap by tap (r3d_kernels.hip, first_level_taps, has the derivation): expand_conv on the gathered
// rows {3r + tap} -> activations H (32 rows x C, MFMA operand order in LDS) -> that tap's third of the 3-tap convolution
// into the level's accumulators; the residual tap last, its activations staying in the expand accumulators; then the 1x1
// convolution on the level's activations and the epilogue.  32 output rows x all C <= 256 columns per tile; a wavefront
// owns column blocks wave and wave + 4.  Weights: expand_conv's two K tiles resident per wavefront when K0 <= 64 (the
// body-part branches), otherwise (MULTI: the trajectory model's K0 = 224) streamed K tile by K tile through the two
// streaming sets, which also carry the 3-tap and 1x1 layers one K tile ahead.
template <bool MULTI, bool UV>
__device__ __forceinline__ void g4_first(ProbRef P, const int4 *tile_list, const int tstride, const int ntiles, const bool new_prob, float *smem,
                                         const gu32 cnt) {
    constexpr int NB = 2;
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int w_voff = lane * 16;
    const int srow = tid >> 3, a_kq = (tid & 7) * 4;         // staging: 32 rows x 8 threads x 4 columns per K tile
    const int K0 = P.K, nk0 = K0 / BK, nch = (nk0 + 1) >> 1;
    const int M = P.M;
    const int res_tap = P.res_tap;
    float *H = smem, *G0 = smem + W4_H_FLOATS;
    int *lut_lds = reinterpret_cast<int *>(smem + W4_LUT_OFF);
    const int *lut1 = lut_lds, *lutk = lut_lds + K0;
    if (new_prob) {
        for (int i = tid; i < K0 + K0 / 4; i += W4_THREADS) lut_lds[i] = *(const R3D_AS1 int *)(P.lut + i);
    }
    __syncthreads();                                         // (also: the previous tile is done with LDS)
    __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.x), 0, P.enc_bytes, 0x00020000);
    typedef f32x4 WSet[NB][4];
    const int nk1 = P.K2 / BK, tiles_per_tap = nk1 / 3, nk2 = P.K3 / BK;
    __amdgpu_buffer_rsrc_t w0rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(P.w + ((size_t)wave_u * nk0) * 1024), 0, 5 * nk0 * 4096, 0x00020000);
    __amdgpu_buffer_rsrc_t w1rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(P.w2 + ((size_t)wave_u * nk1) * 1024), 0, 5 * nk1 * 4096, 0x00020000);
    __amdgpu_buffer_rsrc_t w2rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(P.w3 + ((size_t)wave_u * nk2) * 1024), 0, 5 * nk2 * 4096, 0x00020000);
    auto load_set = [&](__amdgpu_buffer_rsrc_t rs, int kt, int nk_layer, WSet &dst) {
        const int so = __builtin_amdgcn_readfirstlane(kt * 4096);          // (uniform by construction; keeps waterfall loops away)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                dst[j][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, w_voff + q * 1024, so + j * 4 * nk_layer * 4096, 0));
    };
    const float slope0 = P.slope, slope1 = P.slope2, slope2 = P.slope3;
    auto bias_at = [&](const float *b, int j) {
        int c = (wave + 4 * j) * 32 + li;
        asm volatile("" : "+v"(c));
        return gload1(b + c);
    };

    // ---- gather state of the phase whose raw values are in flight / in registers
    struct Raw { f32x4 a[2]; };                              // the two K tiles of a 64-column chunk
    Raw gq;
    unsigned b_first, b_cur;
    CamRow camr;
    auto issue_phase = [&](int row0, int tap, int ch) {      // tile rows [row0, row0 + 32), expand_conv rows 3r + tap
        const int orow = row0 + srow;
        const int e = 3 * (orow < M ? orow : M - 1) + tap;
        const int win = e / P.enc_rows, t3 = e - win * P.enc_rows;
        const unsigned wbase = (unsigned)win * (unsigned)P.enc_ws;
        b_first = (wbase + (unsigned)(t3 * 3 * P.enc_jf)) * 4;
        b_cur = (wbase + (unsigned)P.enc_cur) * 4;
        if constexpr (UV) camr = load_cam_row(P.cam + (long long)win * P.cam_stride);
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
            *reinterpret_cast<f32x4 *>(G + srow * W4_G_LD + h * BK + a_kq) = v;
        }
    };

    WSet sa, sb;                                             // streaming weight fragments (two sets, one K tile ahead)
    WSet w0a, w0b;                                           // !MULTI: expand_conv's two K tiles, resident for the whole run
    if constexpr (MULTI) {
        load_set(w0rsrc, 0, nk0, sa);
        load_set(w0rsrc, nk0 > 1 ? 1 : 0, nk0, sb);
    } else {
        load_set(w0rsrc, 0, nk0, w0a);
        load_set(w0rsrc, nk0 > 1 ? 1 : 0, nk0, w0b);
    }
    int phase = 0;                                           // parity selects the G buffer
    auto tap_of = [&](int ts) { return ts == 0 ? 0 : ts == 2 ? res_tap : 3 - res_tap; };   // the residual tap comes last
    auto mma_ktile = [&](const float *a_rows, const int pitch, const WSet &w, f32x16 (&acc)[NB]) {      // one 32-deep K tile from LDS rows
        const float *sp = a_rows + li * pitch + lh * 16;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 av = *reinterpret_cast<const f32x4 *>(sp + q * 4);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int j = 0; j < NB; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk], w[j][q][kk], acc[j], 0, 0, 0);
        }
    };
    issue_phase(__builtin_amdgcn_readfirstlane(tile_list[0].y), 0, 0);
#pragma unroll 1
    for (int ti = 0; ti < ntiles; ++ti) {
        const int row0 = __builtin_amdgcn_readfirstlane(tile_list[ti * tstride].y);
        const int next_row0 = ti + 1 < ntiles ? __builtin_amdgcn_readfirstlane(tile_list[(ti + 1) * tstride].y) : -1;
        f32x16 acc0[NB], acc1[NB];                           // (accumulators start at the layer's bias)
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const float b1v = bias_at(P.bias2, j);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[j][r] = b1v;
        }
#pragma unroll 1
        for (int ts = 0; ts < 3; ++ts) {
            const int tap = tap_of(ts);
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const float b0v = bias_at(P.bias, j);
#pragma unroll
                for (int r = 0; r < 16; ++r) acc0[j][r] = b0v;
            }
            // ---- expand_conv on the rows {3r + tap}, one 64-column chunk of the operand per phase
#pragma unroll 1
            for (int ch = 0; ch < nch; ++ch) {
                float *G = G0 + (phase & 1) * W4_G_FLOATS;
                commit_phase(ch, G);
                __syncthreads();
                // the next phase's raw values, in front of this phase's matrix work
                if (ch + 1 < nch) issue_phase(row0, tap, ch + 1);
                else if (ts < 2) issue_phase(row0, tap_of(ts + 1), 0);
                else if (next_row0 >= 0) issue_phase(next_row0, 0, 0);
                const bool two = ch * 2 + 1 < nk0;
                if constexpr (MULTI) {
                    // K tiles 2 ch (in sa) and 2 ch + 1 (in sb); the next chunk's follow one K tile behind their use
                    mma_ktile(G, W4_G_LD, sa, acc0);
                    if (ch + 1 < nch) load_set(w0rsrc, 2 * ch + 2, nk0, sa);
                    if (two) mma_ktile(G + BK, W4_G_LD, sb, acc0);
                    if (ch + 1 < nch && 2 * ch + 3 < nk0) load_set(w0rsrc, 2 * ch + 3, nk0, sb);
                } else {
                    mma_ktile(G, W4_G_LD, w0a, acc0);
                    if (two) mma_ktile(G + BK, W4_G_LD, w0b, acc0);
                }
                ++phase;
            }
            // ---- activations (in place: the residual tap's stay in acc0 for the epilogue) -> H
            load_set(w1rsrc, tap * tiles_per_tap, nk1, sa);
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                float *wr = H + (4 * lh) * W4_PAIR_LD + (wave + 4 * j) * 32 + li;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = lrelu(acc0[j][r], slope0);
                    acc0[j][r] = v;
                    wr[((r & 3) + 8 * (r >> 2)) * W4_PAIR_LD] = v;
                }
            }
            __syncthreads();
            // ---- this tap's third of the 3-tap convolution: K = C, barrier-free, weights one K tile ahead
            {
                const int kbase = tap * tiles_per_tap, lastk = tiles_per_tap - 1;
                auto k_tile1 = [&](int kin, const WSet &w_use, WSet &w_load) {
                    load_set(w1rsrc, kbase + (kin + 1 < lastk ? kin + 1 : lastk), nk1, w_load);
                    mma_ktile(H + kin * BK, W4_PAIR_LD, w_use, acc1);
                };
                int kin = 0;
                for (; kin + 1 < tiles_per_tap; kin += 2) {
                    k_tile1(kin, sa, sb);
                    k_tile1(kin + 1, sb, sa);
                }
                if (kin < tiles_per_tap) k_tile1(kin, sa, sb);
            }
            // (no barrier here: the next tap's first chunk phase has one between this loop and the next write of H)
            if constexpr (MULTI) {
                if (ts < 2) {                                // chunk 0 again for the next tap (the streaming sets are dead here)
                    load_set(w0rsrc, 0, nk0, sa);
                    load_set(w0rsrc, nk0 > 1 ? 1 : 0, nk0, sb);
                }
            }
        }
        // ---- level activations -> H; the 1x1 convolution on them
        load_set(w2rsrc, 0, nk2, sa);
        __syncthreads();                                     // every wavefront is done reading the last tap's H
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            float *wr = H + (4 * lh) * W4_PAIR_LD + (wave + 4 * j) * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) wr[((r & 3) + 8 * (r >> 2)) * W4_PAIR_LD] = lrelu(acc1[j][r], slope1);
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const float b2v = bias_at(P.bias3, j);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[j][r] = b2v;
        }
        {
            const int last2 = nk2 - 1;
            auto k_tile2 = [&](int kt, const WSet &w_use, WSet &w_load) {
                load_set(w2rsrc, kt + 1 < last2 ? kt + 1 : last2, nk2, w_load);
                mma_ktile(H + kt * BK, W4_PAIR_LD, w_use, acc1);
            };
            int kt = 0;
            for (; kt + 1 < nk2; kt += 2) {
                k_tile2(kt, sa, sb);
                k_tile2(kt + 1, sb, sa);
            }
            if (kt < nk2) k_tile2(kt, sa, sb);
        }
        if constexpr (MULTI) {
            if (next_row0 >= 0) {                            // chunk 0 for the next tile: lands behind the epilogue
                load_set(w0rsrc, 0, nk0, sa);
                load_set(w0rsrc, nk0 > 1 ? 1 : 0, nk0, sb);
            }
        }
        // ---- epilogue: + the residual tap's activations (registers), rows transposed through H, 1 KiB stores
        __syncthreads();                                     // every wavefront is done reading H
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            float *wr = H + (4 * lh) * W4_PAIR_LD + (wave + 4 * j) * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) wr[((r & 3) + 8 * (r >> 2)) * W4_PAIR_LD] = lrelu(acc1[j][r], slope2) + acc0[j][r];
        }
        __syncthreads();
        {
            const int rd_row = tid >> 6, rd_c4 = (tid & 63) * 4;
            const int N = P.N, ldc = P.ldc;
            const __amdgpu_buffer_rsrc_t crs = act_rsrc(P.c + (size_t)row0 * ldc);
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const int lr = rd_row + 4 * jj, row = row0 + lr;
                if (row >= M) continue;
                const f32x4 v = *reinterpret_cast<const f32x4 *>(H + lr * W4_PAIR_LD + rd_c4);
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
            tile_signal(cnt, __builtin_amdgcn_readfirstlane(te.y), __builtin_amdgcn_readfirstlane(te.z), 1);
        }
    }
}

// ------------------------------------------------------------------------------------ the persistent loop
//
// As gemm_persistent of r3d_kernels.hip: a workgroup walks its list of tile descriptors; DEP: the whole forward's, every
// tile waiting for the ready counters of its producers and raising its own (r3d_device.hpp, wait_deps / tile_signal).
// A descriptor's fourth int carries the tile's width in 32-column granules above bit 8 (4: NB = 1, 8: NB = 2).
template <bool UV, bool DEP>
__device__ __forceinline__ void gemm_persistent4(float *smem) {
    LaunchArgsPtr args = (LaunchArgsPtr)__builtin_amdgcn_kernarg_segment_ptr();
    FwdArgsPtr fargs = (FwdArgsPtr)__builtin_amdgcn_kernarg_segment_ptr();      // (DEP: the same segment holds a FwdArgs)
    constexpr int TS = DEP ? FWD_TILE_INT4 : 1;
    int wg = blockIdx.x;
    if constexpr (!DEP) wg = w4_chunk_of(wg, gridDim.x);   // (XCD- and CU-aware chunk order: observed placement, speed only)
    const int4 *tiles = DEP ? fargs->tiles : args->tiles;
    const int *wg_off = DEP ? fargs->wg_off : args->wg_off;
    const gu32 cnt = DEP ? (gu32)fargs->cnt : (gu32) nullptr;
    const gu32 abort_flag = DEP ? (gu32)(fargs->cnt + fargs->ncnt) : (gu32) nullptr;
    const int t0 = __builtin_amdgcn_readfirstlane(wg_off[wg]);
    const int t1 = __builtin_amdgcn_readfirstlane(wg_off[wg + 1]);
    if constexpr (DEP) {
        // the other bank of ready counters (and its abort flag): zero for the next call, which then runs without r3d_bind_f32
        unsigned *nx = fargs->cnt_next;
        if (nx != nullptr)
            for (int j = blockIdx.x * W4_THREADS + threadIdx.x; j < fargs->ncnt + 4; j += gridDim.x * W4_THREADS) nx[j] = 0u;
    }
    int prev_pi = -1;
    int plain_seen = 0;                      // (test hook: this workgroup's tiles so far)
    for (int t = t0; t < t1; ++t) {
        const int4 td = tiles[t * TS];
        const int pi = __builtin_amdgcn_readfirstlane(td.x & 0xff);
        const bool new_prob = pi != prev_pi;
        prev_pi = pi;
        const int mi = __builtin_amdgcn_readfirstlane(td.x >> 8);
        const int row0 = __builtin_amdgcn_readfirstlane(td.y);
        const int col0 = __builtin_amdgcn_readfirstlane(td.z);
        const int wide = __builtin_amdgcn_readfirstlane(td.w >> 8);          // tile width in 32-column granules
        int sig_base = 0, sig_add = 0;
        if constexpr (DEP) {
            const int4 te = tiles[t * TS + 1];
            const int ndep = __builtin_amdgcn_readfirstlane(te.x);
            sig_base = __builtin_amdgcn_readfirstlane(te.y);
            sig_add = __builtin_amdgcn_readfirstlane(te.z);
            if (ndep > 0) wait_deps(tiles + t * TS, ndep, cnt, abort_flag, fargs->spin_ticks);
        }
        ProbRef P = DEP ? *((const GemmProb __attribute__((address_space(4))) *)fargs->probs + pi) : args->p[pi];
        bool signalled = false;
        do {
            if (P.w3 != nullptr) {       // first level of the pyramid: this workgroup's consecutive tiles of the problem as one run
                int n = 1;
                while (t + n < t1 && __builtin_amdgcn_readfirstlane(tiles[(t + n) * TS].x) == __builtin_amdgcn_readfirstlane(td.x)) ++n;
                const int4 *tl = tiles + t * TS;
                if (P.K <= 64) g4_first<false, UV>(P, tl, TS, n, new_prob, smem, cnt);
                else g4_first<true, UV>(P, tl, TS, n, new_prob, smem, cnt);
                t += n - 1;
                signalled = true;        // (every tile of the run has raised its own counter)
                break;
            }
            if (P.lut != nullptr) {      // gathered operand without the fused level: GlobalInfo.fc_1's current frames
                switch (mi) {
                    case 1: g4_enc<1, UV>(P, row0, col0, new_prob, smem); break;
                    case 2: g4_enc<2, UV>(P, row0, col0, new_prob, smem); break;
                    case 3: g4_enc<3, UV>(P, row0, col0, new_prob, smem); break;
                    default: g4_enc<4, UV>(P, row0, col0, new_prob, smem); break;
                }
                break;
            }
            if (P.w2 != nullptr) {       // fused pair: 32 / 64 rows x all columns
                if (mi >= 2) g4_tile<2, 2, true>(P, row0, col0, smem);
                else g4_tile<1, 2, true>(P, row0, col0, smem);
                break;
            }
            if (wide == 8) {             // 256 columns
                if (mi >= 2) g4_tile<2, 2>(P, row0, col0, smem);
                else g4_tile<1, 2>(P, row0, col0, smem);
                break;
            }
            switch (mi) {                // 128 columns
                case 1: g4_tile<1, 1>(P, row0, col0, smem); break;
                case 2: g4_tile<2, 1>(P, row0, col0, smem); break;
                case 3: g4_tile<3, 1>(P, row0, col0, smem); break;
                default: g4_tile<4, 1>(P, row0, col0, smem); break;
            }
        } while (false);
        if constexpr (DEP) {
            if (!signalled) {            // (the tile functions end on a barrier: drain, one more barrier, raise the counters)
                tile_drain();
                __syncthreads();
                // (test hook: workgroup 0's n-th tile of this kind never reports - what the bounded spins are for)
                if (!(blockIdx.x == 0 && plain_seen++ == fargs->fault_tile1 - 1)) tile_signal(cnt, sig_base, sig_add, mi);
            }
        }
    }
}

#define R3D_W4_ATTR __launch_bounds__(W4_THREADS, 2)       // two waves per SIMD: 256 VGPRs each, two workgroups per CU
extern "C" __global__ R3D_W4_ATTR void r3d_gemm4_f32(const LaunchArgs args_) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    (void)args_;
    gemm_persistent4<false, false>(smem);
}
extern "C" __global__ R3D_W4_ATTR void r3d_gemm4_uv_f32(const LaunchArgs args_) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    (void)args_;
    gemm_persistent4<true, false>(smem);
}
// the whole forward in one launch: 2 x CUs workgroups, all of them resident
extern "C" __global__ R3D_W4_ATTR void r3d_forward4_f32(const FwdArgs args_) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    (void)args_;
    gemm_persistent4<false, true>(smem);
}
extern "C" __global__ R3D_W4_ATTR void r3d_forward4_uv_f32(const FwdArgs args_) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    (void)args_;
    gemm_persistent4<true, true>(smem);
}

static hipError_t w4_attrs() {
    static bool done_dev[64] = {false};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (done_dev[dev]) return hipSuccess;
    for (const void *f : {reinterpret_cast<const void *>(r3d_gemm4_f32), reinterpret_cast<const void *>(r3d_gemm4_uv_f32),
                          reinterpret_cast<const void *>(r3d_forward4_f32), reinterpret_cast<const void *>(r3d_forward4_uv_f32)}) {
        hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, W4_LDS_BYTES);
        if (e != hipSuccess) return e;
    }
    done_dev[dev] = true;
    return hipSuccess;
}

hipError_t launch_gemm4_stage(const LaunchArgs &args, int nwg, bool uv, hipStream_t stream) {
    hipError_t e = w4_attrs();
    if (e != hipSuccess) return e;
    if (uv) r3d_gemm4_uv_f32<<<dim3(nwg), dim3(W4_THREADS), W4_LDS_BYTES, stream>>>(args);
    else r3d_gemm4_f32<<<dim3(nwg), dim3(W4_THREADS), W4_LDS_BYTES, stream>>>(args);
    return hipGetLastError();
}

hipError_t launch_forward4(const FwdArgs &args, int nwg, bool uv, hipStream_t stream) {
    hipError_t e = w4_attrs();
    if (e != hipSuccess) return e;
    if (uv) r3d_forward4_uv_f32<<<dim3(nwg), dim3(W4_THREADS), W4_LDS_BYTES, stream>>>(args);
    else r3d_forward4_f32<<<dim3(nwg), dim3(W4_THREADS), W4_LDS_BYTES, stream>>>(args);
    return hipGetLastError();
}

int forward4_resident_capacity() {
    if (w4_attrs() != hipSuccess) { (void)hipGetLastError(); return 0; }
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(r3d_forward4_f32), W4_THREADS, W4_LDS_BYTES) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return per_cu * device_cu_count();
}

}  // namespace r3d
