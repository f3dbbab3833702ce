// gfx950 (MI355X, CDNA4) kernels of the lifting forward pass.  Written for 64-lane wavefronts and
// the fp32-input matrix cores (v_mfma_f32_32x32x2_f32: exact fp32, 64 FLOP/clk/SIMD, 157 TFLOP/s
// chip peak) - the path is compute-bound, 1e-4 parity rules out bf16.
//
//  r3d_prologue_f32    pointwise: ray encoding (uv -> [(u-cx)/fx, c*y+s, -s*y+c], float64 like the
//                      reference's NumPy, lib/camera/camera.py:423-471) and the camera-embedding MLP
//                      (lib/model/embedding.py:15-18).
//  r3d_gemm_enc_f32    first layer of every temporal branch / GlobalInfo with the input encoding
//                      fused into the operand staging: window gather from a batch or a sliding clip
//                      (lib/train_val/trainer.py:47-58), positional / temporal differences and
//                      body-part grouping (lib/model/rie.py:290-357).
//  r3d_gemm_f32        persistent grouped GEMM + fused epilogue  C = res + lrelu(A W^T + b): every
//                      Conv1d / Linear of TemporalBlock / FCBlock (rie.py:85-105, :122-135, :159-169)
//                      with eval BatchNorm folded.
//  r3d_decode_f32      last Linear of the decoders + joint reassembly (rie.py:409-432) + trajectory
//                      add (lib/train_val/trainer.py:353).
#include <hip/hip_runtime.h>

#include "r3d_internal.hpp"

namespace r3d {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------ GEMM
//
// One persistent launch per DAG level: grid = #CUs, one 512-thread workgroup (8 wavefronts, two per
// SIMD) per CU, owning 144 KiB of LDS.  The host cuts the level's work - all (problem, 256-column
// block, 32-row unit) triples - into one contiguous, cost-balanced chunk per workgroup
// (r3d_schedule.cpp); a chunk is executed as a few tiles of BM = 32*MI rows (MI = 1..8) by 256
// columns.  Wavefront w owns columns [32w, 32w+32) of the tile and all MI row blocks, so any MI is
// perfectly balanced across the 8 wavefronts and the only waste is the 32-row MFMA granularity.
//
// K loop: BK = 32.  A (BM x 32) and W (256 x 32) tiles are staged global -> VGPR -> LDS with a
// one-tile prefetch (tile t+1 is in flight while tile t feeds the matrix cores) and double-buffered
// in LDS; rows are padded to 36 floats so the 16 lanes a ds_read_b128 services together hit 16
// distinct 16-byte slots (no bank conflicts; SQ_LDS_BANK_CONFLICT = 0 in profiles/).
// MFMA operand mapping (v_mfma_f32_32x32x2_f32): lane l supplies A[i = l&31][k = l>>5] and
// B[k = l>>5][j = l&31].  Lane (i, h) reads 4 consecutive floats k = 16h + 4q .. +3 of its row per
// ds_read_b128 and feeds them to 4 MFMAs; since A and W use the same k permutation the sum over
// k is unchanged.

constexpr int LDS_LD = BK + 4;                       // 36 floats = 144 B per staged row
constexpr int GEMM_THREADS = 512;
constexpr int GEMM_BN = 256;
constexpr int GEMM_MAX_MI = 8;
constexpr int STAGE_FLOATS = (GEMM_MAX_MI * 32 + GEMM_BN) * LDS_LD;
constexpr int GEMM_LDS_BYTES = 2 * STAGE_FLOATS * 4; // 147456 B

typedef const LaunchArgs __attribute__((address_space(4))) *LaunchArgsPtr;
typedef const GemmProb __attribute__((address_space(4))) &ProbRef;

template <int MI, bool ENC>
__device__ __forceinline__ void gemm_tile(ProbRef P, const int row0, const int col0, float *smem, long long *dbg) {
    constexpr int NA = (MI + 1) / 2;        // A staging slots per thread (64 rows per slot)
    constexpr int NB = 4;                   // W staging slots per thread (256 rows)
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int M = P.M, N = P.N, K = P.K;
    const int nk = K / BK;
    const int ke0 = P.kend[0], ke1 = P.kend[1], ke2 = P.kend[2];
    const int srow = tid >> 3, a_kq = (tid & 7) * 4;

    int a_row[NA];
    bool a_on[NA];
    // fused prologue (ENC): per staged row, where its first frame and its window's "current" frame
    // start in the raw input (element offsets)
    long long e_first[ENC ? NA : 1], e_cur[ENC ? NA : 1];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int r = srow + 64 * i;
        a_on[i] = (MI % 2 == 0) || (i < NA - 1) || (srow < 32);
        const int gr = row0 + r;
        a_row[i] = gr < M ? gr : M - 1;
        if (ENC) {
            const int win = a_row[i] / P.enc_rows, t3 = a_row[i] - win * P.enc_rows;
            const long long wbase = (long long)win * P.enc_ws;
            e_first[i] = wbase + (long long)t3 * 3 * P.enc_jf;
            e_cur[i] = wbase + P.enc_cur;
        }
    }
    const float *ex = ENC ? P.x : nullptr;
    const int *elut = ENC ? P.lut + a_kq : nullptr;
    const float *w_ptr = P.w + (size_t)(col0 + srow) * K + a_kq;
    const size_t w_step = (size_t)64 * K;
    const int st_off = srow * LDS_LD + a_kq;           // staging offset inside a 64-row slot

    f32x4 ra[NA], rb[NB];
    auto load_global = [&](int kt) {
        const int kb = kt * BK;
        if (ENC) {
            // A[row][k] = x[first/cur + off1] - x[first/cur + off2]: ray differences and body-part
            // gather (lib/model/rie.py:290-357) evaluated while staging; nothing is materialised
            const int4 code = *reinterpret_cast<const int4 *>(elut + kb);
            const int cd[4] = {code.x, code.y, code.z, code.w};
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                if (!a_on[i]) continue;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int c = cd[e], kind = (c >> 20) & 3;
                    const long long b1 = (c >> 22) & 1 ? e_cur[i] : e_first[i];
                    const long long b2 = kind == 2 ? e_cur[i] : e_first[i];
                    const float v1 = ex[b1 + (c & 1023)];
                    const float v2 = ex[b2 + ((c >> 10) & 1023)];
                    ra[i][e] = kind == 3 ? 0.0f : (kind == 0 ? v1 : v1 - v2);
                }
            }
        } else {
            // K segment of the (virtually concatenated) A operand this tile falls in (uniform -> scalar loads)
            const int si = (kb >= ke0) + (kb >= ke1) + (kb >= ke2);
            const float *base = P.a[si];
            const int ld = P.lda[si];
            const int k0 = si ? P.kend[si - 1] : 0;
            const int kofs = kb - k0 + a_kq;
#pragma unroll
            for (int i = 0; i < NA; ++i)
                if (a_on[i]) ra[i] = *reinterpret_cast<const f32x4 *>(base + (size_t)a_row[i] * ld + kofs);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) rb[i] = *reinterpret_cast<const f32x4 *>(w_ptr + i * w_step + kb);
    };
    auto store_lds = [&](int buf) {
        float *s = smem + buf * STAGE_FLOATS + st_off;
#pragma unroll
        for (int i = 0; i < NA; ++i)
            if (a_on[i]) *reinterpret_cast<f32x4 *>(s + i * 64 * LDS_LD) = ra[i];
#pragma unroll
        for (int i = 0; i < NB; ++i) *reinterpret_cast<f32x4 *>(s + (GEMM_MAX_MI * 32 + i * 64) * LDS_LD) = rb[i];
    };

    f32x16 acc[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][r] = 0.0f;

    const int a_frag = li * LDS_LD + lh * 16;
    const int b_frag = (GEMM_MAX_MI * 32 + wave * 32 + li) * LDS_LD + lh * 16;

#ifdef R3D_TIMING
#define R3D_STAMP(slot) do { if (dbg && tid == 0 && kt < 32) dbg[kt * 8 + (slot)] = __builtin_readcyclecounter(); } while (0)
#else
#define R3D_STAMP(slot) do { } while (0)
#endif
    load_global(0);
    store_lds(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        R3D_STAMP(0);
        if (more) load_global(kt + 1);
        R3D_STAMP(1);
        const float *s = smem + (kt & 1) * STAGE_FLOATS;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 av[MI];
            const f32x4 bv = *reinterpret_cast<const f32x4 *>(s + b_frag + q * 4);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
                av[mi] = *reinterpret_cast<const f32x4 *>(s + a_frag + mi * 32 * LDS_LD + q * 4);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
                    acc[mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mi][kk], bv[kk], acc[mi], 0, 0, 0);
        }
        R3D_STAMP(2);
        if (more) store_lds((kt + 1) & 1);
        R3D_STAMP(3);
        __syncthreads();
        R3D_STAMP(4);
    }
#ifdef R3D_TIMING
    if (dbg && tid == 0) dbg[255] = __builtin_readcyclecounter();
#endif

    // epilogue: C = res + lrelu(acc + bias).  C/D layout of the 32x32 MFMA: col = lane & 31,
    // row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).  A wavefront store instruction writes two
    // 128-byte row pieces (lanes 0-31 and 32-63).
    const int col = col0 + wave * 32 + li;
    const float slope = P.slope;
    const float *res = P.res;
    float *c = P.c;
    const int ldc = P.ldc, ldr = P.ldr;
    const float bias = P.bias[col];
    const bool full = (row0 + MI * 32 <= M) && (col0 + GEMM_BN <= N);   // wave-uniform
    if (full) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const size_t rbase = (size_t)(row0 + mi * 32 + 4 * lh);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const size_t row = rbase + (r & 3) + 8 * (r >> 2);
                float v = acc[mi][r] + bias;
                v = v > 0.0f ? v : v * slope;
                if (res) v += res[row * ldr + col];
                c[row * ldc + col] = v;
            }
        }
    } else if (col < N) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (row < M) {
                    float v = acc[mi][r] + bias;
                    v = v > 0.0f ? v : v * slope;
                    if (res) v += res[(size_t)row * ldr + col];
                    c[(size_t)row * ldc + col] = v;
                }
            }
        }
    }
}

template <bool ENC>
__device__ __forceinline__ void gemm_persistent(float *smem) {
    LaunchArgsPtr args = (LaunchArgsPtr)__builtin_amdgcn_kernarg_segment_ptr();
    // XCD-aware chunk order: workgroup b runs on XCD b % 8 (observed; speed only), so give each XCD a
    // contiguous run of chunks - neighbouring chunks share weights (and A rows) through that XCD's L2.
    int wg = blockIdx.x;
    {
        const int n = gridDim.x, q = n >> 3, r = n & 7, xcd = wg & 7, idx = wg >> 3;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int t0 = __builtin_amdgcn_readfirstlane(args->wg_off[wg]);
    const int t1 = __builtin_amdgcn_readfirstlane(args->wg_off[wg + 1]);
    long long *dbg = nullptr;
#ifdef R3D_TIMING
    if (args->dbg && wg < 4) dbg = args->dbg + wg * 256;
    if (args->dbg && threadIdx.x == 0) {
        args->dbg[1024 + wg * 4 + 0] = __builtin_readcyclecounter();
        args->dbg[1024 + wg * 4 + 2] = wall_clock64();
    }
#endif
    for (int t = t0; t < t1; ++t) {
        const int4 td = args->tiles[t];
        const int pi = __builtin_amdgcn_readfirstlane(td.x & 0xff);
        const int mi = __builtin_amdgcn_readfirstlane(td.x >> 8);
        const int row0 = __builtin_amdgcn_readfirstlane(td.y);
        const int col0 = __builtin_amdgcn_readfirstlane(td.z);
        ProbRef P = args->p[pi];
        switch (mi) {
            case 1: gemm_tile<1, ENC>(P, row0, col0, smem, dbg); break;
            case 2: gemm_tile<2, ENC>(P, row0, col0, smem, dbg); break;
            case 3: gemm_tile<3, ENC>(P, row0, col0, smem, dbg); break;
            case 4: gemm_tile<4, ENC>(P, row0, col0, smem, dbg); break;
            case 5: gemm_tile<5, ENC>(P, row0, col0, smem, dbg); break;
            case 6: gemm_tile<6, ENC>(P, row0, col0, smem, dbg); break;
            case 7: gemm_tile<7, ENC>(P, row0, col0, smem, dbg); break;
            default: gemm_tile<8, ENC>(P, row0, col0, smem, dbg); break;
        }
    }
#ifdef R3D_TIMING
    if (args->dbg && threadIdx.x == 0) {
        args->dbg[1024 + wg * 4 + 1] = __builtin_readcyclecounter();
        args->dbg[1024 + wg * 4 + 3] = wall_clock64();
    }
#endif
}

// every layer whose input is an activation matrix in HBM
extern "C" __global__ __launch_bounds__(GEMM_THREADS) void r3d_gemm_f32(const LaunchArgs args_) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    (void)args_;
    gemm_persistent<false>(smem);
}

// first layers (expand_conv of every temporal branch, GlobalInfo.fc_1): input encoding fused in
extern "C" __global__ __launch_bounds__(GEMM_THREADS) void r3d_gemm_enc_f32(const LaunchArgs args_) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    (void)args_;
    gemm_persistent<true>(smem);
}

hipError_t launch_gemm_stage(const LaunchArgs &args, int nwg, bool encode, hipStream_t stream) {
    static bool attr_done = false;
    if (!attr_done) {
        // 144 KiB of dynamic LDS exceeds the 64 KiB default cap
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(r3d_gemm_f32),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(r3d_gemm_enc_f32),
                                hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    if (encode)
        r3d_gemm_enc_f32<<<dim3(nwg), dim3(GEMM_THREADS), GEMM_LDS_BYTES, stream>>>(args);
    else
        r3d_gemm_f32<<<dim3(nwg), dim3(GEMM_THREADS), GEMM_LDS_BYTES, stream>>>(args);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------ prologue

// Pointwise front end.  (1) UV mode: pixel keypoints -> rays, float64 like the reference's NumPy
// (lib/camera/camera.py:438-439 then pt_cam @ Rc2n^T, :471, Rc2n = Rx(pitch), :333-338):
// ray = ((u-cx)/fx, c*y + s, -s*y + c) with y = (v-cy)/fy.  (2) camera-embedding MLP
// (lib/model/embedding.py:15-18; LeakyReLU slope 0.01, BatchNorm folded) for each network.
extern "C" __global__ __launch_bounds__(256) void r3d_prologue_f32(const PrologueArgs a) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (a.uv) {
        const long long n = a.frames * a.J;
        if (gid < n) {
            const long long frame = gid / a.J;
            // the camera of the (first) window this frame belongs to
            long long win = a.window_stride >= a.RF ? frame / a.window_stride : frame - (a.RF - 1);
            win = win < 0 ? 0 : (win >= a.B ? a.B - 1 : win);
            const double *cam = a.cam + win * a.cam_stride;
            const double u = a.uv[gid * 2], v = a.uv[gid * 2 + 1];
            const double y = (v - cam[3]) / cam[1];
            a.rays[gid * 3 + 0] = (float)((u - cam[2]) / cam[0]);
            a.rays[gid * 3 + 1] = (float)(cam[4] * y + cam[5]);
            a.rays[gid * 3 + 2] = (float)(-cam[5] * y + cam[4]);
        }
    }
    for (int m = 0; m < a.nembed; ++m) {
        const int D = a.emb_dim[m], E = a.E;
        if (gid >= a.B * D) continue;
        const long long b = gid / D;
        const int o = (int)(gid - b * D);
        const float *w1 = a.emb_w[m], *b1 = w1 + EMBED_MID * E, *w2 = b1 + EMBED_MID, *b2 = w2 + D * EMBED_MID;
        const float *p = a.param + b * a.param_stride;
        float acc = b2[o];
        for (int k = 0; k < EMBED_MID; ++k) {
            float h = b1[k];
            for (int e = 0; e < E; ++e) h += w1[k * E + e] * p[e];
            h = h > 0.0f ? h : 0.01f * h;
            acc += w2[o * EMBED_MID + k] * h;
        }
        a.emb_out[m][gid] = acc > 0.0f ? acc : 0.01f * acc;
    }
}

hipError_t launch_prologue(const PrologueArgs &args, hipStream_t stream) {
    long long most = args.uv ? args.frames * args.J : 0;
    for (int m = 0; m < args.nembed; ++m) most = most > args.B * args.emb_dim[m] ? most : args.B * args.emb_dim[m];
    if (most == 0) return hipSuccess;
    r3d_prologue_f32<<<dim3((unsigned)((most + 255) / 256)), dim3(256), 0, stream>>>(args);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------ decoder tail

// One wavefront per window: the final Linear(1024 -> 3 n_g) of each Integration block
// (lib/model/rie.py:409-413, :557) as 64-lane dot products, written straight into the joint slot
// the reference's reassembly puts it (rie.py:415-432), plus the trajectory broadcast add
// (lib/train_val/trainer.py:353).  These layers are 0.05 % of the FLOPs; as GEMMs their N = 3..15
// would waste a 256-column tile.
extern "C" __global__ __launch_bounds__(256) void r3d_decode_f32(const DecodeArgs a) {
    const int lane = threadIdx.x & 63;
    const long long b = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= a.B) return;
    float trj0 = 0.0f, trj1 = 0.0f, trj2 = 0.0f;   // scalars: a runtime-indexed array would live in scratch
    // trajectory source first so that its result can be added to every joint
    for (int pass = 0; pass < 2; ++pass) {
        for (int s = 0; s < a.nsrc; ++s) {
            const bool is_trj = a.has_trj && s == a.nsrc - 1;
            if ((pass == 0) != is_trj) continue;
            const float *h = a.h[s] + b * MLP_HIDDEN;
            f32x4 hv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) hv[j] = *reinterpret_cast<const f32x4 *>(h + j * 256 + lane * 4);
            for (int o = 0; o < a.n_out[s]; ++o) {
                const float *w = a.w[s] + (size_t)o * MLP_HIDDEN;
                float acc = 0.0f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 wv = *reinterpret_cast<const f32x4 *>(w + j * 256 + lane * 4);
                    acc += hv[j][0] * wv[0] + hv[j][1] * wv[1] + hv[j][2] * wv[2] + hv[j][3] * wv[3];
                }
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
                acc += a.bias[s][o];
                if (is_trj) {
                    if (o == 0) trj0 = acc; else if (o == 1) trj1 = acc; else trj2 = acc;
                    if (lane == 0) {
                        if (a.out_trj) a.out_trj[b * 3 + o] = acc;
                        if (!a.has_pos) a.out[b * 3 + o] = acc;
                    }
                } else if (lane == 0) {
                    const int e = a.slot[a.first[s] + o];
                    const int c3 = e % 3;
                    a.out[b * (a.J * 3) + e] = acc + (c3 == 0 ? trj0 : c3 == 1 ? trj1 : trj2);
                }
            }
        }
    }
}

hipError_t launch_decode(const DecodeArgs &args, hipStream_t stream) {
    r3d_decode_f32<<<dim3((unsigned)((args.B + 3) / 4)), dim3(256), 0, stream>>>(args);
    return hipGetLastError();
}

}  // namespace r3d
