// gfx950 (MI355X, CDNA4) kernels of the lifting forward pass.  Written for 64-lane wavefronts and
// the fp32-input matrix cores (v_mfma_f32_32x32x2_f32: exact fp32, 64 FLOP/clk/SIMD, 157 TFLOP/s
// chip peak) - the path is compute-bound, 1e-4 parity rules out bf16.
//
//  r3d_encode_f32      prologue: ray encoding (uv -> [(u-cx)/fx, c*y+s, -s*y+c], float64 like the
//                      reference's NumPy, lib/camera/camera.py:423-471), window gather from a batch
//                      or from a sliding clip (lib/train_val/trainer.py:47-58), positional /
//                      temporal differences and body-part grouping (lib/model/rie.py:290-357),
//                      camera-embedding MLP (lib/model/embedding.py:15-18).
//  r3d_gemm_f32_t128   grouped GEMM + fused epilogue  C = res + lrelu(A W^T + b): every Conv1d /
//  r3d_gemm_f32_t64    Linear of TemporalBlock / FCBlock (rie.py:85-105, :122-135, :159-169) with
//                      eval BatchNorm folded; 128x128 and 64x64 workgroup tiles.
//  r3d_assemble_f32    epilogue: joint reassembly (rie.py:415-432) + trajectory add
//                      (lib/train_val/trainer.py:353).
#include <hip/hip_runtime.h>

#include "r3d_internal.hpp"

namespace r3d {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------ GEMM

constexpr int LDS_LD = BK + 4;   // +16 B pad: ds_read_b128 of 16 rows hits 16 distinct 16-B slots

template <int BM, int BN>
struct GemmCfg {
    static constexpr int THREADS = 256;                 // 4 wavefronts, 2 (M) x 2 (N)
    static constexpr int MI = BM / 64, NI = BN / 64;    // 32x32 MFMA tiles per wavefront
    static constexpr int A_V4 = BM * BK / 4 / THREADS;  // float4 global loads per thread per K tile
    static constexpr int B_V4 = BN * BK / 4 / THREADS;
    static constexpr int STAGE_FLOATS = (BM + BN) * LDS_LD;
    static constexpr int LDS_BYTES = 2 * STAGE_FLOATS * 4;
};

// The launch descriptor lives in the kernarg segment; it is read through a constant-address-space
// pointer so that the (wave-uniform) dynamic problem index turns into scalar loads, not scratch.
typedef const StageArgs __attribute__((address_space(4))) *StageArgsPtr;

template <int BM, int BN>
__device__ __forceinline__ void gemm_body(StageArgsPtr argp, float *smem) {
    const StageArgs __attribute__((address_space(4))) &args = *argp;
    using Cfg = GemmCfg<BM, BN>;
    constexpr int MI = Cfg::MI, NI = Cfg::NI, A_V4 = Cfg::A_V4, B_V4 = Cfg::B_V4;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;

    // XCD-aware order: consecutive logical tiles (which share A rows / weights) stay on one XCD's L2.
    // Workgroup b runs on XCD b % 8 (observed; affects speed only).
    int bid = blockIdx.x;
    {
        const int n = args.total_tiles, q = n >> 3, r = n & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    int pi = 0;
#pragma unroll
    for (int i = 1; i < MAX_PROB; ++i)
        if (i < args.nprob && bid >= args.p[i].tile_begin) pi = i;
    const GemmProb __attribute__((address_space(4))) &P = args.p[pi];
    const int t = bid - P.tile_begin;
    const int tn = t % P.tiles_n, tm = t / P.tiles_n;
    const int row0 = tm * BM, col0 = tn * BN;
    const int M = P.M, N = P.N, K = P.K;
    const int nk = K / BK;

    const int ke0 = P.kend[0], ke1 = P.kend[1], ke2 = P.kend[2];

    // per-thread staging coordinates (8 float4 per 32-float tile row)
    int a_row[A_V4], a_lds[A_V4];
#pragma unroll
    for (int i = 0; i < A_V4; ++i) {
        const int v = tid + i * Cfg::THREADS, r = v >> 3, kq = v & 7;
        int gr = row0 + r;
        a_row[i] = gr < M ? gr : M - 1;
        a_lds[i] = r * LDS_LD + kq * 4;
    }
    const float *b_ptr[B_V4];
    int b_lds[B_V4];
#pragma unroll
    for (int i = 0; i < B_V4; ++i) {
        const int v = tid + i * Cfg::THREADS, r = v >> 3, kq = v & 7;
        b_ptr[i] = P.w + (size_t)(col0 + r) * K + kq * 4;
        b_lds[i] = BM * LDS_LD + r * LDS_LD + kq * 4;
    }
    const int a_kq = (tid & 7) * 4;

    f32x4 ra[A_V4], rb[B_V4];
    auto load_global = [&](int kt) {
        const int kb = kt * BK;
        // which K segment of the (virtually concatenated) A operand this tile falls in; the
        // descriptor is in constant memory, so the uniform index becomes three scalar loads
        const int si = (kb >= ke0) + (kb >= ke1) + (kb >= ke2);
        const float *base = P.a[si];
        const int ld = P.lda[si];
        const int k0 = si ? P.kend[si - 1] : 0;
        const int kofs = kb - k0 + a_kq;
#pragma unroll
        for (int i = 0; i < A_V4; ++i)
            ra[i] = *reinterpret_cast<const f32x4 *>(base + (size_t)a_row[i] * ld + kofs);
#pragma unroll
        for (int i = 0; i < B_V4; ++i) rb[i] = *reinterpret_cast<const f32x4 *>(b_ptr[i] + kb);
    };
    auto store_lds = [&](int buf) {
        float *s = smem + buf * Cfg::STAGE_FLOATS;
#pragma unroll
        for (int i = 0; i < A_V4; ++i) *reinterpret_cast<f32x4 *>(s + a_lds[i]) = ra[i];
#pragma unroll
        for (int i = 0; i < B_V4; ++i) *reinterpret_cast<f32x4 *>(s + b_lds[i]) = rb[i];
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.0f;

    // fragment read offsets: lane (li, lh) reads row li, K chunk [lh*16 + q*4, +4)
    const int a_frag = (wm * (BM / 2) + li) * LDS_LD + lh * 16;
    const int b_frag = BM * LDS_LD + (wn * (BN / 2) + li) * LDS_LD + lh * 16;

    load_global(0);
    store_lds(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        if (more) load_global(kt + 1);
        const float *s = smem + (kt & 1) * Cfg::STAGE_FLOATS;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 av[MI], bv[NI];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
                av[mi] = *reinterpret_cast<const f32x4 *>(s + a_frag + mi * 32 * LDS_LD + q * 4);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
                bv[ni] = *reinterpret_cast<const f32x4 *>(s + b_frag + ni * 32 * LDS_LD + q * 4);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mi][kk], bv[ni][kk], acc[mi][ni], 0, 0, 0);
        }
        if (more) store_lds((kt + 1) & 1);
        __syncthreads();
    }

    // epilogue: C = res + lrelu(acc + bias).  C/D layout of the 32x32 MFMA: col = lane & 31,
    // row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
    const float slope = P.slope;
    const float *res = P.res;
    float *c = P.c;
    const int ldc = P.ldc, ldr = P.ldr;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int col = col0 + wn * (BN / 2) + ni * 32 + li;
        if (col >= N) continue;
        const float bias = P.bias[col];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + wm * (BM / 2) + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (row < M) {
                    float v = acc[mi][ni][r] + bias;
                    v = v > 0.0f ? v : v * slope;
                    if (res) v += res[(size_t)row * ldr + col];
                    c[(size_t)row * ldc + col] = v;
                }
            }
        }
    }
}

extern "C" __global__ __launch_bounds__(256) void r3d_gemm_f32_t128(const StageArgs args) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    (void)args;
    gemm_body<128, 128>((StageArgsPtr)__builtin_amdgcn_kernarg_segment_ptr(), smem);
}

extern "C" __global__ __launch_bounds__(256) void r3d_gemm_f32_t64(const StageArgs args) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    (void)args;
    gemm_body<64, 64>((StageArgsPtr)__builtin_amdgcn_kernarg_segment_ptr(), smem);
}

const char *gemm_kernel_name(int tile) { return tile == 128 ? "r3d_gemm_f32_t128" : "r3d_gemm_f32_t64"; }

hipError_t launch_gemm_stage(const StageArgs &args, int tile, hipStream_t stream) {
    constexpr int kLds128 = GemmCfg<128, 128>::LDS_BYTES, kLds64 = GemmCfg<64, 64>::LDS_BYTES;
    static bool attr_done = false;
    if (!attr_done) {
        // 72 KiB of dynamic LDS exceeds the 64 KiB default cap
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(r3d_gemm_f32_t128),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, kLds128);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    if (tile == 128)
        r3d_gemm_f32_t128<<<dim3(args.total_tiles), dim3(256), kLds128, stream>>>(args);
    else
        r3d_gemm_f32_t64<<<dim3(args.total_tiles), dim3(256), kLds64, stream>>>(args);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------ prologue

// One input feature of window b, frame t: element `src` = joint*F + f of the (J,F) frame.
template <int MODE>
__device__ __forceinline__ float fetch_feature(const EncodeArgs &a, long long b, int t, int src, int f) {
    const long long frame = b * a.window_stride + t;
    if (MODE == R3D_INPUT_RAYS) {
        return a.x[frame * (a.J * a.F) + src];
    } else {
        // lib/camera/camera.py:438-439 then pt_cam @ Rc2n^T (:471) with Rc2n = Rx(pitch) (:333-338)
        const int joint = src / 3;
        const float *uv = a.x + (frame * a.J + joint) * 2;
        const double *cam = a.cam + b * a.cam_stride;
        if (f == 0) return (float)(((double)uv[0] - cam[2]) / cam[0]);
        const double y = ((double)uv[1] - cam[3]) / cam[1];
        return f == 1 ? (float)(cam[4] * y + cam[5]) : (float)(-cam[5] * y + cam[4]);
    }
}

template <int MODE>
__device__ __forceinline__ void encode_body(const EncodeArgs &a) {
    const int by = blockIdx.y;
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int T3 = a.RF / 3;
    if (by < a.nbranch) {
        // first-layer GEMM operand of one temporal branch: row (b, t3) = frames 3*t3 .. 3*t3+2,
        // column = tap*Cin + channel of cat(x_g, x_g - root, x_g - x_current)
        const EncodeBranch br = a.br[by];
        const int v4_per_row = br.k0pad >> 2;
        const long long total = a.B * T3 * v4_per_row;
        if (gid >= total) return;
        const long long row = gid / v4_per_row;
        const int c4 = (int)(gid - row * v4_per_row) * 4;
        const long long b = row / T3;
        const int t3 = (int)(row - b * T3);
        f32x4 out;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int code = br.lut[c4 + e];
            const int tap = code & 3, kind = (code >> 2) & 3, src = (code >> 4) & 255, f = (code >> 12) & 3;
            float v = 0.0f;
            if (kind != 3) {
                const int t = 3 * t3 + tap;
                v = fetch_feature<MODE>(a, b, t, src, f);
                if (kind == 1) v -= fetch_feature<MODE>(a, b, t, f, f);              // root joint, rie.py:301
                else if (kind == 2) v -= fetch_feature<MODE>(a, b, a.tcur, src, f);  // frame RF//F, rie.py:304
            }
            out[e] = v;
        }
        *reinterpret_cast<f32x4 *>(br.a0 + row * br.k0pad + c4) = out;
        return;
    }
    // last slice of the grid: current-frame matrix (rie.py:290-292) and camera embeddings
    const long long total = a.B * CUR_LD;
    if (gid >= total) return;
    const long long b = gid / CUR_LD;
    const int c = (int)(gid - b * CUR_LD);
    const int JF = a.J * a.F;
    a.cur[gid] = c < JF ? fetch_feature<MODE>(a, b, a.tcur, c, c % a.F) : 0.0f;
    for (int m = 0; m < a.nembed; ++m) {
        const int D = a.emb_dim[m], E = a.E;
        const float *w1 = a.emb_w[m], *b1 = w1 + EMBED_MID * E, *w2 = b1 + EMBED_MID, *b2 = w2 + D * EMBED_MID;
        const float *p = a.param + b * a.param_stride;
        for (int o = c; o < D; o += CUR_LD) {
            float acc = b2[o];
            for (int k = 0; k < EMBED_MID; ++k) {
                float h = b1[k];
                for (int e = 0; e < E; ++e) h += w1[k * E + e] * p[e];
                h = h > 0.0f ? h : 0.01f * h;                 // nn.LeakyReLU() default slope
                acc += w2[o * EMBED_MID + k] * h;
            }
            a.emb_out[m][b * D + o] = acc > 0.0f ? acc : 0.01f * acc;
        }
    }
}

extern "C" __global__ __launch_bounds__(256) void r3d_encode_f32(const EncodeArgs a) {
    if (a.mode == R3D_INPUT_RAYS) encode_body<R3D_INPUT_RAYS>(a);
    else encode_body<R3D_INPUT_UV>(a);
}

hipError_t launch_encode(const EncodeArgs &args, hipStream_t stream, int *blocks) {
    long long most = args.B * CUR_LD;
    for (int i = 0; i < args.nbranch; ++i) {
        const long long t = args.B * (args.RF / 3) * (args.br[i].k0pad >> 2);
        if (t > most) most = t;
    }
    const unsigned gx = (unsigned)((most + 255) / 256);
    if (blocks) *blocks = (int)(gx * (args.nbranch + 1));
    hipLaunchKernelGGL(r3d_encode_f32, dim3(gx, args.nbranch + 1), dim3(256), 0, stream, args);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------ epilogue

extern "C" __global__ __launch_bounds__(256) void r3d_assemble_f32(const AssembleArgs a) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int per = a.dec ? a.J * 3 : 3;
    if (gid >= a.B * per) return;
    const long long b = gid / per;
    const int e = (int)(gid - b * per);
    float v = 0.0f;
    if (a.dec) v = a.dec[b * (5 * DEC_SLOT) + a.src[e]];
    if (a.trj) v += a.trj[b * a.ldt + e % 3];
    a.out[gid] = v;
}

hipError_t launch_assemble(const AssembleArgs &args, hipStream_t stream, int *blocks) {
    const long long total = args.B * (args.dec ? args.J * 3 : 3);
    const unsigned gx = (unsigned)((total + 255) / 256);
    if (blocks) *blocks = (int)gx;
    hipLaunchKernelGGL(r3d_assemble_f32, dim3(gx), dim3(256), 0, stream, args);
    return hipGetLastError();
}

}  // namespace r3d
