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
#include "r3d_tiles.hpp"

namespace r3d {

// Ahead of r3d_forward_f32 on the same stream: zero the call's ready counters and abort flag, and turn the schedule's
// relative problem table (pointer fields = byte offsets, one base tag per field) into this call's absolute one - the
// table is too large for a kernarg segment (84 problems at RF 243), and it names the caller's buffers.
extern "C" __global__ __launch_bounds__(256) void r3d_bind_f32(const BindArgs b) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, total = gridDim.x * blockDim.x;
    for (int j = i; j < b.ncnt + 4; j += total) b.cnt[j] = 0u;
    if (b.arm != nullptr) {          // poll mode: every activation of the call's bank(s) starts as "not there yet"
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 sv = {ACT_SENTINEL, ACT_SENTINEL, ACT_SENTINEL, ACT_SENTINEL};
        for (long long j = i; j < b.arm_vec4; j += total) reinterpret_cast<u32x4 *>(b.arm)[j] = sv;
    }
    if (i >= b.nprob) return;
    GemmProb g = b.rel[i];
    const unsigned char *tg = b.tags + (size_t)i * BIND_NPTR;
    auto fix = [&](auto &ptr, int k) {
        const int tag = tg[k];
        typedef typename std::remove_reference<decltype(ptr)>::type PT;
        ptr = tag == BIND_NULL ? (PT) nullptr : (PT)((const char *)b.base[tag] + (size_t)ptr);
    };
    for (int sgi = 0; sgi < MAX_SEG; ++sgi) {
        if (tg[sgi] == BIND_PARAM) g.lda[sgi] = b.param_stride;
        fix(g.a[sgi], sgi);
    }
    fix(g.w, 4); fix(g.bias, 5); fix(g.res, 6); fix(g.c, 7); fix(g.w2, 8); fix(g.bias2, 9); fix(g.wb3, 10); fix(g.w2b3, 11);
    fix(g.w3b3, 12); fix(g.w3, 13); fix(g.bias3, 14); fix(g.lut, 15); fix(g.x, 16); fix(g.cam, 17); fix(g.wchain, 18);
    if (g.lut != nullptr) {
        g.enc_ws = b.enc_ws;
        g.enc_bytes = b.enc_bytes;
        g.cam_stride = b.cam_stride;
    }
    b.out[i] = g;
}


hipError_t launch_gemm_stage(const LaunchArgs &args, int nwg, int kind, bool uv, hipStream_t stream) {
    bool b3 = false;                 // some problem runs on the bf16 matrix cores: the kernel that carries those tile kinds
    for (int i = 0; i < args.nprob; ++i) b3 = b3 || args.p[i].wb3 != nullptr;
    // more dynamic LDS than the 64 KiB default cap: raised once per device (a process may drive several)
    static bool attr_done_dev[64] = {false};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    bool &attr_done = attr_done_dev[dev];
    if (!attr_done) {
        for (int u = 0; u < 2; ++u) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_kernel_f32(u != 0)), hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES);
            if (e != hipSuccess) return e;
            if ((e = hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_kernel_b3(u != 0)), hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES)) != hipSuccess) return e;
            if ((e = hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_kernel_enc(u != 0)), hipFuncAttributeMaxDynamicSharedMemorySize, ENC_LDS_BYTES)) != hipSuccess) return e;
        }
        attr_done = true;
    }
    if (kind == STAGE_ENC) gemm_kernel_enc(uv)<<<dim3(nwg), dim3(GEMM_THREADS), ENC_LDS_BYTES, stream>>>(args);
    else (b3 ? gemm_kernel_b3(uv) : gemm_kernel_f32(uv))<<<dim3(nwg), dim3(GEMM_THREADS), GEMM_LDS_BYTES, stream>>>(args);
    return hipGetLastError();
}

static FwdKernel forward_kernel(int kind, bool uv) {
    switch (kind) {
        case FWD_KERNEL_B3: return fwd_kernel_b3(uv);
        case FWD_KERNEL_LAT: return fwd_kernel_lat(uv);
        case FWD_KERNEL_CLIP: return fwd_kernel_clip(uv);
        case FWD_KERNEL_CHAIN: return fwd_kernel_chain(uv);
        default: return fwd_kernel_f32(uv);
    }
}
const char *forward_kernel_name(int kind, bool uv) {
    switch (kind) {
        case FWD_KERNEL_B3: return uv ? "r3d_forward_uv_b3" : "r3d_forward_b3";
        case FWD_KERNEL_LAT: return uv ? "r3d_forward_uv_lat" : "r3d_forward_lat";
        case FWD_KERNEL_CLIP: return uv ? "r3d_forward_clip_uv_f32" : "r3d_forward_clip_f32";
        case FWD_KERNEL_CHAIN: return "r3d_forward_chain_f32";
        default: return uv ? "r3d_forward_uv_f32" : "r3d_forward_f32";
    }
}

// every specialisation may use the whole LDS allocation (set once per device, before the first launch AND before the occupancy query)
static hipError_t forward_set_lds_attr() {
    static bool attr_done_dev[64] = {false};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!attr_done_dev[dev]) {
        for (int k = 0; k < FWD_KERNEL_COUNT; ++k)
            for (int u = 0; u < 2; ++u) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(forward_kernel(k, u != 0)), hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES);
                if (e != hipSuccess) return e;
            }
        attr_done_dev[dev] = true;
    }
    return hipSuccess;
}

hipError_t launch_forward(const FwdArgs &args, int nwg, int kind, bool uv, hipStream_t stream) {
    if (hipError_t e = forward_set_lds_attr(); e != hipSuccess) return e;
    forward_kernel(kind, uv)<<<dim3(nwg), dim3(GEMM_THREADS), GEMM_LDS_BYTES, stream>>>(args);
    return hipGetLastError();
}

// Workgroups of the single-launch forward that can be resident at once on the current device (it needs ALL of its grid
// resident: a waiting workgroup spins for tiles of workgroups that must be running).  0: unknown.
int forward_resident_capacity(int kind, bool uv) {
    int per_cu = 0;
    if (forward_set_lds_attr() != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(forward_kernel(kind, uv)), GEMM_THREADS, GEMM_LDS_BYTES) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return per_cu * device_cu_count();
}

hipError_t launch_bind(const BindArgs &args, hipStream_t stream) {
    const int threads = std::max(args.nprob, std::min((int)std::max<long long>(args.ncnt + 4, args.arm ? args.arm_vec4 : 0), 256 * 256));
    r3d_bind_f32<<<dim3((threads + 255) / 256), dim3(256), 0, stream>>>(args);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------ decoder tail

// One wavefront per (window, decoder): the final Linear(1024 -> 3 n_g) of an Integration block
// (lib/model/rie.py:409-413, :557) as 64-lane dot products, three outputs (one joint) at a time,
// written straight into the joint slot the reference's reassembly puts it (rie.py:415-432) plus the
// trajectory broadcast add (lib/train_val/trainer.py:353).  Every body-part wavefront recomputes the
// 3-output trajectory head itself - cheaper than a dependency between wavefronts.  These layers are
// 0.05 % of the FLOPs; as GEMMs their N = 3..15 would waste a 256-column tile.
// Three output rows (one joint) of a decoder against one hidden row: all loads first, then the reductions.
// Sum over the 64 lanes with DPP adds (six dependent VALU instructions; __shfl_xor goes through the LDS crossbar, ~8x the
// latency per step, and the decoder kernel is a chain of such reductions): pairs, quads, half rows, rows of 16, then
// row 0 -> 1 and 2 -> 3 (row_bcast:15), rows 0-1 -> 2-3 (row_bcast:31); lane 63 holds the total.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float x) {              // (rows outside the mask add 0)
    return x + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_sum(float v) {
    v = dpp_add<0xb1, 0xf>(v);       // quad_perm [1,0,3,2]
    v = dpp_add<0x4e, 0xf>(v);       // quad_perm [2,3,0,1]
    v = dpp_add<0x141, 0xf>(v);      // row_half_mirror
    v = dpp_add<0x140, 0xf>(v);      // row_mirror
    v = dpp_add<0x142, 0xa>(v);      // row_bcast:15 into rows 1 and 3
    v = dpp_add<0x143, 0xc>(v);      // row_bcast:31 into rows 2 and 3
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

struct DecodeJoint {
    f32x4 w[3][4];
    __device__ __forceinline__ void load(const float *wrows, int lane) {
#pragma unroll
        for (int n = 0; n < 3; ++n)
#pragma unroll
            for (int j = 0; j < 4; ++j) w[n][j] = gload4(wrows + (size_t)n * MLP_HIDDEN + j * 256 + lane * 4);
    }
    __device__ __forceinline__ void dot(const f32x4 (&hv)[4], float (&o)[3]) const {
#pragma unroll
        for (int n = 0; n < 3; ++n) {
            float acc = 0.0f;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc += hv[j][0] * w[n][j][0] + hv[j][1] * w[n][j][1] + hv[j][2] * w[n][j][2] + hv[j][3] * w[n][j][3];
            o[n] = wave_sum(acc);
        }
    }
};

// One wavefront per (joint, W consecutive windows): the joint's three decoder rows and the trajectory head's three are
// read once for its windows, each window's two hidden rows stream through two register sets (the next window loads
// while this one is reduced), 64-lane dot products, and the (x, y, z) written straight into the joint slot of the
// reference's reassembly.  Every wavefront recomputes the trajectory head of its windows - cheaper than a dependency
// between wavefronts.  W = 4 from 128 windows (22.7 against 27.9 us at 1024 windows: fewer, longer wavefronts), W = 1 below
// (3.5 us at one window; every further window of a wavefront adds 0.6 us to a launch that is one round of wavefronts anyway).
template <int DECODE_WINDOWS>
__device__ __forceinline__ void decode_body(const DecodeArgs &a) {
    const int lane = threadIdx.x & 63;
    const long long gw = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);   // global wavefront index
    const int per_win = a.has_pos ? a.J : 1;
    const long long b0 = gw / per_win * DECODE_WINDOWS;
    if (b0 >= a.B) return;
    // (single-launch forward: a dependency spin that gave up leaves garbage behind - make it loud)
    const bool poisoned = a.abort_flag != nullptr && __hip_atomic_load((gu32)a.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
    const float poison = poisoned ? __builtin_nanf("") : 0.0f;
    // ... and visible to the host: the handle's status word (pinned host memory), read by r3d_status
    if (poisoned && a.status != nullptr && gw == 0 && lane == 0) __hip_atomic_store((gu32)a.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const int jf = (int)(gw % per_win);              // joint in flat decoder order
    const int ts = a.nsrc - 1;
    int s = 0, o = 0;
    if (a.has_pos) {
        const int npos = a.has_trj ? a.nsrc - 1 : a.nsrc;
        for (int q = 0; q < npos; ++q)
            if (3 * jf >= a.first[q]) { s = q; o = 3 * jf - a.first[q]; }
    }
    f32x4 hv[2][4], ht[2][4];                        // two windows in flight: the next one loads while this one is reduced
    DecodeJoint rt, rj;
    if (a.has_trj) rt.load(a.w[ts], lane);
    if (a.has_pos) rj.load(a.w[s] + (size_t)o * MLP_HIDDEN, lane);
    auto load_window = [&](int i) {                  // (windows past the batch re-read the last one; nothing is stored for them)
        const long long b = b0 + i < a.B ? b0 + i : a.B - 1;
        if (a.has_trj) {
#pragma unroll
            for (int j = 0; j < 4; ++j) ht[i & 1][j] = gload4(a.h[ts] + b * MLP_HIDDEN + j * 256 + lane * 4);
        }
        if (a.has_pos) {
#pragma unroll
            for (int j = 0; j < 4; ++j) hv[i & 1][j] = gload4(a.h[s] + b * MLP_HIDDEN + j * 256 + lane * 4);
        }
    };
    load_window(0);
    // (the epilogue's constants requested before the reductions, not behind them)
    float bt[3] = {0.0f, 0.0f, 0.0f}, bj[3] = {0.0f, 0.0f, 0.0f};
    int e = 0;
    if (a.has_trj) {
#pragma unroll
        for (int n = 0; n < 3; ++n) bt[n] = a.bias[ts][n];
    }
    if (a.has_pos) {
        e = a.slot[a.first[s] + o];                  // (x, y, z) of a joint are consecutive in the output
#pragma unroll
        for (int n = 0; n < 3; ++n) bj[n] = a.bias[s][o + n];
    }
#pragma unroll
    for (int i = 0; i < DECODE_WINDOWS; ++i) {
        if (i + 1 < DECODE_WINDOWS) load_window(i + 1);
        const long long b = b0 + i;
        const bool live = b < a.B;
        float trj[3] = {0.0f, 0.0f, 0.0f};
        if (a.has_trj) {
            rt.dot(ht[i & 1], trj);
#pragma unroll
            for (int n = 0; n < 3; ++n) trj[n] += bt[n] + poison;
            if (live && lane == 0 && jf == 0) {
#pragma unroll
                for (int n = 0; n < 3; ++n) {
                    if (a.out_trj) a.out_trj[b * 3 + n] = trj[n];
                    if (!a.has_pos) a.out[b * 3 + n] = trj[n];
                }
            }
        }
        if (!a.has_pos) continue;
        float v[3];
        rj.dot(hv[i & 1], v);
        if (live && lane == 0) {
#pragma unroll
            for (int n = 0; n < 3; ++n) a.out[b * (a.J * 3) + e + n] = v[n] + bj[n] + trj[n] + poison;
        }
    }
}

extern "C" __global__ __launch_bounds__(256) void r3d_decode_f32(const DecodeArgs a) { decode_body<1>(a); }
extern "C" __global__ __launch_bounds__(256) void r3d_decode_w4_f32(const DecodeArgs a) { decode_body<4>(a); }

hipError_t launch_decode(const DecodeArgs &args, hipStream_t stream) {
    const int w = args.B >= 128 ? 4 : 1;
    const long long waves = (args.B + w - 1) / w * (args.has_pos ? args.J : 1);
    if (w == 4) r3d_decode_w4_f32<<<dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, stream>>>(args);
    else r3d_decode_f32<<<dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, stream>>>(args);
    return hipGetLastError();
}

}  // namespace r3d
