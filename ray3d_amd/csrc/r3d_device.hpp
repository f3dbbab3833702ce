// Device-side helpers shared by the kernel files (r3d_kernels.hip: 512-thread workgroups, one per CU; r3d_kernels4.hip:
// 256-thread workgroups, two per CU): typed loads / stores of activations, the tile hand-off protocol, the UV -> ray
// encoding.  gfx950 only.
#pragma once

#include <hip/hip_runtime.h>

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
    if ((int)threadIdx.x < units) __hip_atomic_fetch_add(cnt + sig_base + threadIdx.x, (unsigned)sig_add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// LeakyReLU for slopes in (0, 1] (0.2, 0.01; 1 = linear layer): max(v, slope v) - a multiply and a max instead of
// multiply, compare, select
__device__ __forceinline__ float lrelu(const float v, const float slope) { return __builtin_fmaxf(v, v * slope); }


typedef const LaunchArgs __attribute__((address_space(4))) *LaunchArgsPtr;
typedef const GemmProb __attribute__((address_space(4))) &ProbRef;

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


constexpr int LDS_LD = BK + 4;                       // 36 floats = 144 B per staged row

constexpr int FL_LUT_INTS = 320;                           // first-layer tables in LDS: K0 + K0/4 ints, K0 <= 256

}  // namespace r3d
