// How long does one dependency hop take between two workgroups - write-through store on one CU, polling load on another -
// when they sit on the same XCD and when they do not?  (XCD-local chains.)
//   hipcc -O3 --offload-arch=gfx950 tools/hop_probe.cpp -o tools/hop_probe.bin && tools/hop_probe.bin
// 256 workgroups of 512 threads (one per CU, as the forward kernel); workgroups A and B play ping-pong on two words with
// the forward kernel's instructions (4-byte agent-scope relaxed atomics = sc1 store / sc1 load) for R rounds; A reports
// wall-clock ticks (100 MHz).  Every workgroup records the XCC it ran on.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned __attribute__((address_space(1))) *gu32;
__device__ __forceinline__ long long wall() { return __builtin_readcyclecounter() * 0 + (long long)__builtin_amdgcn_s_memrealtime(); }
__global__ __launch_bounds__(512) void hop(unsigned *flags, int a, int b, int rounds, long long *out, int *xcc, int mode) {
    if (threadIdx.x == 0) {
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        xcc[blockIdx.x] = (int)(id & 0xf);
    }
    if ((int)blockIdx.x != a && (int)blockIdx.x != b) return;
    if (threadIdx.x != 0) return;
    gu32 f1 = (gu32)flags, f2 = (gu32)(flags + 64);
    if ((int)blockIdx.x == a) {
        const long long t0 = wall();
        for (int i = 1; i <= rounds; ++i) {
            if (mode == 0) __hip_atomic_store(f1, (unsigned)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else __hip_atomic_fetch_add(f1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(f2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned)i) {}
        }
        out[0] = wall() - t0;
    } else {
        for (int i = 1; i <= rounds; ++i) {
            while (__hip_atomic_load(f1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned)i) {}
            if (mode == 0) __hip_atomic_store(f2, (unsigned)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else __hip_atomic_fetch_add(f2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
int main() {
    unsigned *flags; long long *out; int *xcc;
    hipMalloc(&flags, 1024); hipMalloc(&out, 64); hipMalloc(&xcc, 256 * 4);
    const int rounds = 2000;
    std::vector<int> hx(256);
    const int pairs[][2] = {{0, 8}, {0, 16}, {0, 1}, {0, 4}, {3, 11}, {3, 4}, {0, 248}, {0, 255}};
    for (int mode = 0; mode < 2; ++mode)
        for (auto &p : pairs) {
            hipMemset(flags, 0, 1024);
            hipLaunchKernelGGL(hop, dim3(256), dim3(512), 0, 0, flags, p[0], p[1], rounds, out, xcc, mode);
            hipDeviceSynchronize();
            long long t;
            hipMemcpy(&t, out, 8, hipMemcpyDeviceToHost);
            hipMemcpy(hx.data(), xcc, 256 * 4, hipMemcpyDeviceToHost);
            printf("%s  workgroups %3d (XCC %d) <-> %3d (XCC %d): %.3f us per hop\n", mode ? "atomic add" : "store     ", p[0], hx[p[0]], p[1], hx[p[1]],
                   t / 100.0 / rounds / 2.0);
        }
    return 0;
}
