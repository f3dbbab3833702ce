// Where does the hardware put the workgroups of a 512 x 256-thread launch (two per CU by LDS and registers)?
// Every workgroup records XCC_ID and HW_ID and stays resident until all have started, so the map is of co-resident groups.
// build: hipcc -O3 --offload-arch=gfx950 tools/wgmap_probe.cpp -o tools/wgmap_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>

__global__ __launch_bounds__(256, 2) void probe(unsigned *out, unsigned *arrived, int n) {
    extern __shared__ float lds[];
    if (threadIdx.x == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        out[blockIdx.x * 4 + 0] = hw;
        out[blockIdx.x * 4 + 1] = xcc;
        out[blockIdx.x * 4 + 2] = (unsigned)wall_clock64();
        lds[0] = (float)hw;
        __hip_atomic_fetch_add(arrived, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        long long t0 = wall_clock64();
        while (__hip_atomic_load(arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)n && wall_clock64() - t0 < 20000000LL) __builtin_amdgcn_s_sleep(8);
        out[blockIdx.x * 4 + 3] = __hip_atomic_load(arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
}

int main() {
    const int lds = 80 * 1024;
    hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    int per_cu = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)probe, 256, lds);
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    printf("CUs %d, occupancy %d workgroups per CU at 256 threads + %d B LDS\n", prop.multiProcessorCount, per_cu, lds);
    for (int rep = 0; rep < 3; ++rep) {
        const int n = 2 * prop.multiProcessorCount;
        unsigned *d, *arr;
        hipMalloc(&d, n * 16);
        hipMalloc(&arr, 4);
        hipMemset(arr, 0, 4);
        probe<<<n, 256, lds>>>(d, arr, n);
        hipDeviceSynchronize();
        std::vector<unsigned> h(n * 4);
        hipMemcpy(h.data(), d, n * 16, hipMemcpyDeviceToHost);
        std::map<unsigned long long, std::vector<int>> by_cu;
        int all = 0;
        for (int b = 0; b < n; ++b) {
            const unsigned hw = h[b * 4], xcc = h[b * 4 + 1] & 0xf;
            // HW_ID (gfx9): wave_id 3:0, simd_id 5:4, pipe_id 7:6, cu_id 11:8, sh_id 12, se_id 15:13 (gfx90a+: 3 bits)
            const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
            by_cu[((unsigned long long)xcc << 16) | (se << 8) | (sh << 4) | cu].push_back(b);
            all += h[b * 4 + 3] >= (unsigned)n;
        }
        printf("rep %d: %d workgroups, %d saw all %d arrive (co-resident), %zu distinct CUs\n", rep, n, all, n, by_cu.size());
        if (rep == 2) {
            std::map<int, int> hist;
            for (auto &kv : by_cu) hist[(int)kv.second.size()]++;
            for (auto &kv : hist) printf("  CUs holding %d workgroups: %d\n", kv.first, kv.second);
            int shown = 0;
            for (auto &kv : by_cu) {
                if (shown++ >= 40) break;
                printf("  xcc %llu se %llu sh %llu cu %2llu:", kv.first >> 16, (kv.first >> 8) & 0xff, (kv.first >> 4) & 0xf, kv.first & 0xf);
                for (int b : kv.second) printf(" wg %d (xcd-slot %d, idx %d)", b, b & 7, b >> 3);
                printf("\n");
            }
            // is wg b's XCC == b % 8 ?
            int match = 0;
            for (int b = 0; b < n; ++b) match += (int)(h[b * 4 + 1] & 0xf) == (b & 7);
            printf("  workgroups with XCC_ID == blockIdx %% 8: %d of %d\n", match, n);
            // partner pattern: difference of the two workgroup indices sharing a CU
            std::map<int, int> diff;
            for (auto &kv : by_cu) if (kv.second.size() == 2) diff[kv.second[1] - kv.second[0]]++;
            for (auto &kv : diff) printf("  partner index difference %d: %d CUs\n", kv.first, kv.second);
        }
        hipFree(d);
        hipFree(arr);
    }
    return 0;
}
