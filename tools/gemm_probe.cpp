// Stand-alone probe of the persistent GEMM kernel (development tool, not part of the library):
// runs one launch shaped like a DAG level of the RF-243 plan on synthetic data and prints its
// duration, rate and (with -DR3D_TIMING) per-K-tile phase stamps of the first workgroups.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude -Iray3d_amd/csrc [-DR3D_TIMING] \
//         tools/gemm_probe.cpp ray3d_amd/csrc/r3d_kernels.hip ray3d_amd/csrc/r3d_schedule.cpp \
//         ray3d_amd/csrc/r3d_model.cpp -o /tmp/gemm_probe
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

#include "r3d_internal.hpp"

using namespace r3d;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

int main(int argc, char **argv) {
    // usage: gemm_probe nprob M N K [reps]
    const int nprob = argc > 1 ? atoi(argv[1]) : 6;
    const int M = argc > 2 ? atoi(argv[2]) : 6912;
    const int N = argc > 3 ? atoi(argv[3]) : 256;
    const int K = argc > 4 ? atoi(argv[4]) : 768;
    const int reps = argc > 5 ? atoi(argv[5]) : 20;
    const int enc = argc > 6 ? atoi(argv[6]) : 0;   // 1: fused-prologue operand (M must be a multiple of 81)
    const int Npad = round_up(N, 256);
    std::vector<float> hA((size_t)M * K), hW((size_t)Npad * K, 0.f), hb(Npad, 0.f);
    for (size_t i = 0; i < hA.size(); ++i) hA[i] = (float)((i * 2654435761u) % 2001) / 1000.f - 1.f;
    std::vector<float> hWr((size_t)Npad * K, 0.f);   // row-major reference copy; hW is fragment-ordered
    for (size_t i = 0; i < (size_t)N * K; ++i) hWr[i] = (float)((i * 40503u) % 2001) / 1000.f - 1.f;
    for (int o = 0; o < N; ++o)
        for (int k = 0; k < K; ++k) hW[frag_index(o, k, K / BK)] = hWr[(size_t)o * K + k];
    if (getenv("PROBE_ZERO")) {      // all-zero operands: the same instruction stream at the clock a quiet datapath allows (DVFS)
        std::fill(hA.begin(), hA.end(), 0.f);
        std::fill(hW.begin(), hW.end(), 0.f);
        std::fill(hWr.begin(), hWr.end(), 0.f);
    }
    LaunchArgs la;
    memset(&la, 0, sizeof la);
    la.nprob = nprob;
    std::vector<float *> dA(nprob), dW(nprob), dC(nprob);
    float *dbias;
    CK(hipMalloc((void **)&dbias, Npad * 4));
    CK(hipMemcpy(dbias, hb.data(), Npad * 4, hipMemcpyHostToDevice));
    std::vector<SchedProb> sp;
    // fused-prologue inputs: windows of 243 frames x 51 floats, LUT shaped like a temporal branch's
    float *dx = nullptr; int *dlut = nullptr;
    std::vector<float> hx; std::vector<int> hlut(K + K / 4, 0);
    if (enc) {
        const int nwin = M / 81;
        hx.resize((size_t)nwin * 243 * 51);
        for (size_t i = 0; i < hx.size(); ++i) hx[i] = (float)((i * 2246822519u) % 2001) / 1000.f - 1.f;
        for (int k = 0; k < K; ++k) {   // base-grouped columns like the real first layers: 3/4 row-relative, 1/4 current-frame relative
            const bool cur = k >= (K * 3 / 4) / 4 * 4;
            hlut[k] = cur ? ((k * 7) % 51) * 4 : ((k % 3) * 51 + (k * 7) % 51) * 4;
            if (cur) hlut[K + k / 4] = 1;
        }
        CK(hipMalloc((void **)&dx, hx.size() * 4));
        CK(hipMalloc((void **)&dlut, hlut.size() * 4));
        CK(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dlut, hlut.data(), hlut.size() * 4, hipMemcpyHostToDevice));
    }
    for (int i = 0; i < nprob; ++i) {
        CK(hipMalloc((void **)&dA[i], hA.size() * 4));
        CK(hipMalloc((void **)&dW[i], hW.size() * 4));
        CK(hipMalloc((void **)&dC[i], (size_t)M * N * 4));
        CK(hipMemcpy(dA[i], hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dW[i], hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
        GemmProb &g = la.p[i];
        for (int s = 0; s < MAX_SEG; ++s) { g.a[s] = dA[i]; g.lda[s] = K; g.kend[s] = 0x7fffffff; }
        g.w = dW[getenv("PROBE_SHARE_W") ? 0 : i]; g.bias = dbias;   // PROBE_SHARE_W: every problem streams problem 0's weights (L2-resident)
        if (getenv("PROBE_SHARE_A")) for (int s = 0; s < MAX_SEG; ++s) g.a[s] = dA[0];
        g.res = nullptr; g.c = dC[i]; g.ldr = 0; g.ldc = N;
        g.M = M; g.N = N; g.K = K; g.slope = 0.2f;
        if (getenv("PROBE_B3")) g.wb3 = dW[i];   // timing only: the bf16x3 tile on weights that are not in its operand order
        if (enc) { g.lut = dlut; g.x = dx; g.enc_ws = 243 * 51; g.enc_rows = 81; g.enc_jf = 51; g.enc_cur = 81 * 51; g.enc_step = 3; g.enc_bytes = (unsigned)(hx.size() * 4); }
        sp.push_back({M, N, K / BK, enc ? 1 : 4, enc ? std::max(1, std::min(3, (64 * 1024) / ((K + 4) * 4 * 32))) : 0});
    }
    std::vector<int4> tiles;
    std::vector<int> wgoff;
    StageSchedule ss{};
    const int nwg = getenv("PROBE_NWG") ? atoi(getenv("PROBE_NWG")) : device_cu_count();
    schedule_stage(sp, enc ? 2 * nwg : nwg, 6, tiles, wgoff, ss, enc != 0);
    auto launch = [&]() { return launch_gemm_stage(la, ss.nwg, ss.kind, false, 0); };
    int4 *dt; int *dwg; long long *ddbg;
    CK(hipMalloc((void **)&dt, tiles.size() * sizeof(int4)));
    CK(hipMalloc((void **)&dwg, wgoff.size() * sizeof(int)));
    CK(hipMalloc((void **)&ddbg, (1024 + 4 * 1024) * 8 + 65536));
    CK(hipMemset(ddbg, 0, (1024 + 4 * 1024) * 8 + 65536));
    CK(hipMemcpy(dt, tiles.data(), tiles.size() * sizeof(int4), hipMemcpyHostToDevice));
    CK(hipMemcpy(dwg, wgoff.data(), wgoff.size() * sizeof(int), hipMemcpyHostToDevice));
    la.tiles = dt; la.wg_off = dwg; la.dbg = ddbg; la.ks = ss.ks;
    printf("grid %d tiles %d ks %d kind %d imbalance %.3f (CUs %d)\n", ss.nwg, ss.ntiles, ss.ks, ss.kind, ss.imbalance, nwg);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) CK(launch());
    CK(hipDeviceSynchronize());
    float best = 1e9, sum = 0;
    if (getenv("PROBE_NOSYNC")) {   // back to back, no host synchronisation between the launches: the clock of a busy chip
        for (int i = 0; i < 200; ++i) CK(launch());
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < reps; ++i) CK(launch());
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("back to back: %.1f us per launch\n", ms * 1e3 / reps);
    }
    for (int i = 0; i < (getenv("PROBE_NOSYNC") ? 0 : reps); ++i) {
        CK(hipEventRecord(e0, 0));
        CK(launch());
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best; sum += ms;
    }
    const double flops = 2.0 * nprob * (double)M * N * K;
    printf("nprob %d M %d N %d K %d: best %.1f us avg %.1f us -> %.1f TFLOP/s (best)\n", nprob, M, N, K, best * 1e3,
           sum / reps * 1e3, flops / (best * 1e-3) / 1e12);
    // correctness spot check of problem 0, a few entries
    std::vector<float> hC((size_t)M * N);
    CK(hipMemcpy(hC.data(), dC[0], hC.size() * 4, hipMemcpyDeviceToHost));
    double maxerr = 0;
    for (int t = 0; t < 64; ++t) {
        const int r = (t * 7919) % M, c = (t * 104729) % N;
        double acc = 0;
        for (int k = 0; k < K; ++k) {
            double a = hA[(size_t)r * K + k];
            if (enc) {
                const int win = r / 81, t3 = r % 81;
                const size_t first = (size_t)win * 243 * 51 + (size_t)t3 * 3 * 51, cur = (size_t)win * 243 * 51 + 81 * 51;
                a = hx[(hlut[K + k / 4] ? cur : first) + hlut[k] / 4];
            }
            acc += a * hWr[(size_t)c * K + k];
        }
        acc = acc > 0 ? acc : 0.2 * acc;
        const double err = fabs(acc - hC[(size_t)r * N + c]) / (1.0 + fabs(acc));
        maxerr = err > maxerr ? err : maxerr;
    }
    printf("spot-check max rel err %.2e\n", maxerr);
    if (getenv("PROBE_FULLCHECK")) {
        // per 32-row unit: how many of 8 sampled entries are wrong
        const int units = (M + 31) / 32;
        printf("units with errors (unit: bad/8):");
        int nbad = 0;
        for (int u = 0; u < units; ++u) {
            int bad = 0;
            for (int t = 0; t < 8; ++t) {
                const int r = std::min(M - 1, u * 32 + (t * 5) % 32), c = (t * 37 + u) % N;
                double acc = 0;
                for (int k = 0; k < K; ++k) {
                    double a = hA[(size_t)r * K + k];
                    if (enc) {
                        const int win = r / 81, t3 = r % 81;
                        const size_t first = (size_t)win * 243 * 51 + (size_t)t3 * 3 * 51, cur = (size_t)win * 243 * 51 + 81 * 51;
                        a = hx[(hlut[K + k / 4] ? cur : first) + hlut[k] / 4];
                    }
                    acc += a * hWr[(size_t)c * K + k];
                }
                acc = acc > 0 ? acc : 0.2 * acc;
                if (fabs(acc - hC[(size_t)r * N + c]) > 1e-3 * (1 + fabs(acc))) ++bad;
            }
            if (bad) { if (nbad < 40) printf(" %d:%d", u, bad); ++nbad; }
        }
        printf("  (%d of %d units bad)\n", nbad, units);
        for (int t = 0; t < 12 && t < (int)tiles.size(); ++t) printf("tile %d: prob %d mi %d row0 %d col0 %d\n", t, tiles[t].x & 255, tiles[t].x >> 8, tiles[t].y, tiles[t].z);
        if (getenv("PROBE_DUMPROW")) {
            const int r = atoi(getenv("PROBE_DUMPROW"));
            for (int rr = r; rr < r + 3; ++rr) { printf("row %d got:", rr);
            for (int c = 0; c < 6; ++c) printf(" %9.4f", hC[(size_t)rr * N + c]); printf("\n"); }
            printf("\nexpected for rows r-2..r+2 (cols 0..5):\n");
            for (int rr = 0; rr < M; ++rr) {
                if (!(rr % 32 < 3)) continue;
                printf("  row %4d:", rr);
                for (int c = 0; c < 6; ++c) {
                    double acc = 0;
                    for (int k = 0; k < K; ++k) {
                        const int win = rr / 81, t3 = rr % 81;
                        const size_t first = (size_t)win * 243 * 51 + (size_t)t3 * 3 * 51, cur = (size_t)win * 243 * 51 + 81 * 51;
                        acc += (double)hx[(hlut[K + k / 4] ? cur : first) + hlut[k] / 4] * hWr[(size_t)c * K + k];
                    }
                    printf(" %9.4f", acc > 0 ? acc : 0.2 * acc);
                }
                printf("\n");
            }
        }
    }
#ifdef R3D_TIMING
    {
        std::vector<long long> hw(4 * 1024);
        CK(hipMemcpy(hw.data(), ddbg + 1024, hw.size() * 8, hipMemcpyDeviceToHost));
        long long w0 = 1LL << 62, w1 = 0;
        for (int w = 0; w < ss.nwg; ++w) { w0 = std::min(w0, hw[w * 4 + 2]); w1 = std::max(w1, hw[w * 4 + 3]); }
        printf("wall span %lld ticks (100 MHz => %.1f us)\n", w1 - w0, (w1 - w0) / 100.0);
        {
            std::vector<long long> ht(16 * 64);
            CK(hipMemcpy(ht.data(), ddbg + 6144, ht.size() * 8, hipMemcpyDeviceToHost));
            printf("tile phases (us from launch start): wg tile | entry  ready  loop_end  reduced  stored | prologue loop reduce store\n");
            for (int w = 0; w < 16 && w < ss.nwg; ++w)
                for (int t = 0; t < 8 && t < wgoff[w + 1] - wgoff[w]; ++t) {
                    const long long *q = &ht[w * 64 + t * 8];
                    const int4 td = tiles[wgoff[w] + t];
                    printf("  wg %2d tile %d (p%d mi%d ks%d): %7.2f %7.2f %7.2f %7.2f %7.2f | %6.2f %6.2f %6.2f %6.2f\n", w, t, td.x & 255, td.x >> 8, td.w,
                           (q[0] - w0) / 100.0, (q[1] - w0) / 100.0, (q[2] - w0) / 100.0, (q[3] - w0) / 100.0, (q[4] - w0) / 100.0,
                           (q[1] - q[0]) / 100.0, (q[2] - q[1]) / 100.0, (q[3] - q[2]) / 100.0, (q[4] - q[3]) / 100.0);
                }
        }
        printf("chunk: tiles units | start_us end_us | cycles | eff GHz\n");
        for (int w = 0; w < ss.nwg; w += (w < 8 || w > ss.nwg - 9) ? 1 : 13) {
            int units = 0;
            for (int t = wgoff[w]; t < wgoff[w + 1]; ++t) units += tiles[t].x >> 8;
            const double us0 = (hw[w * 4 + 2] - w0) / 100.0, us1 = (hw[w * 4 + 3] - w0) / 100.0;
            printf("  %3d: %d %2d | %7.1f %7.1f | %7lld | %.2f\n", w, wgoff[w + 1] - wgoff[w], units, us0, us1,
                   hw[w * 4 + 1] - hw[w * 4 + 0], (hw[w * 4 + 1] - hw[w * 4 + 0]) / (us1 - us0) / 1e3);
        }
    }
#endif
    return 0;
}
