// Static load balancing of the persistent GEMM launches.
//
// A launch (one DAG level of the plan) consists of independent 32-row x 256-column "units", one
// per (problem, column block, row block); a unit of problem p costs ~ K_p/32 K-tiles of MFMA work.
// With one workgroup per CU the only scheduling freedom is how many units each workgroup gets, so
// the host cuts the unit sequence into `nwg` contiguous chunks of (nearly) equal cost.  Contiguity
// keeps a workgroup - and, through the XCD-aware chunk order in the kernel, an XCD - on one
// problem's weights.  A chunk is then emitted as tiles of at most 6 units (BM <= 192 rows).
// This replaces the hardware dispatcher's "first free slot" placement, which left the second
// round of 128x128 tiles one-third occupied (profiles/r01_v0/pmc_table.txt).
#include <algorithm>
#include <cmath>

#include "r3d_internal.hpp"

namespace r3d {

Schedule::~Schedule() {
    if (d_tiles) (void)hipFree(d_tiles);
    if (d_wgoff) (void)hipFree(d_wgoff);
}

Plan::~Plan() {
    for (auto &kv : schedules) delete kv.second;
}

int device_cu_count() {
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (!cus[dev]) {
        hipDeviceProp_t prop;
        cus[dev] = hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 256;
    }
    return cus[dev];
}

namespace {

struct Segment {
    int prob;      // index inside the launch
    int col0;
    int units;     // 32-row units
    int M;
    double unit_cost;
    int max_units; // largest tile of this problem
};

// cycles of one SIMD for a unit: two wavefronts share the pipe, 16 MFMAs x 64 cycles each per
// K tile, plus staging/barrier overhead per K tile and a share of the tile prologue/epilogue
double unit_cycles(int nk) { return nk * (2048.0 + 120.0) + 1500.0; }

}  // namespace

void schedule_stage(const std::vector<SchedProb> &probs, int nwg, int max_units, std::vector<int4> &tiles,
                    std::vector<int> &wgoff, StageSchedule &out) {
    // Launches that cannot give every CU a unit switch to split-K tiles (128 columns wide, the two K
    // halves computed by different wavefronts of the workgroup): twice the units, half the K loop.
    long long units256 = 0;
    bool all_plain = true;
    for (const auto &p : probs) {
        units256 += (long long)((p.M + 31) / 32) * ((p.N + 255) / 256);
        all_plain = all_plain && p.plain;
    }
    const int ks = (all_plain && units256 * 2 <= (long long)nwg) ? 2 : 1;   // only pays when the doubled units still fit one round
    out.ks = ks;
    if (ks == 2) max_units = 2;
    const int bn = 256 / ks;
    std::vector<Segment> segs;
    double total = 0;
    for (int i = 0; i < (int)probs.size(); ++i) {
        const int units = (probs[i].M + 31) / 32;
        const double cost = unit_cycles((probs[i].nk + ks - 1) / ks);
        for (int c0 = 0; c0 < probs[i].N; c0 += bn) {
            segs.push_back({i, c0, units, probs[i].M, cost, probs[i].max_units > 0 ? std::min(probs[i].max_units, max_units) : max_units});
            total += units * cost;
        }
    }
    long long total_units = 0;
    for (auto &s : segs) total_units += s.units;
    const int grid = (int)std::min<long long>(nwg, std::max<long long>(total_units, 1));
    out.nwg = grid;
    out.tiles_off = tiles.size();
    out.wgoff_off = wgoff.size();
    // sweep: chunk c ends where the running cost crosses (c+1) * total / grid
    const double target = total / grid;
    double run = 0, worst = 0, chunk_cost = 0;
    int chunk = 0;
    wgoff.push_back((int)(tiles.size() - out.tiles_off));
    auto close_chunk = [&]() {
        worst = std::max(worst, chunk_cost);
        chunk_cost = 0;
        ++chunk;
        wgoff.push_back((int)(tiles.size() - out.tiles_off));
    };
    for (const Segment &s : segs) {
        int u = 0;
        while (u < s.units) {
            // how many units of this segment still fit in the current chunk
            const double room = (chunk + 1) * target - run;
            int take = (int)std::floor(room / s.unit_cost + 0.5);
            if (chunk == grid - 1) take = s.units - u;          // last chunk absorbs the remainder
            take = std::max(take, chunk_cost == 0 ? 1 : 0);
            take = std::min(take, s.units - u);
            if (take > 0) {
                // emit `take` units as evenly sized tiles of <= max_units units
                const int nt = (take + s.max_units - 1) / s.max_units;
                int done = 0;
                for (int k = 0; k < nt; ++k) {
                    const int sz = (take - done + (nt - k) - 1) / (nt - k);
                    tiles.push_back(make_int4(s.prob | (sz << 8), (u + done) * 32, s.col0, 0));
                    done += sz;
                }
                u += take;
                run += take * s.unit_cost;
                chunk_cost += take * s.unit_cost;
            }
            if (chunk < grid - 1 && run >= (chunk + 1) * target - 0.5 * s.unit_cost) close_chunk();
        }
    }
    while (chunk < grid) close_chunk();
    out.ntiles = (int)(tiles.size() - out.tiles_off);
    out.imbalance = worst / std::max(target, 1.0);
}

static void build_stage(const Plan *pl, const std::vector<int> &st, int64_t B, int nwg, std::vector<int4> &tiles,
                        std::vector<int> &wgoff, StageSchedule &out) {
    std::vector<SchedProb> probs;
    double flops = 0, bytes = 0;
    for (int i = 0; i < (int)st.size(); ++i) {
        const ProbSpec &q = pl->probs[st[i]];
        const Layer &L = pl->m[q.model]->layers[q.layer];
        const int M = (int)(B * q.rows_per_window);
        // fused-prologue tiles hold the whole encoded operand in 64 KiB of LDS: rows * (K + 4) floats
        const int enc_cap = q.enc_lut >= 0 ? std::max(1, std::min(3, (64 * 1024) / ((L.Kpad + 4) * 4 * 32))) : 0;
        probs.push_back({M, L.N, L.Kpad / BK, q.enc_lut < 0 && q.nseg == 1, enc_cap});
        flops += q.flops_per_window * (double)B;
        bytes += 4.0 * ((double)M * L.K + (double)L.N * L.K + (double)M * L.N * (q.res_buf >= 0 ? 2.0 : 1.0));
    }
    bool enc = false;
    for (int id : st) enc = enc || pl->probs[id].enc_lut >= 0;
    // the fused-prologue kernel runs two workgroups per CU (one encodes while the other multiplies)
    schedule_stage(probs, enc ? 2 * nwg : nwg, 6, tiles, wgoff, out);
    out.flops = flops;
    out.bytes = bytes;
}

Schedule *schedule_get(Plan *pl, int64_t B, int nwg) {
    auto it = pl->schedules.find(B);
    if (it != pl->schedules.end()) return it->second;
    // bound the cache: evict the oldest batch size (the caller synchronises nothing here; a schedule
    // is only freed after 16 newer batch sizes were used, by which time its launches have long retired)
    if (pl->schedule_lru.size() >= 16) {
        const int64_t old = pl->schedule_lru.front();
        pl->schedule_lru.erase(pl->schedule_lru.begin());
        (void)hipDeviceSynchronize();
        delete pl->schedules[old];
        pl->schedules.erase(old);
    }
    Schedule *s = new Schedule();
    s->B = B;
    std::vector<int4> tiles;
    std::vector<int> wgoff;
    for (const auto &st : pl->stages) {
        StageSchedule ss{};
        build_stage(pl, st, B, nwg, tiles, wgoff, ss);
        s->stages.push_back(ss);
    }
    hipError_t e;
    if ((e = hipMalloc((void **)&s->d_tiles, std::max<size_t>(tiles.size(), 1) * sizeof(int4))) != hipSuccess ||
        (e = hipMalloc((void **)&s->d_wgoff, std::max<size_t>(wgoff.size(), 1) * sizeof(int))) != hipSuccess ||
        (e = hipMemcpy(s->d_tiles, tiles.data(), tiles.size() * sizeof(int4), hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(s->d_wgoff, wgoff.data(), wgoff.size() * sizeof(int), hipMemcpyHostToDevice)) != hipSuccess) {
        hip_fail(e, "schedule upload");
        delete s;
        return nullptr;
    }
    pl->schedules[B] = s;
    pl->schedule_lru.push_back(B);
    return s;
}

}  // namespace r3d
