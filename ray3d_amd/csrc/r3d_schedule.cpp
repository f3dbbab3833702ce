// Static load balancing of the persistent GEMM launches.
//
// A launch (one DAG level of the plan) consists of independent 32-row x 256-column "units", one
// per (problem, column block, row block); a unit of problem p costs ~ K_p/32 K-tiles of MFMA work.
// Problems whose units are coarse against a workgroup's share (the M = B MLP layers, the top of the
// conv pyramid) are cut further along K: 32 x 128 or 32 x 64 split-K units (r3d_kernels.hip, KS).
// With one workgroup per CU the only scheduling freedom is which units each workgroup gets: the host packs
// them into at most `nwg` chunks under the smallest feasible chunk budget (first-fit decreasing inside a
// binary search; whole units first - consecutive units of a column block in consecutive workgroups, so a
// workgroup and, through the XCD-aware chunk order in the kernel, an XCD stays on one problem's weights -
// then the units that found no room, cut along K).  A chunk is emitted as tiles of at most 6 units
// (BM <= 192 rows; 4 for fused pairs, 1 for the fused first level).  This replaces the hardware
// dispatcher's "first free slot" placement, which left the second round of 128x128 tiles one-third
// occupied (profiles/r01_v0/pmc_table.txt).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <map>

#include "r3d_internal.hpp"

namespace r3d {

Schedule::~Schedule() {
    if (d_tiles) (void)hipFree(d_tiles);
    if (d_wgoff) (void)hipFree(d_wgoff);
    if (fwd.d_tiles) (void)hipFree(fwd.d_tiles);
    if (fwd.d_wgoff) (void)hipFree(fwd.d_wgoff);
    if (fwd.d_ctrl) (void)hipFree(fwd.d_ctrl);
    if (fwd.d_act) (void)hipFree(fwd.d_act);
    for (int i = 0; i < 4; ++i) {
        if (fwd.d_rel[i]) (void)hipFree(fwd.d_rel[i]);
        if (fwd.d_tags[i]) (void)hipFree(fwd.d_tags[i]);
    }
    if (d_frame_tiles) (void)hipFree(d_frame_tiles);
    if (d_frame_wgoff) (void)hipFree(d_frame_wgoff);
}

bool forward_single_launch() {
    static const bool on = !env_on("R3D_STAGED");
    return on;
}

// development aid (R3D_FWD_DUMP=<file>): the problems with their producers and every workgroup's tiles
static void dump_fwd(const Plan *pl, int64_t B, int grid, const std::vector<int> &out_tiles, const std::vector<int> &out_wgoff) {
    const int np = (int)pl->probs.size();
    std::vector<int> M(np);
    for (int i = 0; i < np; ++i) M[i] = (int)(B * pl->probs[i].rows_per_window);
    if (const char *dump = hook_env("R3D_FWD_DUMP")) {             // development aid: problems (with producers) and every workgroup's tiles
        if (FILE *f = fopen(dump, "w")) {
            for (int i = 0; i < np; ++i) {
                const ProbSpec &q = pl->probs[i];
                const Model *mm = pl->m[q.model];
                fprintf(f, "P %d %s rpw %d M %d N %d K %d K2 %d K3 %d deps", i, mm->layers[q.layer].weight_key.c_str(), q.rows_per_window, M[i],
                        mm->layers[q.layer].N, mm->layers[q.layer].Kpad, q.layer2 >= 0 ? mm->layers[q.layer2].Kpad : 0,
                        q.layer3 >= 0 ? mm->layers[q.layer3].Kpad : 0);
                for (int dpi : q.deps) fprintf(f, " %d", dpi);
                fprintf(f, "\n");
            }
            for (int b = 0; b < grid; ++b)
                for (int t = out_wgoff[b]; t < out_wgoff[b + 1]; ++t) {
                    const int *d = &out_tiles[(size_t)t * FWD_TILE_INT4 * 4];
                    fprintf(f, "T %d %d %d %d %d %d %d\n", b, t, d[0] & 0xff, d[0] >> 8, d[1], d[2], d[3]);
                }
            fclose(f);
        }
    }
}

static int64_t units_of(int64_t B, const ProbSpec &q) { return (B * q.rows_per_window + 31) / 32; }
static SchedProb sched_prob_of(const Plan *pl, const ProbSpec &q, int64_t B);

size_t fwd_ctrl_bytes(const Plan *pl, int64_t B) {
    int64_t ncnt = 0;
    for (const auto &q : pl->probs) ncnt += units_of(B, q);
    return ((size_t)(ncnt + 4) * sizeof(unsigned) + 255) / 256 * 256 + pl->probs.size() * sizeof(GemmProb) + 256;
}

// The single-launch form of a schedule: every workgroup's chunks of all launches concatenated in launch order, each tile
// carrying the ready counters it raises and the producer units it waits for (r3d_internal.hpp, FWD_TILE_INT4).  Launch
// order is a topological order of the tiles, and a workgroup's list follows it, so the tile that is first in that order
// among the unfinished ones can always run: its producers are finished and its workgroup has nothing else in front of
// it - no waiting cycle exists as long as all workgroups are resident (grid <= CU count).
// Returns false when the plan cannot run this way (a launch of r3d_gemm_enc_f32, too many producers for a descriptor).
bool schedule_build_fwd(const Plan *pl, int64_t B, int nwg, const std::vector<std::vector<int>> &levels, const std::vector<StageSchedule> &stages,
                        const std::vector<int4> &tiles, const std::vector<int> &wgoff, Schedule::Fwd &fw, std::vector<int> &out_tiles,
                        std::vector<int> &out_wgoff) {
    const int np = (int)pl->probs.size();
    if (np > 255) return false;
    for (const auto &ss : stages)
        if (ss.kind != STAGE_BIG) return false;
    for (const Model *m : pl->m)
        if (m && m->cfg.dense) return false;      // (the dense ablation's overlapping operand rows are not row-local per 32-row unit)
    fw.nprob = np;
    fw.cnt_base.assign(np, 0);
    std::vector<int> gcols(np), M(np);
    int ncnt = 0;
    for (int i = 0; i < np; ++i) {
        const ProbSpec &q = pl->probs[i];
        fw.cnt_base[i] = ncnt;
        ncnt += (int)units_of(B, q);
        gcols[i] = (pl->m[q.model]->layers[q.layer].N + COL_GRANULE - 1) / COL_GRANULE;
        M[i] = (int)(B * q.rows_per_window);
        if (q.enc_lut >= 0) fw.uses_gather = true;
    }
    fw.ncnt = ncnt;
    // problems that run as GEMV / latency tiles, and the problems all of whose consumers do (or that only the decoder kernel
    // reads): in poll mode nobody waits for their ready counters, and their tiles carry FWD_TILE_NOSIGNAL (descriptor int 7)
    std::vector<char> narrow(np, 0), quiet(np, 1);
    for (size_t si = 0; si < stages.size(); ++si)
        for (int t = 0; t < stages[si].ntiles; ++t) {
            const int4 &tl = tiles[stages[si].tiles_off + t];
            if (tile_is_narrow(tl.w)) narrow[levels[si][tl.x & 0xff] & ~STAGE_SPILL_IN] = 1;
        }
    for (int i = 0; i < np; ++i)
        for (int dp : pl->probs[i].deps)
            if (!narrow[i]) quiet[dp] = 0;
    std::vector<std::vector<int>> lists(nwg);     // per workgroup: tile descriptors, 24 ints each
    fw.flops = fw.bytes = 0;
    for (size_t si = 0; si < stages.size(); ++si) {
        const StageSchedule &ss = stages[si];
        const auto &st = levels[si];
        fw.flops += ss.flops;
        fw.bytes += ss.bytes;
        if (ss.nwg > nwg) return false;
        const int n = ss.nwg, qd = n >> 3, r = n & 7;
        for (int b = 0; b < n; ++b) {
            // the chunk workgroup b takes in the staged launch of this level (XCD-aware order, r3d_kernels.hip)
            const int xcd = b & 7, idx = b >> 3;
            const int c = (xcd < r ? xcd * (qd + 1) : r * (qd + 1) + (xcd - r) * qd) + idx;
            if (c >= n) continue;                  // (n < 8: workgroups b >= n do not exist in the staged launch either)
            for (int t = wgoff[ss.wgoff_off + c]; t < wgoff[ss.wgoff_off + c + 1]; ++t) {
                const int4 &tl = tiles[ss.tiles_off + t];
                const int slot = tl.x & 0xff, mi = tl.x >> 8, ks = tl.w;
                const int id = st[slot] & ~STAGE_SPILL_IN;
                const ProbSpec &q = pl->probs[id];
                int d[FWD_TILE_INT4 * 4] = {0};
                d[0] = id | (mi << 8);
                d[1] = tl.y;
                d[2] = tl.z;
                d[3] = ks;
                d[5] = fw.cnt_base[id] + tl.y / 32;
                {   // granules of 64 columns this tile completes for each of its units
                    const int g0 = tl.z / COL_GRANULE, g1 = std::min(gcols[id], (tl.z + tile_width(ks)) / COL_GRANULE);
                    d[6] = std::max(g1 - g0, 0);
                    if (d[6] <= 0) return false;
                }
                const int r1 = std::min(tl.y + mi * 32, M[id]);
                const int w0 = tl.y / q.rows_per_window, w1 = (r1 - 1) / q.rows_per_window + 1;
                std::vector<int> deps(q.deps);
                std::sort(deps.begin(), deps.end());
                deps.erase(std::unique(deps.begin(), deps.end()), deps.end());
                int nd = 0;
                for (int dp : deps) {
                    const ProbSpec &pq = pl->probs[dp];
                    const int a0 = w0 * pq.rows_per_window, a1 = std::min(w1 * pq.rows_per_window, M[dp]);
                    if (a1 <= a0) continue;
                    const int u0 = a0 / 32, u1 = (a1 + 31) / 32;
                    if (nd == FWD_MAX_DEP || u1 - u0 > 0xffff || gcols[dp] > 0x7fff) return false;
                    d[8 + 2 * nd] = fw.cnt_base[dp] + u0;
                    d[9 + 2 * nd] = (u1 - u0) | (gcols[dp] << 16);
                    ++nd;
                }
                d[4] = nd;
                d[7] = quiet[id] ? 1 : 0;              // (any kind of tile: a first-layer tile whose consumers all poll needs no drain either)
                lists[b].insert(lists[b].end(), d, d + FWD_TILE_INT4 * 4);
            }
        }
    }
    out_tiles.clear();
    out_wgoff.assign(1, 0);
    int grid = 0;
    for (int b = 0; b < nwg; ++b) {
        out_tiles.insert(out_tiles.end(), lists[b].begin(), lists[b].end());
        out_wgoff.push_back((int)(out_tiles.size() / (FWD_TILE_INT4 * 4)));
        if (!lists[b].empty()) grid = b + 1;
    }
    out_wgoff.resize(grid + 1);
    fw.grid = grid;
    fw.ntiles = (int)(out_tiles.size() / (FWD_TILE_INT4 * 4));
    dump_fwd(pl, B, grid, out_tiles, out_wgoff);
    return grid > 0;
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

// Cycles of one SIMD for a 32-row unit: two wavefronts share the pipe, 16 MFMAs x 64 cycles each per
// K-loop iteration (2048), plus staging/barrier overhead per iteration and a share of the tile
// prologue/epilogue.  Calibrated with the phase stamps of tools/gemm_probe -DR3D_TIMING at 2.2 GHz: a
// split-K piece (one unit per tile) pays 1.5 us of prologue latency, 0.4 us of reduction, 0.8 us of
// epilogue and 0.4 us between tiles, and 1.17 us per iteration; units of the wide tiles share one
// prologue and run at 1.0 us per iteration.
// Units of one to three iterations (the camera-embedding layers, K = 2 / 32) are all prologue and epilogue: 4 us per
// unit by the workgroups' busy times (-DR3D_TIMING, R3D_TIMING_ALL), whatever the tile height.
// (R3D_COST="iter,fixed,ks_iter,ks_fixed,first_extra,first_extra_wide,pair_scale[,nb_iter_extra,nb_fixed_extra]": the constants, for tools/tune_cost.py)
struct CostModel {
    // (first_extra_wide 12 -> 7 and the fused pairs' second layer priced 8 % up: tools/tune_cost.py on the GPU - 1.946 against
    //  1.976 ms at 1024 windows, 0.591 against 0.589 at 256; the other constants sit on a plateau)
    double iter = 2200.0, fixed = 2500.0, ks_iter = 2600.0, ks_fixed = 6800.0, first_extra = 5.0, first_extra_wide = 7.0, pair_scale = 1.08;
    double nb_iter_extra = 60.0, nb_fixed_extra = 1500.0;      // (gemm_tile_nb on top of the whole tile's staging / prologue: nb_cycles)
};
const CostModel &cost_model() {
    static const CostModel c = [] {
        CostModel m;
        if (const char *e = hook_env("R3D_COST"))
            sscanf(e, "%lf,%lf,%lf,%lf,%lf,%lf,%lf,%lf,%lf", &m.iter, &m.fixed, &m.ks_iter, &m.ks_fixed, &m.first_extra, &m.first_extra_wide, &m.pair_scale,
                   &m.nb_iter_extra, &m.nb_fixed_extra);
        return m;
    }();
    return c;
}
double unit_cycles(int iters, int ks) {
    const CostModel &c = cost_model();
    if (ks > 1) return iters * c.ks_iter + c.ks_fixed;
    return iters * c.iter + c.fixed + (iters < 4 ? (4 - iters) * 1400.0 : 0.0);
}

// a single-unit tile of nb column blocks (gemm_tile_nb): every SIMD issues 16 + 4 (nb - 4) MFMAs of 64 cycles per K tile; staging and
// barrier per K tile as in the whole tile (iter - 2048), prologue / epilogue plus the reduction of the split blocks
double nb_cycles(int nk, int nb) {
    const CostModel &c = cost_model();
    return nk * (64.0 * (16 + 4 * (nb - 4)) + (c.iter - 2048.0) + c.nb_iter_extra) + c.fixed + c.nb_fixed_extra;
}

// a GEMV tile (r3d_kernels.hip): one memory round trip for the weights of its 32 columns, the operand copy, two barriers,
// the reduction and the store - ~4 us plus the stream (K tiles / 8 per wavefront)
double gemv_cycles(int nk) { return 8000.0 + nk / 8.0 * 900.0; }
// a latency tile: the same plus 16 MFMAs per K tile of a wavefront's share (two wavefronts per SIMD)
double lat_cycles(int nk) { return 9000.0 + nk / 8.0 * 2600.0; }

struct Run {           // `n` consecutive row units of one column block, all in one workgroup's chunk
    int prob, col0, ks, u0, n, cap;
};

struct Seg {           // one column block of a problem
    int prob, col0, units, nk, cap, max_ks;
    double c1;
    int ncols;         // columns of the problem inside this block (<= 256)
    bool gemv = false; // 32-column GEMV tiles instead of row units
    int code = 8;      // ... their tile code: 8 GEMV, 16 latency tile
};

struct Assignment {
    std::vector<std::vector<Run>> bins;
    double worst = 0;
};

// First-fit-decreasing feasibility test for a chunk budget T.  Phase 1 places whole units (largest
// first, consecutive units of a column block into consecutive workgroups: weights and rows stay
// local).  Units that found no room - or are larger than T altogether - are cut along K into `ksplit`
// column sub-blocks (split-K tiles) and fill the remaining room.
bool assign(const std::vector<Seg> &segs, int nbins, double T, int ksplit, Assignment *out) {
    std::vector<double> room(nbins, T);
    if (out) out->bins.assign(nbins, {});
    struct Left { const Seg *s; int u0; };
    std::vector<Left> left;
    int cursor = 0;
    double cursor_cost = -1;
    // the costliest class of units is spread evenly over the workgroups instead of filling them one by one: when a
    // launch mixes two sizes (the trajectory model's first-level tiles are 1.4x the body branches'), the big ones
    // must not pile up in a few chunks that the small ones then cannot level
    long long top_units = 0;
    const double top_c1 = segs.empty() ? 0.0 : segs.front().c1;
    for (const Seg &s : segs)
        if (s.c1 == top_c1) top_units += s.units;
    // (two per workgroup where one would do - 64-row first-level tiles of the trajectory model instead of 32-row ones - was
    //  measured: 0.586 against 0.5786 ms at 256 windows, 1.928 against 1.930 at 1024; round 5)
    const int top_cap = (int)std::max<long long>(1, (top_units + nbins - 1) / nbins);
    for (const Seg &s : segs) {          // sorted by c1, descending
        if (s.gemv) {
            // one tile per 32 columns, each to the workgroup with the most room (they are many and short)
            // (equal pieces: to the workgroups in the order of their room, round after round)
            const double ck = s.code == 8 ? gemv_cycles(s.nk) : lat_cycles(s.nk);
            std::vector<int> by_room(nbins);
            for (int b = 0; b < nbins; ++b) by_room[b] = b;
            std::stable_sort(by_room.begin(), by_room.end(), [&](int x, int y) { return room[x] > room[y]; });
            for (int j = 0; j * 32 < s.ncols; ++j) {
                const int best = by_room[j % nbins];
                if (room[best] + 1e-6 < ck) return false;
                room[best] -= ck;
                if (out) out->bins[best].push_back({s.prob, s.col0 + j * 32, s.code, 0, 1, 1});
            }
            continue;
        }
        if (s.c1 != cursor_cost) { cursor = 0; cursor_cost = s.c1; }
        const int spread = s.c1 == top_c1 && segs.back().c1 != top_c1 ? top_cap : 1 << 30;
        int u = 0;
        if (s.c1 * 6.0 <= T && s.c1 * s.units <= 2.0 * T && s.c1 != top_c1) {
            // fillers (units far below the budget in a column block that is small change for the launch as a whole: the
            // camera embedding, GlobalInfo's gathered first layer) go two at a
            // time to the workgroup with the most room left, not to the first one they fit: their modelled cost is the
            // least certain, and first-fit stacked a dozen of them on ONE full workgroup that then ended 20 us after the
            // rest of the launch while others idled (fp32 first-level launch: 249 against 228 us; bf16x3 second launch:
            // 57 against 48 us)
            // (a heap of the workgroups by room: the roomiest first, lower index first at equal room - as a linear scan would pick)
            auto less_room = [&](int x, int y) { return room[x] != room[y] ? room[x] < room[y] : x > y; };
            std::vector<int> heap(nbins);
            for (int b = 0; b < nbins; ++b) heap[b] = b;
            std::make_heap(heap.begin(), heap.end(), less_room);
            while (u < s.units) {
                std::pop_heap(heap.begin(), heap.end(), less_room);
                const int best = heap.back();
                if (room[best] + 1e-6 < s.c1) break;
                const int take = std::min(std::min(2, (int)std::floor((room[best] + 1e-6) / s.c1)), s.units - u);
                room[best] -= take * s.c1;
                if (out) out->bins[best].push_back({s.prob, s.col0, 1, u, take, s.cap});
                u += take;
                std::push_heap(heap.begin(), heap.end(), less_room);
            }
        } else if (s.c1 <= T) {
            while (u < s.units) {
                while (cursor < nbins && room[cursor] + 1e-6 < s.c1) ++cursor;
                if (cursor == nbins) break;
                const int take = std::min(std::min((int)std::floor((room[cursor] + 1e-6) / s.c1), spread), s.units - u);
                room[cursor] -= take * s.c1;
                if (out) out->bins[cursor].push_back({s.prob, s.col0, 1, u, take, s.cap});
                u += take;
                if (take == spread) ++cursor;
            }
        }
        if (u < s.units) {
            if (s.max_ks < 2) return false;
            left.push_back({&s, u});
        }
    }
    // phase 2: split-K pieces, largest first
    std::stable_sort(left.begin(), left.end(), [](const Left &a, const Left &b) { return a.s->nk > b.s->nk; });
    cursor = 0;
    cursor_cost = -1;
    for (const Left &l : left) {
        const Seg &s = *l.s;
        int ks = std::min(ksplit, s.max_ks);
        while (ks > 1 && s.nk / ks < 2) ks /= 2;              // keep at least two K-loop iterations
        if (ks == 1) return false;
        const double ck = unit_cycles((s.nk + ks - 1) / ks, ks);
        if (ck > T) return false;
        if (ck != cursor_cost) { cursor = 0; cursor_cost = ck; }
        for (int j = 0; j < ks; ++j) {
            if (j * (256 / ks) >= s.ncols) break;               // a ragged last block: nothing to compute there
            int u = l.u0;
            while (u < s.units) {
                while (cursor < nbins && room[cursor] + 1e-6 < ck) ++cursor;
                if (cursor == nbins) return false;
                const int take = std::min((int)std::floor((room[cursor] + 1e-6) / ck), s.units - u);
                room[cursor] -= take * ck;
                if (out) out->bins[cursor].push_back({s.prob, s.col0 + j * (256 / ks), ks, u, take, ks == 2 ? 2 : 1});
                u += take;
            }
        }
    }
    if (out) {
        out->worst = 0;
        for (double r : room) out->worst = std::max(out->worst, T - r);
    }
    return true;
}

}  // namespace

namespace {

struct Packed {
    Assignment a;
    double T = 0, total = 0;
    int ks = 0;
};

// smallest feasible chunk budget for `nbins` workgroups and tiles of at most `max_units` row units
void pack(const std::vector<SchedProb> &probs, int nbins_max, int max_units, Packed &out) {
    std::vector<Seg> segs;
    double total = 0, biggest_fixed = 0;
    long long total_units = 0;
    for (int i = 0; i < (int)probs.size(); ++i) {
        const SchedProb &p = probs[i];
        const int units = (std::max(p.M - p.row0, 0) + 31) / 32;
        const double c1 = unit_cycles(p.nk + p.nk2, 1);
        for (int c0 = 0; c0 < p.N; c0 += 256) {
            Seg sg{i, c0, units, p.nk, p.max_units > 0 ? std::min(p.max_units, max_units) : max_units, p.max_ks, c1,
                   std::min(256, p.N - c0)};
            sg.gemv = (p.gemv || p.lat) && units == 1 && p.row0 == 0;
            sg.code = p.gemv ? 8 : 16;
            segs.push_back(sg);
            if (sg.gemv) {
                const int pieces = (sg.ncols + 31) / 32;
                total += pieces * (p.gemv ? gemv_cycles(p.nk) : lat_cycles(p.nk));
                total_units += pieces;
                continue;
            }
            total += units * c1;
            total_units += units;
        }
        if ((p.gemv || p.lat) && units == 1 && p.row0 == 0) biggest_fixed = std::max(biggest_fixed, p.gemv ? gemv_cycles(p.nk) : lat_cycles(p.nk));
        else if (p.max_ks < 2) biggest_fixed = std::max(biggest_fixed, c1);
    }
    std::stable_sort(segs.begin(), segs.end(), [](const Seg &a, const Seg &b) { return a.c1 > b.c1; });
    const int nbins = (int)std::min<long long>(nbins_max, std::max<long long>(total_units * 4, 1));
    // split-K pieces of 128 and of 64 columns; wider pieces win ties
    out.ks = 0;
    int widest_split = 1;
    for (const Seg &sg : segs) widest_split = std::max(widest_split, sg.max_ks);
    for (int ksplit = 2; ksplit <= 4; ksplit *= 2) {
        if (ksplit > 2 && widest_split < ksplit) break;      // (nothing in this launch can be cut that finely: same packing)
        double lo = std::max(total / nbins, biggest_fixed), top = std::max(total, lo) + 1.0;
        for (const Seg &s : segs) top = std::max(top, (s.gemv ? lat_cycles(s.nk) * 8 : s.c1 * s.units) + 1.0);
        // gallop up from the ideal budget (a feasible one is rarely more than a unit above it), then bisect
        double hi = lo;
        if (!assign(segs, nbins, lo, ksplit, nullptr)) {
            double step = 0.03 * lo + 1.0;
            for (hi = std::min(lo + step, top); hi < top && !assign(segs, nbins, hi, ksplit, nullptr); hi = std::min(hi + step, top)) {
                lo = hi;
                step *= 2.0;
            }
        }
        for (int it = 0; it < 40 && hi - lo > 0.001 * hi; ++it) {
            const double mid = 0.5 * (lo + hi);
            if (assign(segs, nbins, mid, ksplit, nullptr)) hi = mid; else lo = mid;
        }
        if (out.ks == 0 || hi < out.T * 0.98) {
            out.T = hi;
            out.ks = ksplit;
        }
    }
    assign(segs, nbins, out.T, out.ks, &out.a);
    out.total = total;
}

}  // namespace

// One-tile-deep launches (the M = B levels of a 256-window call: 160 - 224 whole 32 x 256 tiles for 256 CUs): the launch is
// as long as one tile whatever the packing, and a quarter of the chip idles.  gemm_tile_nb's tiles of 4 - 7 column blocks are
// proportionally shorter, so the eligible problems' rows of N / 32 blocks are cut into more, narrower tiles - one per
// workgroup - and the rest of the launch (pyramid riders, short layers) is packed classically into the workgroups left.
// Returns false when that is not shorter than the classic packing by the cost model.
static bool pack_nb(const std::vector<SchedProb> &probs, int nwg, int max_units, const Packed &classic, Packed &out) {
    std::vector<int> elig, rest_idx;
    std::vector<SchedProb> rest;
    for (int i = 0; i < (int)probs.size(); ++i) {
        const SchedProb &p = probs[i];
        if (p.nb_ok && p.row0 == 0 && p.M > 0 && p.N % 32 == 0) elig.push_back(i);
        else { rest_idx.push_back(i); rest.push_back(p); }
    }
    if (elig.empty()) return false;
    // only where the classic packing is one whole tile per workgroup for these problems: its budget is a single unit's time
    double unit_max = 0;
    long long whole_tiles = 0;
    for (int i : elig) {
        unit_max = std::max(unit_max, unit_cycles(probs[i].nk, 1));
        whole_tiles += (long long)((probs[i].M + 31) / 32) * ((probs[i].N + 255) / 256);
    }
    const bool force = hook_on("R3D_NB_FORCE");        // (hooks build: A/B of the cost model's verdict)
    if (whole_tiles > nwg || (classic.T > 1.05 * unit_max && !force)) return false;
    double best_T = force ? 1e30 : classic.T * 0.97;   // (modelled gains of a few percent do not materialise)
    int best_nb = 0;
    Packed best_rest;
    for (int nbmax = force ? 4 : 5; nbmax <= 7; ++nbmax) {
        long long ntiles = 0;
        double T_nb = 0;
        for (int i : elig) {
            const int nblk = probs[i].N / 32, per_row = (nblk + nbmax - 1) / nbmax, widest = (nblk + per_row - 1) / per_row;
            if ((widest < 5 && !force) || nblk / per_row < 4) { ntiles = nwg + 1; break; }   // (the kernel's tile kinds: 4 .. 7 blocks, 8 = the whole tile)
            ntiles += (long long)((probs[i].M + 31) / 32) * per_row;
            T_nb = std::max(T_nb, nb_cycles(probs[i].nk, widest));
        }
        if (ntiles > nwg) continue;
        double T = T_nb;
        Packed pr;
        if (!rest.empty()) {
            if (nwg - ntiles < 1) continue;
            pack(rest, (int)(nwg - ntiles), max_units, pr);
            T = std::max(T, pr.T);
        }
        if (T < best_T) { best_T = T; best_nb = nbmax; best_rest = pr; }
    }
    if (!best_nb) return false;
    out = Packed();
    out.T = best_T;
    out.ks = best_rest.ks ? best_rest.ks : 1;
    out.total = 0;
    // the narrow tiles first: (problem, column tile, unit) - consecutive workgroups share a column tile's weights
    for (int i : elig) {
        const int nblk = probs[i].N / 32, per_row = (nblk + best_nb - 1) / best_nb, units = (probs[i].M + 31) / 32;
        int b0 = 0;
        for (int t = 0; t < per_row; ++t) {
            const int w = (nblk - b0 + (per_row - t) - 1) / (per_row - t);      // evenly sized, wider ones first
            for (int u = 0; u < units; ++u) {
                out.a.bins.push_back({Run{i, b0 * 32, NB_CODE + w, u, 1, 1}});          // (4 <= w <= 7: best_nb is 5 .. 7, rows split evenly)
                out.total += nb_cycles(probs[i].nk, w);
            }
            b0 += w;
        }
    }
    for (const auto &bin : best_rest.a.bins) {
        if (bin.empty()) continue;
        std::vector<Run> b2 = bin;
        for (Run &r : b2) r.prob = rest_idx[r.prob];
        out.a.bins.push_back(b2);
    }
    out.total += best_rest.total;
    out.a.worst = best_T;
    return true;
}

void schedule_stage(const std::vector<SchedProb> &probs, int nwg, int max_units, std::vector<int4> &tiles,
                    std::vector<int> &wgoff, StageSchedule &out, bool enc) {
    Packed big, narrow;
    const Packed *best = &big;
    out.kind = enc ? STAGE_ENC : STAGE_BIG;
    pack(probs, nwg, max_units, big);
    if (!enc && pack_nb(probs, nwg, max_units, big, narrow)) best = &narrow;
    out.ks = 1;
    out.tiles_off = tiles.size();
    out.wgoff_off = wgoff.size();
    const size_t t0 = tiles.size();
    int grid = 0;
    wgoff.push_back(0);
    for (const auto &bin : best->a.bins) {
        if (bin.empty()) continue;
        for (const Run &r : bin) {
            // emit the run as evenly sized tiles of <= cap units
            const int nt = (r.n + r.cap - 1) / r.cap;
            int done = 0;
            for (int k = 0; k < nt; ++k) {
                const int sz = (r.n - done + (nt - k) - 1) / (nt - k);
                tiles.push_back(make_int4(r.prob | (sz << 8), probs[r.prob].row0 + (r.u0 + done) * 32, r.col0, r.ks));
                done += sz;
            }
            out.ks = std::max(out.ks, r.ks);
        }
        wgoff.push_back((int)(tiles.size() - t0));
        ++grid;
    }
    if (grid == 0) { wgoff.push_back(0); grid = 1; }
    if (const char *e = hook_env("R3D_SCHED_DUMP")) {   // development aid: the chunks of every launch, one line per workgroup
        (void)e;
        fprintf(stderr, "[sched] launch of %zu problems: budget %.0f cycles, ideal %.0f, split-K %d\n", probs.size(), best->T,
                best->total / std::max(grid, 1), best->ks);
        for (size_t i = 0; i < probs.size(); ++i)
            fprintf(stderr, "[sched]   p%zu: M %d row0 %d N %d nk %d nk2 %d max_ks %d max_units %d unit %.0f cycles\n", i, probs[i].M, probs[i].row0,
                    probs[i].N, probs[i].nk, probs[i].nk2, probs[i].max_ks, probs[i].max_units, unit_cycles(probs[i].nk + probs[i].nk2, 1));
        for (int b = 0; b < grid; ++b) {
            fprintf(stderr, "[sched] chunk %3d:", b);
            for (int t = wgoff[out.wgoff_off + b]; t < wgoff[out.wgoff_off + b + 1]; ++t) {
                const int4 &tl = tiles[t0 + t];
                fprintf(stderr, " p%d/mi%d/r%d/c%d/ks%d", tl.x & 0xff, tl.x >> 8, tl.y, tl.z, tl.w);
            }
            fprintf(stderr, "\n");
        }
    }
    out.nwg = grid;
    out.ntiles = (int)(tiles.size() - t0);
    out.imbalance = best->a.worst / std::max(best->total / grid, 1.0);
    out.makespan = best->a.worst;
}

// What the packers need to know about one problem of the plan at B windows: K-loop iterations of a 32-row unit (nk, plus
// nk2 for the further layers of fused tiles and for measured extras), how far it may be cut along K, the tile height cap.
static SchedProb sched_prob_of(const Plan *pl, const ProbSpec &q, int64_t B) {
        const Layer &L = pl->m[q.model]->layers[q.layer];
        const int M = (int)(B * q.rows_per_window);
        // fused-prologue tiles hold the whole encoded operand in 64 KiB of LDS: rows * (K + 4) floats
        const int enc_cap = q.enc_lut >= 0 ? std::max(1, std::min(3, (64 * 1024) / ((L.Kpad + 4) * 4 * 32))) : 0;
        // split-K sub-tiles of one iteration must come from one buffer of a concatenated operand
        int max_ks = q.enc_lut >= 0 ? 1 : 4;
        for (int sgi = 0, k = 0; sgi + 1 < q.nseg; ++sgi) {
            k += q.seg[sgi].width;
            while (max_ks > 1 && k % (BK * max_ks)) max_ks /= 2;
        }
        SchedProb sp{M, L.N, L.Kpad / BK, max_ks, enc_cap};
        if (q.nseg == 1 && q.seg[0].width < L.Kpad) sp.max_ks = 1;   // an operand narrower than its padded K: one bounded descriptor
        // a plain layer of a few rows (calls of up to eight windows: the MLPs, the top of the pyramid): GEMV tiles
        const bool narrow_ok = q.enc_lut < 0 && q.layer2 < 0 && !(q.nseg == 1 && q.seg[0].width < L.Kpad) && L.Kpad >= 64;
        sp.gemv = M <= GEMV_ROWS && narrow_ok && !hook_on("R3D_NO_GEMV");
        // ... and of up to 32 rows (one unit): latency tiles on the matrix cores (not beside the bf16x3 tiles: B >= 96 there)
        sp.lat = !sp.gemv && M <= 32 && narrow_ok && !hook_on("R3D_NO_LAT");
        const bool b3 = B >= b3_min_batch();               // (r3d_api.cpp passes the bf16x3 operands under the same condition)
        // single-unit tiles of 5 - 7 column blocks (gemm_tile_nb) for the wide plain fp32 layers: the FCBlocks' 1024-wide Linears
        sp.nb_ok = q.enc_lut < 0 && q.layer2 < 0 && !sp.gemv && !sp.lat && !(b3 && L.bf3) && L.N % 32 == 0 && L.N >= 512 &&
                   L.Kpad / BK >= 8 && !hook_on("R3D_NO_NB");
        if (b3 && L.bf3 && q.layer2 < 0 && q.enc_lut < 0) {   // bf16-matrix-core tiles: whole tiles of <= 128 rows, ~1.5x the iteration rate
            sp.max_ks = 1;
            sp.max_units = 4;
            sp.nk = (sp.nk * 3 + 2) / 4;            // (31 against 39 us for a single-unit M = B tile by the per-tile stamps)
        }
        if (q.enc_lut >= 0 && q.layer3 < 0) {
            // a gathered operand without the fused level (GlobalInfo's current frames in the first-level launch): table load,
            // scattered gather and a three-slab epilogue around two K tiles - 4-5 us per unit by the workgroups' busy times,
            // not the 3.3 us of a plain two-iteration unit.  Under-priced, a dozen workgroups collected all of them on top
            // of five first-level units and ended 15-25 us after the rest of the launch.
            sp.nk2 = 2;
        }
        if (q.layer3 >= 0) {                       // fused first level: one 32-row tile runs three layers (three input rows per row)
            const Model *mm = pl->m[q.model];
            sp.max_ks = 1;
            sp.max_units = 2;                          // tap by tap, a tile holds 64 output rows (r3d_kernels.hip, first_level_taps)
            // (+ the gather passes over the tile's 96 first-layer rows and the phase changes: one pass when the operand
            // tile is narrow enough to hold all rows, three otherwise; 5 / 12 iterations' time by the phase stamps:
            // 45.5 us for a body-part tile, 67.5 us for the trajectory model's at 2.1 GHz)
            sp.nk2 = 2 * sp.nk + mm->layers[q.layer2].Kpad / BK + mm->layers[q.layer3].Kpad / BK + (int)(L.Kpad <= 64 ? cost_model().first_extra : cost_model().first_extra_wide);
            if (b3 && L.bf3_conv) {
                // on the bf16 matrix cores, by the per-tile stamps of the single-launch forward (tools/fwd_gantt.py): a
                // body-part tile 57.7 us against 89.3 in fp32 (0.65x), the trajectory model's 60 against 73 (0.82x).  (Round 2,
                // launch by launch, did better with one 0.7x for both; with tile-level dependencies the measured factors win:
                // 0.465 against 0.471 ms at 256 windows, 1.387 against 1.406 at 1024.)
                sp.nk2 = (sp.nk2 + sp.nk) * (L.Kpad <= 64 ? 65 : 82) / 100 - sp.nk;
            }
        } else if (q.layer2 >= 0) {                // fused pair: whole tiles of <= 128 rows, no split
            sp.max_ks = 1;
            sp.max_units = 4;
            sp.nk2 = (int)(pl->m[q.model]->layers[q.layer2].Kpad / BK * cost_model().pair_scale + 0.5);
            if (b3 && L.bf3_conv && pl->m[q.model]->layers[q.layer2].bf3_conv && q.nseg == 1) {   // on the bf16 matrix cores: tiles of <= 96 rows,
                sp.max_units = 3;                                                            // 0.65x (two units) .. 0.8x (one) the time per unit
                sp.nk = (sp.nk * 7 + 5) / 10;
                sp.nk2 = (sp.nk2 * 7 + 5) / 10;
            }
        }
        return sp;
}

// `spill_row0`: rows [spill_row0, M) of Plan::spill_prob belong to the launch that lists it with STAGE_SPILL_IN, the
// rows before to its own launch (-1: all rows stay).
struct StageMemo {
    std::vector<int4> tiles;
    std::vector<int> wgoff;
    StageSchedule ss;
};
static thread_local std::map<std::vector<int>, StageMemo> g_stage_memo;     // per schedule_build_host call

static void build_stage(const Plan *pl, const std::vector<int> &st, int64_t B, int nwg, int spill_row0, std::vector<int4> &tiles,
                        std::vector<int> &wgoff, StageSchedule &out) {
    std::vector<SchedProb> probs;
    double flops = 0, bytes = 0;
    for (int i = 0; i < (int)st.size(); ++i) {
        const int id = st[i] & ~STAGE_SPILL_IN;
        const ProbSpec &q = pl->probs[id];
        const Layer &L = pl->m[q.model]->layers[q.layer];
        const int M_all = (int)(B * q.rows_per_window);
        int M = M_all, row0 = 0;
        if (st[i] & STAGE_SPILL_IN) row0 = spill_row0 >= 0 ? spill_row0 : M_all;
        else if (id == pl->spill_prob && spill_row0 >= 0) M = spill_row0;
        const double share = M_all > 0 ? (double)(M - row0) / M_all : 0.0;
        SchedProb sp = sched_prob_of(pl, q, B);
        sp.M = M;
        sp.row0 = row0;
        probs.push_back(sp);
        flops += q.flops_per_window * (double)B * share;
        if (M == row0) continue;
        bytes += 4.0 * ((double)(M - row0) * L.K + (double)L.N * L.K + (double)(M - row0) * L.N * (q.res_buf >= 0 ? 2.0 : 1.0));
        if (q.layer2 >= 0) bytes += 4.0 * (double)L.N * pl->m[q.model]->layers[q.layer2].K;
        if (q.layer3 >= 0) bytes += 4.0 * (double)L.N * L.N + 4.0 * 2.0 * (double)(M - row0) * L.K;   // (three input rows per output row)
    }
    bool enc = false;
    for (int e : st) enc = enc || pl->probs[e & ~STAGE_SPILL_IN].enc_kernel;
    // The level assignments and spill candidates schedule_build_host compares differ in two or three launches only: a
    // launch's packing is a function of its problems' shapes, so it is computed once per distinct launch of a build.
    std::vector<int> key{enc ? 1 : 0, nwg};
    for (const SchedProb &sp : probs)
        key.insert(key.end(), {sp.M, sp.N, sp.nk, sp.max_ks, sp.max_units, sp.nk2, sp.row0, (sp.gemv ? 1 : sp.lat ? 2 : 0) + (sp.nb_ok ? 4 : 0)});
    auto hit = g_stage_memo.find(key);
    if (hit == g_stage_memo.end()) {
        StageMemo m;
        // the fused-prologue kernel runs two workgroups per CU (one encodes while the other multiplies)
        schedule_stage(probs, enc ? 2 * nwg : nwg, GEMM_SCHED_MAX_UNITS, m.tiles, m.wgoff, m.ss, enc);
        hit = g_stage_memo.emplace(key, std::move(m)).first;
    }
    const StageMemo &m = hit->second;
    out = m.ss;
    out.tiles_off = tiles.size();
    out.wgoff_off = wgoff.size();
    tiles.insert(tiles.end(), m.tiles.begin(), m.tiles.end());
    wgoff.insert(wgoff.end(), m.wgoff.begin(), m.wgoff.end());
    out.flops = flops;
    out.bytes = bytes;
}

// Host part of schedule_get: picks the level assignment and the row spill for this batch size by the modelled
// length of the launches (sum over launches of the longest chunk), then builds every launch's tile lists.
const std::vector<std::vector<int>> *schedule_build_host(const Plan *pl, int64_t B, int nwg, int &spill_row0, std::vector<int4> &tiles,
                                                        std::vector<int> &wgoff, std::vector<StageSchedule> &stages) {
    spill_row0 = -1;
    g_stage_memo.clear();
    const std::vector<std::vector<int>> *levels = &pl->stages;
    const bool dump = hook_env("R3D_PLAN_DUMP") != nullptr;
    const auto t_begin = std::chrono::steady_clock::now();
    auto build_all = [&](const std::vector<std::vector<int>> &lv, int row0, std::vector<int4> &t, std::vector<int> &w,
                         std::vector<StageSchedule> &ss) {
        double sum = 0;
        for (const auto &st : lv) {
            StageSchedule a{};
            build_stage(pl, st, B, nwg, row0, t, w, a);
            ss.push_back(a);
            sum += a.makespan;
        }
        return sum;
    };
    if (!pl->stages_asap.empty() && B <= GEMV_ROWS) {
        build_all(pl->stages_asap, -1, tiles, wgoff, stages);
        return &pl->stages_asap;
    }
    const double c_plain = build_all(pl->stages, -1, tiles, wgoff, stages);
    double c_best = c_plain;
    std::vector<int> searched0, searched1;      // the two launches the last row search ran on, and what it found
    int searched_row0 = -1;
    auto try_spill = [&](const std::vector<std::vector<int>> &lv) {
        if (pl->spill_prob < 0 || lv.empty()) return;
        // the launch that holds the spilling problem and the one after it
        int s0 = -1, s1 = -1;
        for (int si = 0; si < (int)lv.size(); ++si)
            for (int e : lv[si]) {
                if (e == pl->spill_prob) s0 = si;
                if (e == (pl->spill_prob | STAGE_SPILL_IN)) s1 = si;
            }
        const ProbSpec &q = pl->probs[pl->spill_prob];
        const int M_all = (int)(B * q.rows_per_window);
        const long long own = (M_all + 31) / 32;
        long long fl_tiles = 0;
        for (int e : lv[std::max(s0, 0)])
            if (!(e & STAGE_SPILL_IN) && pl->probs[e].layer3 >= 0) fl_tiles += (B * pl->probs[e].rows_per_window + 31) / 32;
        if (!(s0 >= 0 && s1 >= 0 && fl_tiles > nwg)) return;
        // candidates for the tiles that run late: none, the remainder of the division of the first-level tiles by
        // the CU count, and multiples of 32 up to half a round
        const long long rem = fl_tiles % nwg;
        std::vector<long long> cands{0};
        if (rem > 0 && rem <= nwg / 2) cands.push_back(rem);
        for (long long r = 32; r <= nwg / 2; r += 32)
            if (r != rem) cands.push_back(r);
        double best = 0;
        int best_row0 = M_all;
        const bool same_pair = lv[s0] == searched0 && lv[s1] == searched1;   // (the variants differ in later launches only)
        if (same_pair) best_row0 = searched_row0;
        for (long long r : cands) {
            if (same_pair) break;
            if (r >= own) continue;
            const int row0 = r ? (int)((own - r) * 32) : M_all;
            std::vector<int4> t;
            std::vector<int> w;
            StageSchedule a{}, b{};
            build_stage(pl, lv[s0], B, nwg, row0, t, w, a);
            build_stage(pl, lv[s1], B, nwg, row0, t, w, b);
            const double cost = a.makespan + b.makespan;
            if (r == 0 || cost < best * 0.995) { best = cost; best_row0 = row0; }
        }
        searched0 = lv[s0];
        searched1 = lv[s1];
        searched_row0 = best_row0;
        if (best_row0 >= M_all) return;
        std::vector<int4> t;
        std::vector<int> w;
        std::vector<StageSchedule> ss;
        const double c_spill = build_all(lv, best_row0, t, w, ss);
        if (dump) fprintf(stderr, "[plan] B=%lld: %d rows late: modelled %.0f cycles, best so far %.0f\n", (long long)B, M_all - best_row0, c_spill, c_best);
        if (c_spill < c_best * 0.99) {      // (measured: modelled gains under 1 % do not materialise)
            c_best = c_spill;
            levels = &lv;
            spill_row0 = best_row0;
            tiles.swap(t);
            wgoff.swap(w);
            stages.swap(ss);
        }
    };
    try_spill(pl->stages_spill);
    try_spill(pl->stages_spill_alt);
    if (dump)
        fprintf(stderr, "[plan] B=%lld: schedule built in %.2f ms\n", (long long)B,
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count());
    return levels;
}

Schedule *schedule_get(Plan *pl, int64_t B, int nwg, bool pin, int lane) {
    // (R3D_OPT_LANES: every lane has its own schedule of a batch size - its own tile lists for its share of the CUs and, above all,
    //  its own control region: forwards of different lanes run at the same time)
    const int64_t key = schedule_key(B, lane);
    auto it = pl->schedules.find(key);
    if (it != pl->schedules.end()) {
        // least recently USED goes first: a hit moves the size to the back of the queue
        auto pos = std::find(pl->schedule_lru.begin(), pl->schedule_lru.end(), key);
        if (pos != pl->schedule_lru.end() && pos + 1 != pl->schedule_lru.end()) std::rotate(pos, pos + 1, pl->schedule_lru.end());
        if (pin) it->second->pinned = true;
        return it->second;
    }
    // The cache is bounded at 64 UNPINNED batch sizes (per lane): the least recently used one goes.  Sizes named in r3d_prepare are
    // pinned - a captured hipGraph replays kernels whose arguments point into the schedule and never comes back here -
    // and do not count.  The victim's launches may still be in flight on any stream, hence the device-wide
    // synchronisation before its tile lists are freed; when that fails (another stream is capturing, a sticky error)
    // nothing is freed and the cache grows by one instead.
    {
        size_t unpinned = 0;
        for (int64_t b : pl->schedule_lru) unpinned += pl->schedules[b]->pinned ? 0 : 1;
        // (R3D_OPT_LANES: every lane holds its own schedule of a size - the bound is per lane, or an evaluation's 25 call sizes on four
        //  lanes would evict, and synchronise the device, with every call)
        const size_t cap = 64 * (size_t)std::max(1, pl->m[0] ? pl->m[0]->lanes : 1);
        if (unpinned >= cap) {
            auto victim = std::find_if(pl->schedule_lru.begin(), pl->schedule_lru.end(), [&](int64_t b) { return !pl->schedules[b]->pinned; });
            if (victim != pl->schedule_lru.end() && hipDeviceSynchronize() == hipSuccess) {
                const int64_t old = *victim;
                pl->schedule_lru.erase(victim);
                for (const Model *mm : pl->m)                  // (r3d_last_clock must not read a freed control region)
                    if (mm) const_cast<Model *>(mm)->last_clk_dev = nullptr;
                delete pl->schedules[old];
                pl->schedules.erase(old);
            } else {
                (void)hipGetLastError();       // (the failed synchronisation is not this call's error)
            }
        }
    }
    Schedule *s = new Schedule();
    s->B = B;
    std::vector<int4> tiles;
    std::vector<int> wgoff;
    s->levels = schedule_build_host(pl, B, nwg, s->spill_row0, tiles, wgoff, s->stages);
    hipError_t e;
    if ((e = hipMalloc((void **)&s->d_tiles, std::max<size_t>(tiles.size(), 1) * sizeof(int4))) != hipSuccess ||
        (e = hipMalloc((void **)&s->d_wgoff, std::max<size_t>(wgoff.size(), 1) * sizeof(int))) != hipSuccess ||
        (e = hipMemcpy(s->d_tiles, tiles.data(), tiles.size() * sizeof(int4), hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(s->d_wgoff, wgoff.data(), wgoff.size() * sizeof(int), hipMemcpyHostToDevice)) != hipSuccess) {
        hip_fail(e, "schedule upload");
        delete s;
        return nullptr;
    }
    // ---- the per-frame first layers of a clip call (Plan::frame_probs): one launch of gathered GEMMs over B + RF - 3 rows
    if (pl->frame_buf >= 0) {
        const int rows = (int)(B + pl->m[0]->RF - 3);
        std::vector<SchedProb> fp;
        for (const auto &f : pl->frame_probs) {
            const Layer &L = pl->m[f.model]->layers[f.layer];
            SchedProb sp{rows, L.N, L.Kpad / BK, 1, std::max(1, std::min(3, (64 * 1024) / ((L.Kpad + 4) * 4 * 32)))};
            sp.nk2 = 2;
            fp.push_back(sp);
        }
        std::vector<int4> ft;
        std::vector<int> fo;
        schedule_stage(fp, nwg, GEMM_SCHED_MAX_UNITS, ft, fo, s->frame_stage, false);
        if ((e = hipMalloc((void **)&s->d_frame_tiles, std::max<size_t>(ft.size(), 1) * sizeof(int4))) != hipSuccess ||
            (e = hipMalloc((void **)&s->d_frame_wgoff, std::max<size_t>(fo.size(), 1) * sizeof(int))) != hipSuccess ||
            (e = hipMemcpy(s->d_frame_tiles, ft.data(), ft.size() * sizeof(int4), hipMemcpyHostToDevice)) != hipSuccess ||
            (e = hipMemcpy(s->d_frame_wgoff, fo.data(), fo.size() * sizeof(int), hipMemcpyHostToDevice)) != hipSuccess) {
            hip_fail(e, "schedule upload (per-frame launch)");
            delete s;
            return nullptr;
        }
        s->frame_stage.tiles_off = s->frame_stage.wgoff_off = 0;
    }
    // ---- the single-launch form: tile lists with dependencies + the relative problem tables (rays / UV input)
    if (forward_single_launch()) {
        std::vector<int> ft, fo;
        Schedule::Fwd &fw = s->fwd;
        if (schedule_build_fwd(pl, B, nwg, *s->levels, s->stages, tiles, wgoff, fw, ft, fo)) {
            const Model *a = pl->m[0];
            bool ok = (e = hipMalloc((void **)&fw.d_tiles, ft.size() * sizeof(int))) == hipSuccess &&
                      (e = hipMalloc((void **)&fw.d_wgoff, fo.size() * sizeof(int))) == hipSuccess &&
                      (e = hipMemcpy(fw.d_tiles, ft.data(), ft.size() * sizeof(int), hipMemcpyHostToDevice)) == hipSuccess &&
                      (e = hipMemcpy(fw.d_wgoff, fo.data(), fo.size() * sizeof(int), hipMemcpyHostToDevice)) == hipSuccess;
            for (int v = 0; ok && v < 4; ++v) {           // v = UV input + 2 * first levels on the per-frame buffer
                const int uv = v & 1, shared = v >> 1;
                if (uv && a->cfg.in_features != 3) continue;
                if (shared && pl->frame_buf < 0) break;
                std::vector<GemmProb> rel(fw.nprob);
                std::vector<unsigned char> tags((size_t)fw.nprob * BIND_NPTR);
                Bases none;
                CallShape cs;
                cs.uv = uv != 0;
                cs.shared = shared != 0;
                if (shared) {                              // (a clip call: window stride one frame - r3d_api.cpp, run)
                    cs.window_stride = 1;
                    cs.frames = B + a->RF - 1;
                }
                bool filled = true;
                for (int i = 0; i < fw.nprob && filled; ++i)
                    filled = fill_prob(pl, pl->probs[i], B, a, none, cs, rel[i], tags.data() + (size_t)i * BIND_NPTR) == R3D_OK;
                if (!filled) { if (v) continue; ok = false; break; }
                ok = (e = hipMalloc((void **)&fw.d_rel[v], rel.size() * sizeof(GemmProb))) == hipSuccess &&
                     (e = hipMalloc((void **)&fw.d_tags[v], tags.size())) == hipSuccess &&
                     (e = hipMemcpy(fw.d_rel[v], rel.data(), rel.size() * sizeof(GemmProb), hipMemcpyHostToDevice)) == hipSuccess &&
                     (e = hipMemcpy(fw.d_tags[v], tags.data(), tags.size(), hipMemcpyHostToDevice)) == hipSuccess;
            }
            fw.h_tiles = ft;
            fw.h_wgoff = fo;
            if (ok) {                  // the library-owned control region (Schedule::Fwd::d_ctrl): two counter banks + the bound table
                fw.bank_bytes = ((size_t)(fw.ncnt + 4) * sizeof(unsigned) + 255) / 256 * 256;
                ok = (e = hipMalloc((void **)&fw.d_ctrl, 2 * fw.bank_bytes + 2 * (size_t)fw.nprob * sizeof(GemmProb))) == hipSuccess;
            }
            bool narrow = false;       // GEMV / latency tiles in the lists: activation banks of the library's own (poll mode)
            for (size_t t = 0; t * FWD_TILE_INT4 * 4 < ft.size(); ++t) narrow |= tile_is_narrow(ft[t * FWD_TILE_INT4 * 4 + 3]);
            // which specialisation of the persistent kernel runs these lists (r3d_kernels.hip, R3D_FORWARD_KERNEL)
            const bool b3_tiles = B >= b3_min_batch() && ((pl->m[0] && pl->m[0]->use_b3) || (pl->m[1] && pl->m[1]->use_b3));
            fw.kernel = narrow ? FWD_KERNEL_LAT : b3_tiles ? FWD_KERNEL_B3 : FWD_KERNEL_F32;
            // The single launch is only correct with ALL its workgroups resident.  Checked here against what the device
            // holds of that kernel; a CU mask (which the occupancy query does not see) or a lists-mix no specialisation
            // carries sends the size to the launch-by-launch form instead.
            if (ok) {
                // (every specialisation that may run these lists: the selected kind, its pixel-input form, and the clip
                //  kernels when the plan has a per-frame buffer - their register / LDS figures are their own)
                int cap = forward_resident_capacity(fw.kernel, false);
                auto also = [&](int kind, bool uvk) {
                    const int c = forward_resident_capacity(kind, uvk);
                    if (c > 0) cap = cap > 0 ? std::min(cap, c) : c;
                };
                if (fw.uses_gather && pl->m[0]->cfg.in_features == 3) also(fw.kernel, true);
                if (fw.kernel == FWD_KERNEL_F32 && hook_on("R3D_CHAIN")) also(FWD_KERNEL_CHAIN, false);   // (experiment: r3d_forward_chain_f32 runs these lists)
                if (pl->frame_buf >= 0 && fw.kernel == FWD_KERNEL_F32) {
                    also(FWD_KERNEL_CLIP, false);
                    if (pl->m[0]->cfg.in_features == 3) also(FWD_KERNEL_CLIP, true);
                }
                const bool masked = getenv("HSA_CU_MASK") != nullptr || getenv("ROC_GLOBAL_CU_MASK") != nullptr;
                if ((narrow && b3_tiles) || (cap > 0 && fw.grid > cap) || masked) {
                    if (fw.d_tiles) (void)hipFree(fw.d_tiles);
                    if (fw.d_wgoff) (void)hipFree(fw.d_wgoff);
                    if (fw.d_ctrl) (void)hipFree(fw.d_ctrl);
                    for (int i = 0; i < 4; ++i) {
                        if (fw.d_rel[i]) (void)hipFree(fw.d_rel[i]);
                        if (fw.d_tags[i]) (void)hipFree(fw.d_tags[i]);
                    }
                    fw = Schedule::Fwd();
                    s->pinned = pin;
                    pl->schedules[key] = s;
                    pl->schedule_lru.push_back(key);
                    return s;
                }
            }
            // (up to 16 windows: at 32 the sentinel fill of a 10 MB bank costs more than the hops save - 0.219 against 0.213 ms)
            if (ok && narrow && pl->kind == PLAN_SMALL && B <= 16) {     // (that plan writes no activation twice: r3d_plan.cpp, single_assign)
                fw.act_bytes = (((size_t)pl->floats_per_window * (size_t)B + (size_t)pl->tail_floats + 64) * sizeof(float) + 255) / 256 * 256;
                ok = (e = hipMalloc((void **)&fw.d_act, 2 * fw.act_bytes)) == hipSuccess;
            }
            if (!ok) {
                hip_fail(e, "schedule upload (single-launch form)");
                delete s;
                return nullptr;
            }
        } else {
            fw = Schedule::Fwd();      // this plan runs launch by launch
        }
    }
    s->pinned = pin;
    pl->schedules[key] = s;
    pl->schedule_lru.push_back(key);
    return s;
}

}  // namespace r3d
