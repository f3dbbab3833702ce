// Host-only: prints the static schedule the library would build for the RF-243 plan's launch shapes.
#include <cstdio>
#include <map>
#include <string>
#include "r3d_internal.hpp"
using namespace r3d;
int main(int argc, char **argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 256, nwg = argc > 2 ? atoi(argv[2]) : 256;
    struct St { const char *name; std::vector<SchedProb> p; };
    auto rep = [](int n, SchedProb q) { return std::vector<SchedProb>(n, q); };
    auto cat = [](std::vector<SchedProb> a, std::vector<SchedProb> b) { a.insert(a.end(), b.begin(), b.end()); return a; };
    std::vector<St> st = {
        {"L1a+G", cat(rep(6, {B * 27, 256, 24, 4, 0}), rep(2, {B, 1024, 32, 4, 0}))},
        {"L1b+G", cat(rep(6, {B * 27, 256, 8, 4, 0}), rep(2, {B, 1024, 32, 4, 0}))},
        {"L2a+G", cat(rep(6, {B * 9, 256, 24, 4, 0}), rep(2, {B, 1024, 32, 4, 0}))},
        {"L2b+G", cat(rep(6, {B * 9, 256, 8, 4, 0}), rep(2, {B, 1024, 32, 4, 0}))},
        {"L3a+G", cat(rep(6, {B * 3, 256, 24, 4, 0}), rep(2, {B, 256, 32, 4, 0}))},
        {"L3b", rep(6, {B * 3, 256, 8, 4, 0})},
        {"L4a", rep(6, {B, 256, 24, 4, 0})},
        {"L4b", rep(6, {B, 256, 8, 4, 0})},
        {"shrink", rep(6, {B, 256, 8, 4, 0})},
        {"Fuse.fc1+T", cat(rep(5, {B, 1024, 32, 4, 0}), rep(1, {B, 1024, 18, 4, 0}))},
        {"Fuse.w1+T", rep(6, {B, 1024, 32, 4, 0})},
        {"Fuse.fc2", rep(5, {B, 256, 32, 4, 0})},
        {"Int.fc1", rep(5, {B, 1024, 26, 4, 0})},
        {"Int.w1", rep(5, {B, 1024, 32, 4, 0})},
    };
    for (auto &s : st) {
        std::vector<int4> tiles; std::vector<int> wgoff; StageSchedule ss{};
        schedule_stage(s.p, nwg, 6, tiles, wgoff, ss);
        std::map<int, int> shapes;
        for (auto &t : tiles) shapes[(t.w << 8) | (t.x >> 8)]++;
        printf("%-12s ks_cap %d grid %3d tiles %4d imbalance %.3f  shapes:", s.name, ss.ks, ss.nwg, ss.ntiles, ss.imbalance);
        for (auto &kv : shapes) printf(" ks%d/mi%d x%d", kv.first >> 8, kv.first & 255, kv.second);
        printf("\n");
        if (argc > 3) {   // chunk histogram: units per chunk by (ks)
            std::map<std::string, int> hist;
            for (int c = 0; c < ss.nwg; ++c) {
                char buf[128]; int n = 0; int cnt[5] = {0, 0, 0, 0, 0}; int kt = 0;
                for (int t = wgoff[c]; t < wgoff[c + 1]; ++t) { cnt[tiles[t].w] += tiles[t].x >> 8; kt += (tiles[t].x >> 8) * ((s.p[tiles[t].x & 255].nk + tiles[t].w - 1) / tiles[t].w); }
                n = snprintf(buf, sizeof buf, "k1:%d k2:%d k4:%d iters:%d", cnt[1], cnt[2], cnt[4], kt);
                hist[buf]++;
            }
            for (auto &kv : hist) printf("      %4d chunks of %s\n", kv.second, kv.first.c_str());
        }
    }
}
