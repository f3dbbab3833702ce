// Launch plan of the lifting forward pass: which GEMMs exist, what they read/write, and which
// of them can share a launch.
//
// The reference executes RIEModel.forward / RIETrajectoryModel.forward (lib/model/rie.py:284-434,
// :518-559) as ~170 small ATen ops per network.  Here both networks are lowered once, per pair of
// configurations, to a DAG of BN-folded GEMM "problems" over channels-last activations
// (Conv1d(k=3,stride=3) on (B,T,C) is a plain GEMM on the (B*T/3, 3C) view), and
// the DAG is levelised: all problems of equal depth - the six temporal branches, the GlobalInfo
// MLPs, the five FuseBlocks ... - run as ONE grouped launch.  Concatenations (rie.py:371-407) are
// never materialised: a problem's A operand is a list of column segments of other buffers.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <mutex>

#include "r3d_internal.hpp"

namespace r3d {

namespace {

struct Builder {
    Plan &p;
    int mi;             // model index inside the plan
    const Model &m;

    int buffer(const std::string &name, int64_t floats_per_window, int external = 0) {
        BufferSpec b;
        b.name = name;
        b.floats_per_window = (floats_per_window + 15) / 16 * 16;
        b.external = external;
        b.offset_per_window = 0;
        p.buffers.push_back(b);
        return (int)p.buffers.size() - 1;
    }
    bool first_level_fused = false;
    bool enc_in_gemm = false;    // gathered first layers as enc_tile problems of r3d_gemm_f32 (not r3d_gemm_enc_f32): calls of few windows,
                                 // where the second kernel's two-workgroups-per-CU overlap buys nothing and a launch of its own shape
                                 // would keep the call out of the single-launch form
    bool single_assign = false;  // no buffer element is written twice in a call (the plan of calls of few windows: their GEMV /
                                 // latency tiles may take data as its own ready flag - r3d_kernels.hip, ACT_SENTINEL - which a
                                 // stale value from earlier in the call would defeat)
    bool fuse_pairs = true;
    bool fuse_top = false;       // the top pyramid level (one row per window) as a fused pair too
    struct In { int buf, col, ld, width, dep; };
    int problem(const std::string &layer_prefix, int rows_pw, const std::vector<In> &ins, int res_buf, int res_col,
                int res_ld, int c_buf, int c_col, int c_ld, std::vector<int> extra_deps = {}, int enc_lut = -1,
                int enc_lut_uv = -1) {
        ProbSpec q;
        q.model = mi;
        q.layer = m.layer_index.at(layer_prefix);
        q.rows_per_window = rows_pw;
        q.nseg = (int)ins.size();
        q.layer2 = -1;
        q.layer3 = -1;
        q.enc_lut = enc_lut;
        q.enc_lut_uv = m.cfg.in_features == 3 ? enc_lut_uv : -1;
        q.enc_kernel = enc_lut >= 0 && !first_level_fused && !enc_in_gemm;
        q.enc_rows = rows_pw;
        q.enc_step = 3;
        const Layer &L0 = m.layers[q.layer];
        int k = enc_lut >= 0 ? L0.Kpad : 0;
        for (int s = 0; s < q.nseg; ++s) {
            q.seg[s] = {ins[s].buf, ins[s].col, ins[s].ld, ins[s].width};
            k += ins[s].width;
            if (ins[s].dep >= 0) q.deps.push_back(ins[s].dep);
        }
        for (int d : extra_deps)
            if (d >= 0) q.deps.push_back(d);
        const Layer &L = m.layers[q.layer];
        if (k != L.Kpad && !(q.nseg == 1 && k == L.K)) {   // (one narrow operand: the kernel's descriptor bound zero-fills up to Kpad)
            fprintf(stderr, "r3d: internal plan error: layer %s expects K=%d, operands give %d\n", layer_prefix.c_str(), L.Kpad, k);
            abort();
        }
        q.res_buf = res_buf; q.res_col = res_col; q.res_ld = res_ld;
        q.c_buf = c_buf; q.c_col = c_col; q.c_ld = c_ld;
        // level 0 is the fused-prologue launch (encoded operands only) - unless the first level runs fused in the
        // GEMM kernel, which takes plain problems alongside
        q.depth = enc_lut >= 0 || first_level_fused || enc_in_gemm ? 0 : 1;
        for (int d : q.deps) q.depth = std::max(q.depth, p.probs[d].depth + 1);
        q.flops_per_window = 2.0 * rows_pw * (double)L.K * (double)L.N;
        p.probs.push_back(q);
        return (int)p.probs.size() - 1;
    }

    // FCBlock.forward (lib/model/rie.py:159-169): fc_1+bn+lrelu, n residual units, fc_2.
    // c_buf < 0: the last Linear is left to the fused decoder kernel (recorded in plan.decs).
    int fc_block(const std::string &prefix, const std::vector<In> &ins, int nblocks, int c_buf, int c_col, int c_ld,
                 int enc_lut = -1, int enc_lut_uv = -1) {
        const int H = MLP_HIDDEN;
        int h = buffer(prefix + ".h", H), y = buffer(prefix + ".y", H);
        int last = problem(prefix + ".fc_1", 1, ins, -1, 0, 0, h, 0, H, {}, enc_lut, enc_lut_uv);
        for (int n = 0; n < nblocks; ++n) {
            const std::string q = prefix + ".layers." + std::to_string(n);
            if (single_assign && n > 0) y = buffer(prefix + ".y" + std::to_string(n), H);
            const int p1 = problem(q + ".w1", 1, {{h, 0, H, H, last}}, -1, 0, 0, y, 0, H);
            // out = x + lrelu(bn(w2 y))  (rie.py:122-135): residual = h, written in place - or, where every element of a
            // call's activations must be written exactly once (single_assign), into a buffer of its own
            const int hn = single_assign ? buffer(prefix + ".h" + std::to_string(n + 1), H) : h;
            last = problem(q + ".w2", 1, {{y, 0, H, H, p1}}, h, 0, H, hn, 0, H, {last});
            h = hn;
        }
        if (c_buf < 0) {
            p.decs.push_back({mi, m.layer_index.at(prefix + ".fc_2"), h});
            return last;
        }
        return problem(prefix + ".fc_2", 1, {{h, 0, H, H, last}}, -1, 0, 0, c_buf, c_col, c_ld);
    }

    // The dense ablation (rie.py:49-53 with Optimize1f == False): stride-1 convolutions of 3, then 2*3^i + 1 taps.  Every
    // level keeps RF rows per window (position p of a window at row w*RF + p; the valid prefix shrinks by 2*pad per level
    // and the rows behind it hold garbage no valid row ever reads), so that an operand row - the taps*C contiguous floats
    // that start at position p - is addressed with a plain leading dimension C: overlapping rows, no im2col.
    int temporal_block_dense(int bi, int c_buf, int c_col, int c_ld) {
        const Model::Branch &br = m.branches[bi];
        const int C = m.cfg.channels, L = m.cfg.num_levels, RF = m.RF;
        const int pp[2] = {buffer(br.prefix + ".P0", (int64_t)RF * C), L > 1 ? buffer(br.prefix + ".P1", (int64_t)RF * C) : -1};
        const int hb = L > 1 ? buffer(br.prefix + ".H", (int64_t)RF * C) : -1;
        int last = problem(br.prefix + ".expand_conv", RF, {}, -1, 0, 0, pp[0], 0, C, {}, (int)br.lut_off, (int)br.lut_uv_off);
        p.probs[last].enc_step = 1;
        int dil = 3;
        for (int i = 1; i < L; ++i, dil *= 3) {
            const int src = pp[(i - 1) & 1], dst = pp[i & 1], taps = 2 * dil + 1;
            const std::string a = br.prefix + ".layers_conv." + std::to_string(2 * (i - 1));
            const std::string b = br.prefix + ".layers_conv." + std::to_string(2 * (i - 1) + 1);
            // res = x[:, :, pad+shift : T-pad+shift] (rie.py:91-92): position p + pad (+ pad when causal) of the input
            const int rc = (dil + (m.cfg.causal ? dil : 0)) * C;
            const int pa = problem(a, RF, {{src, 0, C, taps * C, last}}, -1, 0, 0, hb, 0, C);
            last = problem(b, RF, {{hb, 0, C, C, pa}}, src, rc, C, dst, 0, C, {last});
        }
        p.tail_floats = std::max<int64_t>(p.tail_floats, (int64_t)(RF + 8) * C);
        const int fin = pp[(L - 1) & 1];
        return finish(br, fin, RF * C, last, c_buf, c_col, c_ld);
    }

    // the block's last activations (one row of C floats per window at leading dimension `ld`) -> its `shrink`; or, when
    // shrink is folded into the layers that read it (Layer::pre), just remembered for them
    struct Local { int buf, ld, prob; };
    std::vector<Local> locals;                               // per branch, in the order the blocks are built
    int finish(const Model::Branch &br, int fin, int ld, int last, int c_buf, int c_col, int c_ld) {
        const int C = m.cfg.channels;
        if (m.fold_shrink) {
            p.probs[last].flops_per_window += 2.0 * (double)C * (double)m.cfg.latent;   // (the reference's arithmetic, still counted)
            locals.push_back({fin, ld, last});
            return last;
        }
        const int q = problem(br.prefix + ".shrink", 1, {{fin, 0, ld, C, last}}, -1, 0, 0, c_buf, c_col, c_ld);
        locals.push_back({c_buf, c_ld, q});
        return q;
    }

    // TemporalBlock.forward (rie.py:85-105) for branch `bi`; returns the shrink problem id.
    int temporal_block(int bi, int c_buf, int c_col, int c_ld) {
        if (m.cfg.dense) return temporal_block_dense(bi, c_buf, c_col, c_ld);
        const Model::Branch &br = m.branches[bi];
        const int C = m.cfg.channels, L = m.cfg.num_levels;
        int rows = m.RF / 3;
        // One buffer per level (and one per un-fused level's intermediate): level i's output has rows / 3^(i-1) rows per
        // window.  Never ping-pong: in the single-launch forward a level-(i+1) tile may run while level-i tiles of OTHER
        // windows still read level i-1's output, and the only ordering between tiles is producer -> consumer of the same
        // windows.  (With the first level fused the expand_conv output never exists in memory.)
        std::vector<int> lvl(L + 1, -1);             // lvl[i]: output of level i (0 = expand_conv)
        {
            int r = rows;
            for (int i = 0; i < L; ++i, r = std::max(r / 3, 1))
                if (!(i == 0 && first_level_fused)) lvl[i] = buffer(br.prefix + ".L" + std::to_string(i), (int64_t)r * C);
        }
        // a level's 1x1 convolution is applied to the 3-tap one's output tile inside the kernel when a tile holds
        // all of its columns (r3d_kernels.hip, PAIR); otherwise the intermediate goes through a buffer
        // Fusing costs the level its split-K freedom (a fused tile is a whole 32-row unit through both layers), so
        // the top of the pyramid - one row per window, fewer units than CUs at the usual batch sizes - stays unfused.
        auto fuse = [&](int level_rows) { return C <= N_ALIGN && (level_rows >= 3 || fuse_top) && fuse_pairs; };
        // first layer: the A operand is generated from the raw input inside the kernel (fused prologue)
        int last, i0 = 1;
        if (first_level_fused) {
            // expand_conv + level 1 (3-tap and 1x1 convolutions) as one problem of rows/3 output rows per window
            const std::string a = br.prefix + ".layers_conv.0", b = br.prefix + ".layers_conv.1";
            last = problem(br.prefix + ".expand_conv", rows / 3, {}, -1, 0, 0, lvl[1], 0, C, {}, (int)br.lut_off, (int)br.lut_uv_off);
            ProbSpec &q = p.probs[last];
            q.enc_rows = rows;
            q.layer2 = m.layer_index.at(a);
            q.layer3 = m.layer_index.at(b);
            q.flops_per_window = 2.0 * rows * (double)m.layers[q.layer].K * C + 2.0 * (rows / 3) * (3.0 * C * C + (double)C * C);
            rows /= 3;
            i0 = 2;
        } else {
            last = problem(br.prefix + ".expand_conv", rows, {}, -1, 0, 0, lvl[0], 0, C, {}, (int)br.lut_off, (int)br.lut_uv_off);
        }
        for (int i = i0; i < L; ++i) {
            const int src = lvl[i - 1], dst = lvl[i];
            rows /= 3;
            const std::string a = br.prefix + ".layers_conv." + std::to_string(2 * (i - 1));
            const std::string b = br.prefix + ".layers_conv." + std::to_string(2 * (i - 1) + 1);
            // three consecutive frames of the previous level form one GEMM row (stride == kernel);
            // res = x[:, :, 1::3] is the centre third of that row (rie.py:94); the last third for causal models
            // (rie.py:92 with shift == pad)
            const int rc = (1 + m.cfg.causal) * C;
            if (fuse(rows)) {
                last = problem(a, rows, {{src, 0, 3 * C, 3 * C, last}}, src, rc, 3 * C, dst, 0, C);
                ProbSpec &q = p.probs[last];
                q.layer2 = m.layer_index.at(b);
                q.flops_per_window += 2.0 * rows * (double)C * (double)C;
            } else {
                const int hb = buffer(br.prefix + ".H" + std::to_string(i), (int64_t)rows * C);
                const int pa = problem(a, rows, {{src, 0, 3 * C, 3 * C, last}}, -1, 0, 0, hb, 0, C);
                last = problem(b, rows, {{hb, 0, C, C, pa}}, src, rc, 3 * C, dst, 0, C, {last});
            }
        }
        const int fin = lvl[L - 1];
        return finish(br, fin, C, last, c_buf, c_col, c_ld);
    }
};

}  // namespace

static Plan *build_plan(const Model *a, const Model *b, int kind) {
    Plan *pl = new Plan();
    pl->m[0] = a;
    pl->m[1] = b;
    pl->kind = kind;
    const bool small = kind == PLAN_SMALL;
    for (int mi = 0; mi < 2; ++mi) {
        const Model *m = pl->m[mi];
        if (!m) continue;
        Builder B{*pl, mi, *m};
        // the first pyramid level runs fused (r3d_kernels.hip, first_level_taps) when a tile can hold it: at least two
        // levels, all channels in one 256-column tile, first-layer operand tile next to the intermediate.  Decided once
        // for the PAIR (both models or neither): a launch is shared by the two models, and one that mixed the fused
        // kernel's problems with r3d_gemm_enc_f32's could not be launched (pos and trj may differ in CHANNELS).
        {
            auto can_fuse = [&](const Model *mm) {
                int k0max = 0;
                for (const auto &br : mm->branches) k0max = std::max(k0max, br.k0pad);
                return mm->cfg.num_levels >= 2 && mm->cfg.channels <= N_ALIGN && k0max <= 256 && !mm->cfg.dense;
            };
            bool all = true;
            for (const Model *mm : pl->m)
                if (mm) all = all && can_fuse(mm);
            B.first_level_fused = all && !small;
            B.enc_in_gemm = small;
            B.single_assign = small;
            B.fuse_pairs = (kind == PLAN_FUSED || kind == PLAN_LARGE);
            B.fuse_top = kind == PLAN_LARGE;
        }
        const int lat = m->cfg.latent, D = m->cfg.embed_dim;
        int pe = -1;
        if (D > 0) {
            // Embedding.forward (lib/model/embedding.py:15-18) on the caller's [height, pitch] rows
            if (pl->param_buf < 0) pl->param_buf = B.buffer("param", 0, 3);
            const int eh = B.buffer("emb.h", EMBED_MID);
            pl->emb_buf[mi] = B.buffer("emb", D);
            const int E = m->cfg.extrinsic_dim;
            const int p1 = B.problem("embedder.w1", 1, {{pl->param_buf, 0, 0, E, -1}}, -1, 0, 0, eh, 0, EMBED_MID);
            pe = B.problem("embedder.w2", 1, {{eh, 0, EMBED_MID, EMBED_MID, p1}}, -1, 0, 0, pl->emb_buf[mi], 0, D);
        }
        const int g = B.buffer("global", lat);
        // in_current = x[:, RF // F] (rie.py:290-292): gathered from the raw input like the first layers' operands (one
        // row per window, every column relative to the window's current frame) - in UV mode encoded on the way
        const int pg = B.fc_block("GlobalInfo", {}, 2, g, 0, lat, (int)m->global_lut_off, (int)m->global_lut_uv_off);
        if (m->cfg.kind == R3D_KIND_POS) {
            pl->pos_model = mi;
            const bool fold = m->fold_shrink;
            const int C = m->cfg.channels;
            const int tmp5 = fold ? -1 : B.buffer("tmp5", 5 * lat);
            int sh[5];
            for (int bi = 0; bi < 5; ++bi) sh[bi] = B.temporal_block(bi, tmp5, bi * lat, 5 * lat);
            int mix5 = -1, pf[5] = {-1, -1, -1, -1, -1};
            if (m->cfg.stage != 1) {
                mix5 = B.buffer("mix5", 5 * lat);
                for (int i = 0; i < 5; ++i) {
                    // cat of the other four local features (rie.py:393-394) = <=2 column ranges of tmp5 - or, with
                    // shrink folded into fc_1, the four blocks' last activations
                    std::vector<Builder::In> ins;
                    if (fold) {
                        for (int k = 0; k < 5; ++k)
                            if (k != i) ins.push_back({B.locals[k].buf, 0, B.locals[k].ld, C, sh[k]});
                    } else {
                        if (i > 0) ins.push_back({tmp5, 0, 5 * lat, i * lat, sh[0]});
                        if (i < 4) ins.push_back({tmp5, (i + 1) * lat, 5 * lat, (4 - i) * lat, sh[4]});
                    }
                    // every shrink must be complete, not only the two named above
                    const int first = (int)pl->probs.size();
                    pf[i] = B.fc_block("FuseBlocks." + std::to_string(i), ins, 1, mix5, i * lat, 5 * lat);
                    for (int k = 0; k < 5; ++k) pl->probs[first].deps.push_back(sh[k]);
                    // re-derive depths of the block just added (deps were extended)
                    for (int q = first; q < (int)pl->probs.size(); ++q) {
                        int d = 0;
                        for (int dep : pl->probs[q].deps) d = std::max(d, pl->probs[dep].depth + 1);
                        pl->probs[q].depth = d;
                    }
                }
            }
            for (int bi = 0; bi < 5; ++bi) {
                // cat(local, [mix], global, [embedding])  (rie.py:376-407)
                std::vector<Builder::In> ins;
                if (fold) ins.push_back({B.locals[bi].buf, 0, B.locals[bi].ld, C, sh[bi]});
                else ins.push_back({tmp5, bi * lat, 5 * lat, lat, sh[bi]});
                if (mix5 >= 0) ins.push_back({mix5, bi * lat, 5 * lat, lat, pf[bi]});
                ins.push_back({g, 0, lat, lat, pg});
                if (D > 0) ins.push_back({pl->emb_buf[mi], 0, D, D, pe});
                B.fc_block(std::string("Integration_") + (bi == 0 ? "Torso" : bi == 1 ? "LArm" : bi == 2 ? "RArm" : bi == 3 ? "LLeg" : "RLeg"),
                           ins, 1, -1, 0, 0);
            }
        } else {
            pl->trj_model = mi;
            const int local = m->fold_shrink ? -1 : B.buffer("local", lat);
            const int sh = B.temporal_block(0, local, 0, lat);
            std::vector<Builder::In> ins;
            if (m->fold_shrink) ins.push_back({B.locals[0].buf, 0, B.locals[0].ld, m->cfg.channels, sh});
            else ins.push_back({local, 0, lat, lat, sh});
            ins.push_back({g, 0, lat, lat, pg});
            if (D > 0) ins.push_back({pl->emb_buf[mi], 0, D, D, pe});
            B.fc_block("Integration", ins, 1, -1, 0, 0);
        }
    }
    // Two level assignments are kept; schedule_get picks one per batch size by the modelled length of the launches.
    //  * plain: every problem as early as its inputs allow, except the movable ones (below).
    //  * spill (Plan::spill_prob): the first-level launch is made of tiles of 32 output rows of a branch, rarely a
    //    multiple of the CU count of them: 1296 at 256 windows leave 16 tiles for a sixth round that 240 CUs sit
    //    out.  The trajectory model's chain is four launches shorter than the pose model's, so here it sits one
    //    launch lower and the tail of ITS first level may run next to the second pyramid level of the pose
    //    branches, a launch with CUs to spare; the rest of its pyramid then rides wherever a launch is as long.
    std::vector<int> asap;
    for (const auto &q : pl->probs) asap.push_back(q.depth);
    // `pin_first`: the problem that reads the spilled one stays right behind it instead of moving as late as it can
    auto levelise = [&](bool spill, bool pin_first, std::vector<std::vector<int>> &stages, bool early = false) -> bool {
        const int n = (int)pl->probs.size();
        for (int i = 0; i < n; ++i) pl->probs[i].depth = asap[i];
        int deepest = 0, pt = -1;
        for (const auto &q : pl->probs) deepest = std::max(deepest, q.depth);
        if (spill) {
            bool pos_fused = false;
            for (int i = 0; i < n; ++i) {
                const ProbSpec &q = pl->probs[i];
                if (q.layer3 >= 0 && pl->m[q.model]->cfg.kind == R3D_KIND_TRJ) pt = i;
                pos_fused = pos_fused || (q.layer3 >= 0 && pl->m[q.model]->cfg.kind == R3D_KIND_POS);
            }
            if (pt < 0 || !pos_fused) return false;
            for (auto &q : pl->probs)
                for (int d : q.deps)
                    if (d == pt) q.depth = std::max(q.depth, pl->probs[pt].depth + 2);
            int grown = 0;
            for (auto &q : pl->probs) {           // (problems are created in dependency order)
                for (int d : q.deps) q.depth = std::max(q.depth, pl->probs[d].depth + 1);
                grown = std::max(grown, q.depth);
            }
            if (grown > deepest) return false;
        }
        // The GlobalInfo MLP (rie.py:362) is independent of the conv pyramid and only needed by the Integration
        // blocks.  As early as possible its 32-iteration units ride in the short launches at the top of the pyramid
        // and define their length; as LATE as possible they ride in the FuseBlock launches, which have idle CUs
        // anyway.  So does the trajectory model's decoder MLP ("Integration."; the body-part decoders are
        // "Integration_<part>."), which only the final decoder kernel waits for - and, in the spill assignment, the
        // rest of the trajectory model's pyramid.
        std::vector<std::vector<int>> users(n);
        for (int i = 0; i < n; ++i)
            for (int d : pl->probs[i].deps) users[d].push_back(i);
        auto movable = [&](const ProbSpec &q) {
            // (a call of a few windows is a latency chain, not a packing problem: everything as early as its inputs allow,
            //  so that GlobalInfo's five layers are long done when the decoders ask for them)
            if (early) return false;
            const std::string &key = pl->m[q.model]->layers[q.layer].weight_key;
            if (pin_first && pt >= 0 && std::find(q.deps.begin(), q.deps.end(), pt) != q.deps.end()) return false;
            if (spill && pl->m[q.model]->cfg.kind == R3D_KIND_TRJ && key.rfind("LocalLayer.", 0) == 0 && q.layer3 < 0) return true;
            return (key.rfind("GlobalInfo.", 0) == 0 && key.rfind("GlobalInfo.fc_1", 0) != 0) || key.rfind("Integration.", 0) == 0;
        };
        auto iterations = [&](const ProbSpec &q) {      // K-loop iterations of one 32-row unit through the problem
            const Model *mm = pl->m[q.model];
            int it = mm->layers[q.layer].Kpad / BK;
            if (q.layer3 >= 0) it = 3 * it + mm->layers[q.layer2].Kpad / BK + mm->layers[q.layer3].Kpad / BK;
            else if (q.layer2 >= 0) it += mm->layers[q.layer2].Kpad / BK;
            return it;
        };
        // longest unit among the problems that stay where they are, per level: a launch that long takes a
        // movable problem's units along for free (as long as there are CUs to spare)
        std::vector<int> level_iters(deepest + 1, 0);
        for (const auto &q : pl->probs)
            if (!movable(q)) level_iters[q.depth] = std::max(level_iters[q.depth], iterations(q));
        for (int i = n - 1; i >= 0; --i) {
            ProbSpec &q = pl->probs[i];
            if (!movable(q)) continue;
            int latest = users[i].empty() ? deepest : 1 << 30;
            for (int u : users[i]) latest = std::min(latest, pl->probs[u].depth - 1);
            if (latest >= (1 << 30) || latest <= q.depth) continue;
            // the latest level in [earliest, latest] whose launch is at least as long as this problem's units;
            // failing that, the latest one
            int pick = latest;
            for (int l = latest; l >= q.depth; --l)
                if (level_iters[l] >= iterations(q)) { pick = l; break; }
            q.depth = pick;
        }
        int maxd = 0;
        for (auto &q : pl->probs) maxd = std::max(maxd, q.depth);
        std::vector<std::vector<int>> lv(maxd + 1);
        for (int i = 0; i < n; ++i) lv[pl->probs[i].depth].push_back(i);
        if (spill) lv[pl->probs[pt].depth + 1].push_back(pt | STAGE_SPILL_IN);
        if (hook_env("R3D_PLAN_DUMP"))
            for (size_t s = 0; s < lv.size(); ++s) {
                fprintf(stderr, "[plan%s] launch %zu:", spill ? " spill" : "", s);
                for (int e : lv[s]) {
                    const ProbSpec &q = pl->probs[e & ~STAGE_SPILL_IN];
                    fprintf(stderr, " %s%s(x%d)", e & STAGE_SPILL_IN ? "+" : "", pl->m[q.model]->layers[q.layer].weight_key.c_str(), q.rows_per_window);
                }
                fprintf(stderr, "\n");
            }
        // split launches that exceed the kernarg capacity
        stages.clear();
        for (auto &st : lv)
            for (size_t i = 0; i < st.size(); i += MAX_PROB)
                stages.emplace_back(st.begin() + i, st.begin() + std::min(st.size(), i + MAX_PROB));
        if (spill) pl->spill_prob = pt;
        return true;
    };
    const bool can_spill = pl->m[0] && pl->m[1] && !hook_on("R3D_NO_SPILL");
    if (!(can_spill && levelise(true, false, pl->stages_spill))) pl->stages_spill.clear();
    if (!(can_spill && levelise(true, true, pl->stages_spill_alt)) || pl->stages_spill_alt == pl->stages_spill) pl->stages_spill_alt.clear();
    levelise(false, false, pl->stages);
    // (the un-fused plan also as early as possible: what calls of up to four windows run - schedule_build_host)
    if (kind == PLAN_SMALL) {
        levelise(false, false, pl->stages_asap, true);
        if (pl->stages_asap == pl->stages) pl->stages_asap.clear();
    }
    // the per-frame buffer of clip calls (Plan::frame_buf): LAST, so that its RF - 1 rows beyond B lie in the tail
    {
        bool all = true;
        int col = 0;
        std::vector<Plan::FrameProb> fps;
        for (auto &q : pl->probs) {
            if (q.layer3 < 0) continue;
            const Model *mm = pl->m[q.model];
            const Model::Branch *br = nullptr;
            for (const auto &b : mm->branches)
                if ((int)b.lut_off == q.enc_lut) br = &b;
            if (!br || br->frame_layer < 0) { all = false; break; }
            q.frame_col = col;
            fps.push_back({q.model, br->frame_layer, col, (int)br->lut_frame_off, mm->cfg.in_features == 3 ? (int)br->lut_frame_uv_off : -1});
            col += (mm->layers[br->frame_layer].N + 15) / 16 * 16;
        }
        if (all && !fps.empty() && fps.size() <= (size_t)MAX_PROB) {
            Builder B{*pl, 0, *pl->m[0]};
            pl->frame_ld = col;
            pl->frame_buf = B.buffer("frames", col);
            pl->frame_probs = fps;
            int RF = 1;
            for (const Model *mm : pl->m)
                if (mm) RF = std::max(RF, mm->RF);
            pl->tail_floats = std::max<int64_t>(pl->tail_floats, (int64_t)(RF + 2) * col);
        } else {
            for (auto &q : pl->probs) q.frame_col = -1;
        }
    }
    // workspace offsets
    int64_t off = 0;
    for (auto &bf : pl->buffers) {
        bf.offset_per_window = off;
        off += bf.floats_per_window;
    }
    pl->floats_per_window = off;
    return pl;
}

// Plans are cached per pair of model ids.  Ids are unique for the life of the process (a pointer is not: a model
// freed and another allocated at its address must not inherit the plan, whose layer indices and K paddings belong to
// the old configuration), and a model's destructor / re-finalisation on another device drops every plan naming it.
static std::mutex g_plans_mutex;
static std::map<std::pair<std::pair<uint64_t, uint64_t>, int>, Plan *> g_plans;

// A call of few windows is one tile's latency per launch, and the fused tiles (first level tap by tap, two convolutions
// per pyramid level) are the LONG tiles.  Up to 48 windows the small plan leaves everything un-fused, so that every
// layer is a launch of split-K tiles - 17 launches instead of 13, 0.202 against 0.259 ms at one window, 0.216 against
// 0.266 at 16, 0.242 against 0.270 at 32, 0.250 against 0.269 at 48; up to 96 windows the pairs alone stay un-fused
// (first level fused): 0.270 against 0.287 ms at 64, 0.355 against 0.360 at 96; from 128 on everything fused wins
// (0.379 against 0.383; 0.608 against 0.627 at 256).  From 1024 windows on the top pyramid level (one row per window)
// has enough rows to run as a fused pair too (11 launches): 1.977 against 1.986 ms at 1024, 3.765 against 3.817 at 2048,
// 7.430 against 7.459 at 4096; at 512 it loses (1.092 against 1.083).  bench.py --batch.
constexpr int64_t SMALL_PLAN_MAX = 48, MEDIUM_PLAN_MAX = 96, LARGE_PLAN_MIN = 1024;

std::vector<int64_t> plan_kind_edges() { return {SMALL_PLAN_MAX, MEDIUM_PLAN_MAX, LARGE_PLAN_MIN - 1}; }

int plan_kind(int64_t B) {
    if (B >= LARGE_PLAN_MIN) return PLAN_LARGE;
    if (hook_on("R3D_NO_SMALL_PLAN")) return PLAN_FUSED;       // (hooks build: the parity tests run small calls on the fused plan too)
    return B <= SMALL_PLAN_MAX ? PLAN_SMALL : B <= MEDIUM_PLAN_MAX ? PLAN_MEDIUM : PLAN_FUSED;
}

Plan *plan_get(Model *a, Model *b, int kind) {
    std::lock_guard<std::mutex> lock(g_plans_mutex);
    const auto key = std::make_pair(std::make_pair(a->id, b ? b->id : (uint64_t)0), kind);
    auto it = g_plans.find(key);
    if (it != g_plans.end()) return it->second;
    Plan *p = build_plan(a, b, kind);
    g_plans[key] = p;
    return p;
}

void plans_drop(const Model *m) {
    std::lock_guard<std::mutex> lock(g_plans_mutex);
    for (auto it = g_plans.begin(); it != g_plans.end();) {
        if (it->first.first.first == m->id || it->first.first.second == m->id) {
            // (r3d_last_clock must not read the control region of a schedule that is freed here)
            for (const Model *mm : it->second->m)
                if (mm) const_cast<Model *>(mm)->last_clk_dev = nullptr;
            delete it->second;
            it = g_plans.erase(it);
        } else {
            ++it;
        }
    }
}

bool plans_pinned(const Model *m) {
    std::lock_guard<std::mutex> lock(g_plans_mutex);
    for (const auto &kv : g_plans)
        if (kv.first.first.first == m->id || kv.first.first.second == m->id)
            for (const auto &sc : kv.second->schedules)
                if (sc.second && sc.second->pinned) return true;
    return false;
}

}  // namespace r3d
